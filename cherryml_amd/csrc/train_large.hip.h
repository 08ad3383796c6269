// Fused optimiser for large state spaces (S > 32, one bank): the reference's whole epoch loop
// (trainer.py:156-218 with rate.py:167-188, "pande_reversible") driven from C, no torch in the
// loop.  Same closed-form chain rule as train_small.hip.h, spread over the chip:
//
//   lt_build     pi = softmax(log_pi), d = sqrt(pi); A (padded, symmetric) straight from the parameters,
//                Q of this epoch -> Q_last (+ power-of-two snapshot)        1 workgroup per row
//   eigh, K1..K4 loss and G = dL/dA                                          (large_bank.hip.h)
//   lt_gd        dL/d log d_k (one wave per k)
//   lt_step      Adam / SGD on the S(S-1)/2 upper-diagonal logits, Q_best <- Q_last when this epoch
//                improved (1 workgroup per row) + loss curve, best loss, Adam / SGD on log_pi (1 workgroup)
//
//   A_ij = R_ij = softplus(up_k) mask_ij,  A_ii = -sum_j R_ij d_j / d_i,  Q_ij = R_ij d_j / d_i
//   dR_ij = mask_ij (G_ij - G_ii d_j / d_i);  dup_k = sigmoid(up_k) (dR_ij + dR_ji)
//   dld_k = -d_k sum_{i != k} G_ii A_ik / d_i - G_kk A_kk - (colsum_k - rowsum_k) / n
//   dlogpi_k = (dld_k - pi_k sum_m dld_m) / 2
#pragma once
#include "train_small.hip.h"

struct LargeTrain {
  int S, LD, do_adam, n_pow2;
  int epoch0;                           // epochs done by earlier calls of this optimisation (CB_TRAIN_RESUME)
  double *p_pi, *p_up;                  // [S], [S(S-1)/2]
  double *m_pi, *v_pi, *m_up, *v_up;    // Adam moments
  const double *mask;                   // [S][S] or null
  double lr, beta1, beta2, eps;
  double *pi, *dsq;                     // [LD]: softmax, its square root (pad: 0 / 1)
  double *A;                            // [LD][LD]
  const double *G;                      // [LD][LD] dL/dA of this epoch
  const double *loss;                   // device scalar: loss of this epoch
  const double *dirsum;                 // [S]
  double inv_n;
  double *gd;                           // [S] scratch: dL/d log d
  double *state;                        // [epoch & 1]: best loss before that epoch (lt_step)
  double *loss_curve;                   // [E]
  double *time_curve;                   // [E] or null: the 100 MHz wall clock at the end of each epoch's parameter step
  double *Q_last, *Q_best, *Q_pow2;     // [S][S], [S][S], [n_pow2][S][S]
  // sigma = max |A_ii| falls out of the build (round 6: the planned eigensolve's prologue, lge_begin, was a launch of its own --
  // one workgroup reading 400 diagonal entries, 5 us per epoch): every row's workgroup folds |A_ii| into sig_cur with an integer
  // maximum (non-negative doubles order like their bits), workgroup 0 clears the word of the NEXT epoch (the two are used in
  // turn) and, when a planned solve follows (ectl != null), resets its control block and statistics lines
  unsigned long long *sig_cur = nullptr, *sig_next = nullptr;
  unsigned long long *ectl = nullptr, *eacc = nullptr;
};

__device__ void lge_reset_words(unsigned long long *ctl, unsigned long long *acc);   // eigh_planned.hip.h

// (256 threads, fixed order: a butterfly inside each wave, then the four waves' sums -- two barriers; the shared-memory tree this
// replaces had nine, and lt_build goes through three such reductions behind one another: 9.7 -> 7 us per epoch)
__device__ __forceinline__ double lt_block_sum(double v, double *s) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ double lt_block_max(double v, double *s) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
  __syncthreads();
  return r;
}

// pi = softmax(log_pi) and d = sqrt(pi) are recomputed by EVERY row's workgroup (400 exponentials: nothing) with the same
// fixed-order reductions, so that the build needs no launch in front of it; workgroup i publishes pi_i, d_i for the kernels
// behind it (pad rows: 0 / 1).
__global__ __launch_bounds__(256) void lt_build(LargeTrain a, int epoch) {
  __shared__ double s[256];
  __shared__ double sd[1024];   // d_j (LD <= 1024)
  const int S = a.S, LD = a.LD, i = blockIdx.x;
  double *Arow = a.A + (size_t)i * LD;
  if (i == 0) {
    if (a.sig_next && threadIdx.x == 255) *a.sig_next = 0ull;
    if (a.ectl) lge_reset_words(a.ectl, a.eacc);
  }
  if (i >= S) {
    for (int j = threadIdx.x; j < LD; j += 256) Arow[j] = 0.0;
    if (threadIdx.x == 0) {
      a.pi[i] = 0.0;
      a.dsq[i] = 1.0;
    }
    return;
  }
  {
    double mx = -INFINITY;
    for (int k = threadIdx.x; k < S; k += 256) mx = fmax(mx, a.p_pi[k]);
    mx = lt_block_max(mx, s);
    double acc = 0.0;
    for (int k = threadIdx.x; k < S; k += 256) acc += exp(a.p_pi[k] - mx);
    const double sum = lt_block_sum(acc, s);
    for (int k = threadIdx.x; k < LD; k += 256) {
      const double p = k < S ? exp(a.p_pi[k] - mx) / sum : 0.0;
      sd[k] = k < S ? sqrt(p) : 1.0;
      if (k == i) {
        a.pi[i] = p;
        a.dsq[i] = sd[k];
      }
    }
    __syncthreads();
  }
  const bool pow2 = a.Q_pow2 && ((epoch & (epoch + 1)) == 0);  // epochs 1, 2, 4, ... (1-based)
  int pidx = 0;
  for (int e1 = epoch + 1; e1 > 1; e1 >>= 1) ++pidx;
  double *Qp = (pow2 && pidx < a.n_pow2) ? a.Q_pow2 + (size_t)pidx * S * S + (size_t)i * S : nullptr;
  double *Ql = a.Q_last + (size_t)i * S;
  const double di = sd[i], inv_di = 1.0 / di;
  double acc = 0.0;
  for (int j = threadIdx.x; j < LD; j += 256) {
    double r = 0.0;
    if (j < S && j != i) {
      const int lo_ = min(i, j), hi_ = max(i, j);
      const size_t k = (size_t)lo_ * S - (size_t)lo_ * (lo_ + 1) / 2 + (hi_ - lo_ - 1);
      r = softplus_t(a.p_up[k]) * (a.mask ? a.mask[(size_t)i * S + j] : 1.0);
      const double rd = r * sd[j];
      acc += rd;
      const double q = rd * inv_di;
      Ql[j] = q;
      if (Qp) Qp[j] = q;
    }
    if (j != i) Arow[j] = r;
  }
  const double tot = lt_block_sum(acc, s);
  if (threadIdx.x == 0) {
    const double aii = -tot * inv_di;
    Arow[i] = aii;
    Ql[i] = aii;
    if (Qp) Qp[i] = aii;
    if (a.sig_cur) atomicMax(a.sig_cur, (unsigned long long)__double_as_longlong(fabs(aii)));
  }
}

// one wave per k
__global__ __launch_bounds__(256) void lt_gd(LargeTrain a) {
  const int S = a.S, LD = a.LD;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= S) return;
  double acc = 0.0;
  for (int i = lane; i < S; i += 64)
    if (i != k) acc = fma(a.G[(size_t)i * LD + i] / a.dsq[i], a.A[(size_t)k * LD + i], acc);
  acc = wave_sum(acc);
  if (lane == 0)
    a.gd[k] = -a.dsq[k] * acc - a.G[(size_t)k * LD + k] * a.A[(size_t)k * LD + k] - a.dirsum[k] * a.inv_n;
}

// The parameter step, ONE launch: workgroups 0 .. S-1 take the rows of the upper-diagonal logits (and copy Q_last -> Q_best
// when this epoch improved), workgroup S the stationary logits, the loss curve and the best-loss word.  Every workgroup decides
// "improved" by itself from the loss and the best loss BEFORE this epoch, state[epoch & 1]; workgroup S leaves the best loss
// after it in state[(epoch + 1) & 1] (two words in turn, so that no workgroup can read the updated one).
__global__ __launch_bounds__(256) void lt_step(LargeTrain a, int epoch, double bc1, double bc2s) {
  __shared__ double s[256];
  const int S = a.S, LD = a.LD, i = blockIdx.x;
  const double loss = *a.loss, best = a.state[epoch & 1];
  const bool better = epoch == 0 || loss < best;  // strict <, first iterate always taken (trainer.py:179)
  if (i == S) {
    if (threadIdx.x == 0) {
      a.loss_curve[epoch - a.epoch0] = loss;   // (the curve of THIS call)
      if (a.time_curve) a.time_curve[epoch - a.epoch0] = (double)__builtin_amdgcn_s_memrealtime();
      a.state[(epoch + 1) & 1] = better ? loss : best;
    }
    double acc = 0.0;
    for (int k = threadIdx.x; k < S; k += 256) acc += a.gd[k];
    const double tot = lt_block_sum(acc, s);
    for (int k = threadIdx.x; k < S; k += 256) {
      const double g = 0.5 * (a.gd[k] - a.pi[k] * tot);
      adam_update(a.p_pi[k], a.m_pi[k], a.v_pi[k], g, a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s, a.do_adam);
    }
    return;
  }
  if (better)  // this epoch's (pre-step) Q is the best so far
    for (int j = threadIdx.x; j < S; j += 256) a.Q_best[(size_t)i * S + j] = a.Q_last[(size_t)i * S + j];
  const double di = a.dsq[i], gii = a.G[(size_t)i * LD + i];
  const size_t kbase = (size_t)i * S - (size_t)i * (i + 1) / 2;
  for (int j = i + 1 + threadIdx.x; j < S; j += 256) {
    const size_t k = kbase + (j - i - 1);
    const double mk = a.mask ? a.mask[(size_t)i * S + j] : 1.0;
    const double dj = a.dsq[j];
    const double dR_ab = mk * (a.G[(size_t)i * LD + j] - gii * dj / di);
    const double dR_ba = mk * (a.G[(size_t)j * LD + i] - a.G[(size_t)j * LD + j] * di / dj);
    const double g = sigmoid_t(a.p_up[k]) * (dR_ab + dR_ba);
    adam_update(a.p_up[k], a.m_up[k], a.v_up[k], g, a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s, a.do_adam);
  }
}
