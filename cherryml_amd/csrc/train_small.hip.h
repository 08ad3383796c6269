// Fused optimiser for small state spaces (S <= 32): one workgroup per site runs
// ALL epochs of the reference's loop on the device,
//
//   theta -> A, pi        (rate.py:167-188  /  _cherryml_vectorized.py:242-262)
//   eigh, bank: loss, dL/dA                      (small_site_eval)
//   dL/dtheta              (chain rule through the parameterisation, in closed form)
//   best-iterate / snapshot bookkeeping          (trainer.py:179-184, vectorized :366-372)
//   Adam or SGD step       (torch.optim.Adam defaults: betas .9/.999, eps 1e-8)
//
// so an LG-sized problem (launch-latency bound: 0.4 MB, 6 MFLOP per epoch) costs one
// kernel launch for the whole optimisation instead of ~40 launches per epoch.
//
// Parameterisations (KIND):
//   0 pande_reversible: up[S(S-1)/2] (row-major upper triangle), log_pi[S], mask[S][S]
//       R_ij = softplus(up_k) mask_ij (symmetric), pi = softmax(log_pi), d = sqrt(pi)
//   1 SiteRM: Theta[S][S] (full), theta[S]:  R_ij = softplus(Theta_ij + Theta_ji), i != j
//   A_ij = R_ij (i != j),  A_ii = -sum_j R_ij d_j / d_i,  Q_ij = R_ij d_j / d_i, Q_ii = A_ii.
// Gradient, given G = dL/dA (free matrix) and the direct term of the loss in log d:
//   dR_ij   = mask_ij (G_ij - G_ii d_j / d_i)
//   dup_k   = sigmoid(up_k) (dR_ij + dR_ji)            [SiteRM: same for Theta_ij and Theta_ji]
//   dld_k   = -d_k sum_{i != k} G_ii R_ik / d_i - G_kk A_kk - (colsum_k - rowsum_k) / n
//   dlogpi_k = (dld_k - pi_k sum_m dld_m) / 2
#pragma once
#include "small_bank.hip.h"

struct TrainArgs {
  int S, L, B, E, kind, do_adam, n_pow2;
  const double *t, *Ct, *inv_n, *dirsum;
  double *p_pi, *p_up;              // parameters  [L][S], [L][NUP]
  double *m_pi, *v_pi, *m_up, *v_up;  // Adam moments (zero initialised)
  const double *mask;               // [S][S] or null
  double lr, beta1, beta2, eps;
  double *loss_curve;               // [E][L]
  double *Q_best, *Q_last;          // [L][S][S]
  double *Q_pow2;                   // [n_pow2][S][S] (site 0) or null
};

__device__ __forceinline__ double softplus_t(double x) {  // torch: beta 1, threshold 20
  return x > 20.0 ? x : log1p(exp(x));
}
__device__ __forceinline__ double sigmoid_t(double x) { return 1.0 / (1.0 + exp(-x)); }

__device__ __forceinline__ void adam_update(double &p, double &m, double &v, double g, double lr,
                                            double b1, double b2, double eps, double bc1,
                                            double bc2_sqrt, int do_adam) {
  if (do_adam) {
    m = fma(1.0 - b1, g - m, m);          // exp_avg.lerp_(grad, 1 - beta1)
    v = fma((1.0 - b2) * g, g, b2 * v);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const double denom = sqrt(v) / bc2_sqrt + eps;
    p -= (lr / bc1) * (m / denom);
  } else {
    p -= lr * g;
  }
}

template <int NT, int KS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void small_train_kernel(TrainArgs a) {
  extern __shared__ double lds[];
  using LD = SmallLds<NW>;
  double *sA = lds + LD::A, *sG = lds + LD::G, *sD = lds + LD::D;
  double *sPi = lds + LD::TOTAL;       // [32]
  double *sGd = sPi + 32;              // [32] dL/d log d
  double *sFlag = sGd + 32;            // [2]  best loss, improved flag
  const int l = blockIdx.x, S = a.S, B = a.B, tid = threadIdx.x;
  const int NUP = a.kind == 0 ? S * (S - 1) / 2 : S * S;
  double *p_pi = a.p_pi + (size_t)l * S, *p_up = a.p_up + (size_t)l * NUP;
  double *m_pi = a.m_pi + (size_t)l * S, *v_pi = a.v_pi + (size_t)l * S;
  double *m_up = a.m_up + (size_t)l * NUP, *v_up = a.v_up + (size_t)l * NUP;
  double *Qlast = a.Q_last + (size_t)l * S * S, *Qbest = a.Q_best + (size_t)l * S * S;
  const size_t lb = (size_t)l * B;
  const double inv_n = a.inv_n[l];
  const double *dirsum = a.dirsum + (size_t)l * S;
  if (tid == 0) sFlag[0] = INFINITY;
  double pow_b1 = 1.0, pow_b2 = 1.0;

  for (int epoch = 0; epoch < a.E; ++epoch) {
    // ---- pi = softmax(log_pi), d = sqrt(pi) ------------------------------------
    if (tid < 64) {
      const double x = tid < S ? p_pi[tid] : -INFINITY;
      const double mx = wave_max(x);
      const double e = tid < S ? exp(x - mx) : 0.0;
      const double sum = wave_sum(e);
      if (tid < 32) {
        const double pk = tid < S ? e / sum : 1.0;
        sPi[tid] = pk;
        sD[tid] = sqrt(pk);
      }
    }
    __syncthreads();
    // ---- off-diagonal of A (= R) ---------------------------------------------------
    for (int e = tid; e < S * S; e += blockDim.x) {
      const int i = e / S, j = e - i * S;
      double r = 0.0;
      if (i != j) {
        if (a.kind == 0) {
          const int lo_ = min(i, j), hi_ = max(i, j);
          const int k = lo_ * S - lo_ * (lo_ + 1) / 2 + (hi_ - lo_ - 1);
          r = softplus_t(p_up[k]) * (a.mask ? a.mask[e] : 1.0);
        } else {
          r = softplus_t(p_up[i * S + j] + p_up[j * S + i]);
        }
      }
      sA[i * CB_LS + j] = r;
    }
    __syncthreads();
    if (tid < S) {
      double acc = 0.0;
      for (int j = 0; j < S; ++j)
        if (j != tid) acc = fma(sA[tid * CB_LS + j], sD[j], acc);
      sA[tid * CB_LS + tid] = -acc / sD[tid];
    }
    __syncthreads();
    // ---- Q of this epoch (pre-step) -> Q_last; snapshots at epochs 1, 2, 4, ... -------
    const bool pow2 = a.Q_pow2 && l == 0 && ((epoch & (epoch + 1)) == 0);
    int pidx = 0;
    if (pow2) {
      int e1 = epoch + 1;
      while (e1 > 1) {
        e1 >>= 1;
        ++pidx;
      }
    }
    for (int e = tid; e < S * S; e += blockDim.x) {
      const int i = e / S, j = e - i * S;
      const double q = (i == j) ? sA[i * CB_LS + i] : sA[i * CB_LS + j] * sD[j] / sD[i];
      Qlast[e] = q;
      if (pow2 && pidx < a.n_pow2) a.Q_pow2[(size_t)pidx * S * S + e] = q;
    }
    // ---- loss and dL/dA -------------------------------------------------------------------
    // sV still holds the previous epoch's eigenvectors (zero padded): warm start
    small_site_eval<NT, KS, NW, SMALL_LOSSGRAD>(lds, S, B, a.t + lb, a.Ct + lb * S * S, inv_n,
                                                dirsum, nullptr, true, nullptr, epoch > 0);
    // (ends with a barrier: sG = dA, sA = A, LOSSTOT = loss)
    const double loss = lds[LD::LOSSTOT];
    if (tid == 0) {
      a.loss_curve[(size_t)epoch * a.L + l] = loss;
      const bool better = loss < sFlag[0];  // strict <, as trainer.py:179
      sFlag[1] = better ? 1.0 : 0.0;
      if (better) sFlag[0] = loss;
    }
    // ---- dL/d log d -----------------------------------------------------------------------------
    if (tid < S) {
      const int k = tid;
      double acc = 0.0;
      for (int i = 0; i < S; ++i)
        if (i != k) acc = fma(sG[i * CB_LS + i] * sA[i * CB_LS + k], 1.0 / sD[i], acc);
      sGd[k] = -sD[k] * acc - sG[k * CB_LS + k] * sA[k * CB_LS + k] - dirsum[k] * inv_n;
    }
    __syncthreads();
    if (sFlag[1] != 0.0)
      for (int e = tid; e < S * S; e += blockDim.x) Qbest[e] = Qlast[e];  // own writes: visible
    // ---- parameter gradients + optimiser step ---------------------------------------------------
    pow_b1 *= a.beta1;
    pow_b2 *= a.beta2;
    const double bc1 = 1.0 - pow_b1, bc2s = sqrt(1.0 - pow_b2);
    if (tid < S) {
      double tot = 0.0;
      for (int m = 0; m < S; ++m) tot += sGd[m];
      const double g = 0.5 * (sGd[tid] - sPi[tid] * tot);
      adam_update(p_pi[tid], m_pi[tid], v_pi[tid], g, a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s,
                  a.do_adam);
    }
    for (int k = tid; k < NUP; k += blockDim.x) {
      int i, j;
      if (a.kind == 0) {  // k-th entry of the row-major upper triangle
        i = 0;
        int rem = k;
        while (rem >= S - 1 - i) {
          rem -= S - 1 - i;
          ++i;
        }
        j = i + 1 + rem;
      } else {
        i = k / S;
        j = k - i * S;
      }
      double g = 0.0;
      if (i != j) {
        const int lo_ = min(i, j), hi_ = max(i, j);
        const double mk = (a.kind == 0 && a.mask) ? a.mask[lo_ * S + hi_] : 1.0;
        const double dR_ab = mk * (sG[lo_ * CB_LS + hi_] - sG[lo_ * CB_LS + lo_] * sD[hi_] / sD[lo_]);
        const double dR_ba = mk * (sG[hi_ * CB_LS + lo_] - sG[hi_ * CB_LS + hi_] * sD[lo_] / sD[hi_]);
        const double x = a.kind == 0 ? p_up[k] : p_up[i * S + j] + p_up[j * S + i];
        g = sigmoid_t(x) * (dR_ab + dR_ba);
      }
      // SiteRM: Theta_ij and Theta_ji get the same gradient, computed from the pre-step sum
      // Theta_ij + Theta_ji; the partner's update must not race with that read, so the
      // SiteRM step is a second, barrier-separated phase.
      if (a.kind == 0) {
        adam_update(p_up[k], m_up[k], v_up[k], g, a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s, a.do_adam);
      } else {
        (lds + LD::RED)[k] = g;  // reduction scratch (>= 1024 doubles) is free here
      }
    }
    if (a.kind != 0) {
      __syncthreads();
      for (int k = tid; k < NUP; k += blockDim.x)
        adam_update(p_up[k], m_up[k], v_up[k], (lds + LD::RED)[k], a.lr, a.beta1, a.beta2, a.eps, bc1,
                    bc2s, a.do_adam);
    }
    __threadfence_block();
    __syncthreads();
  }
}
