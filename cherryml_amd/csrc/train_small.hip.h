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
  const int *nlive;                 // [L] live buckets per site (stored first); B = stride
  const double *Cq;                 // S <= 20: counts in quad order (SmallArgs::Cq)
  int nq;
  const double *t, *Ct, *inv_n, *dirsum;
  double *p_pi, *p_up;              // parameters  [L][S], [L][NUP]
  double *m_pi, *v_pi, *m_up, *v_up;  // Adam moments (zero initialised)
  const double *mask;               // [S][S] or null
  double lr, beta1, beta2, eps;
  double *loss_curve;               // [E][L]
  double *time_curve;               // [E] or null: the 100 MHz wall clock when site 0 finished each epoch (cb_train_epoch_times)
  double *Q_best, *Q_last;          // [L][S][S]
  double *Q_pow2;                   // [n_pow2][S][S] (site 0) or null
  int sym;                          // all count matrices symmetric: sp_bank's symmetric form, sp_finish mirrors M
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

// theta -> pi, d (sPi, sD), A (sA); Q of this epoch -> Q_last (+ snapshot at epochs 1, 2, 4 ...)
__device__ __forceinline__ void tr_build(const TrainArgs &a, int l, int epoch, double *sA, double *sD,
                                         double *sPi) {
  const int S = a.S, tid = threadIdx.x;
  const int NUP = a.kind == 0 ? S * (S - 1) / 2 : S * S;
  const double *p_pi = a.p_pi + (size_t)l * S, *p_up = a.p_up + (size_t)l * NUP;
  double *Qlast = a.Q_last + (size_t)l * S * S;
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
}

// given loss, sG = dL/dA, sA = A, sD, sPi: bookkeeping (loss curve, best iterate), parameter
// gradients, optimiser step.  best[0] = best loss so far (+inf initially), lives in `best`
// (LDS or global).  sGd: 32 doubles, sRed: >= S*S doubles of scratch.  Ends with a barrier.
__device__ __forceinline__ void tr_update(const TrainArgs &a, int l, int epoch, double loss,
                                          double bc1, double bc2s, const double *sA,
                                          const double *sG, const double *sD, const double *sPi,
                                          double *sGd, double *sFlag, double *sRed, double *best) {
  const int S = a.S, tid = threadIdx.x;
  const int NUP = a.kind == 0 ? S * (S - 1) / 2 : S * S;
  double *p_pi = a.p_pi + (size_t)l * S, *p_up = a.p_up + (size_t)l * NUP;
  double *m_pi = a.m_pi + (size_t)l * S, *v_pi = a.v_pi + (size_t)l * S;
  double *m_up = a.m_up + (size_t)l * NUP, *v_up = a.v_up + (size_t)l * NUP;
  double *Qlast = a.Q_last + (size_t)l * S * S, *Qbest = a.Q_best + (size_t)l * S * S;
  const double inv_n = a.inv_n[l];
  const double *dirsum = a.dirsum + (size_t)l * S;
  if (tid == 0) {
    a.loss_curve[(size_t)epoch * a.L + l] = loss;
    if (l == 0 && a.time_curve) a.time_curve[epoch] = (double)__builtin_amdgcn_s_memrealtime();
    // strict <, as trainer.py:179; there the first iterate is always taken (`best_loss is None`), also
    // when its loss is NaN; the SiteRM loop starts from +inf instead (_cherryml_vectorized.py:366)
    const bool better = (a.kind == 0 && epoch == 0) || loss < *best;
    sFlag[0] = better ? 1.0 : 0.0;
    if (better) *best = loss;
  }
  if (tid < S) {
    const int k = tid;
    double acc = 0.0;
    for (int i = 0; i < S; ++i)
      if (i != k) acc = fma(sG[i * CB_LS + i] * sA[i * CB_LS + k], 1.0 / sD[i], acc);
    sGd[k] = -sD[k] * acc - sG[k * CB_LS + k] * sA[k * CB_LS + k] - dirsum[k] * inv_n;
  }
  __syncthreads();
  if (sFlag[0] != 0.0)
    for (int e = tid; e < S * S; e += blockDim.x) Qbest[e] = Qlast[e];
  if (tid < S) {
    double tot = 0.0;
    for (int m = 0; m < S; ++m) tot += sGd[m];
    const double g = 0.5 * (sGd[tid] - sPi[tid] * tot);
    adam_update(p_pi[tid], m_pi[tid], v_pi[tid], g, a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s, a.do_adam);
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
    if (a.kind == 0) adam_update(p_up[k], m_up[k], v_up[k], g, a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s, a.do_adam);
    else sRed[k] = g;
  }
  if (a.kind != 0) {
    __syncthreads();
    for (int k = tid; k < NUP; k += blockDim.x)
      adam_update(p_up[k], m_up[k], v_up[k], sRed[k], a.lr, a.beta1, a.beta2, a.eps, bc1, bc2s, a.do_adam);
  }
  __threadfence_block();
  __syncthreads();
}

// The per-epoch evaluation behind a real call: inlined into the epoch loop, the optimiser state
// that lives across it pushed the bucket loop of the 20-state instance to 256 VGPRs + 81 spilled
// registers (~35 scratch accesses per bucket); as a callee it is allocated on its own (the plain
// bank kernel's allocation), and the call costs one jump per epoch.
template <int NT, int KS, int NW>
__device__ __attribute__((noinline)) void small_site_eval_call(double *lds, int S, int B, const double *t_l,
                                                               const double *Ct_l, double inv_n,
                                                               const double *dirsum_l, bool warm, const double *Cq_l) {
  small_site_eval<NT, KS, NW, SMALL_LOSSGRAD>(lds, S, B, t_l, Ct_l, inv_n, dirsum_l, nullptr, true, nullptr, warm, Cq_l);
}

template <int NT, int KS, int NW>
__global__ __launch_bounds__(NW * 64, (KS >= 6 && NW == 4) ? 1 : CB_SMALL_MIN_WGS) void small_train_kernel(TrainArgs a) {
  extern __shared__ double lds[];
  using LD = SmallLds<NW>;
  double *sA = lds + LD::A, *sG = lds + LD::G, *sD = lds + LD::D;
  double *sPi = lds + LD::TOTAL;       // [32]
  double *sGd = sPi + 32;              // [32] dL/d log d
  double *sFlag = sGd + 32;            // [2]  improved flag, best loss
  const int l = blockIdx.x, S = a.S, B = a.B;
  const size_t lb = (size_t)l * B;
  if (threadIdx.x == 0) sFlag[1] = INFINITY;
  double pow_b1 = 1.0, pow_b2 = 1.0;
  for (int epoch = 0; epoch < a.E; ++epoch) {
    tr_build(a, l, epoch, sA, sD, sPi);
    // sV still holds the previous epoch's eigenvectors (zero padded): warm start
    small_site_eval_call<NT, KS, NW>(lds, S, a.nlive[l], a.t + lb, a.Ct + lb * S * S, a.inv_n[l],
                                     a.dirsum + (size_t)l * S, epoch > 0,
                                     a.Cq ? a.Cq + (size_t)l * a.nq * (KS * KS * 64) : nullptr);
    // (ends with a barrier: sG = dA, sA = A, LOSSTOT = loss)
    pow_b1 *= a.beta1;
    pow_b2 *= a.beta2;
    tr_update(a, l, epoch, lds[LD::LOSSTOT], 1.0 - pow_b1, sqrt(1.0 - pow_b2), sA, sG, sD, sPi, sGd,
              sFlag, lds + LD::RED, sFlag + 1);
  }
}

// ---------------------------------------------------------------------------------------
// L = 1 (LG-sized single bank): the epoch split over the chip.  A single workgroup holds at
// most 8 waves of this register-heavy code, so 129 buckets cost 17 sequential buckets per
// wave; instead three small kernels per epoch, enqueued back to back (no host sync):
//   lg_prepare : theta -> A, eigh (warm), frames to global            1 workgroup
//   lg_bank    : one bucket per wavefront, partial M / loss to global  ceil(B/4) workgroups
//   lg_finish  : sum the partials (fixed order), dA, gradients, Adam   1 workgroup
struct LgSplit {
  double *frames;  // A[32*33] | V[32*33] | lam[32] | d[32] | pi[32]
  double *Mpart;   // [B][1024]
  double *lpart;   // [B]
  double *best;    // [1] best loss so far
};
#define LGS_A 0
#define LGS_V (32 * CB_LS)
#define LGS_LAM (2 * 32 * CB_LS)
#define LGS_D (LGS_LAM + 32)
#define LGS_PI (LGS_D + 32)
#define LGS_TOTAL (LGS_PI + 32)

__global__ __launch_bounds__(256) void lg_prepare(TrainArgs a, LgSplit g, int epoch) {
  extern __shared__ double lds[];
  using LD = SmallLds<4>;
  double *sA = lds + LD::A, *sG = lds + LD::G, *sV = lds + LD::V, *sLam = lds + LD::LAM,
         *sD = lds + LD::D, *sPi = lds + LD::TOTAL;
  const int S = a.S, tid = threadIdx.x;
  if (epoch == 0 && tid == 0) *g.best = INFINITY;
  tr_build(a, 0, epoch, sA, sD, sPi);
  if (epoch > 0)  // previous eigenvectors: warm start
    for (int e = tid; e < 32 * CB_LS; e += 256) sV[e] = g.frames[LGS_V + e];
  __syncthreads();
  if (tid < 64) wave_eigh_rate(S, sA, sG, sV, sLam, CB_LS, epoch > 0);
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int k = e >> 5, i = e & 31;
    if (k >= S || i >= S) sV[k * CB_LS + i] = 0.0;
  }
  for (int k = S + tid; k < 32; k += 256) sLam[k] = 0.0;
  __syncthreads();
  for (int e = tid; e < 32 * CB_LS; e += 256) {
    g.frames[LGS_A + e] = sA[e];
    g.frames[LGS_V + e] = sV[e];
  }
  if (tid < 32) {
    g.frames[LGS_LAM + tid] = sLam[tid];
    g.frames[LGS_D + tid] = sD[tid];
    g.frames[LGS_PI + tid] = sPi[tid];
  }
}

template <int NT, int KS>
__global__ __launch_bounds__(256, 2) void lg_bank(TrainArgs a, LgSplit g) {
  extern __shared__ double lds[];
  using LD = SmallLds<4>;
  double *sA = lds + LD::A, *sV = lds + LD::V, *sLam = lds + LD::LAM, *sD = lds + LD::D;
  const int S = a.S, B = a.B, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  for (int e = tid; e < 32 * CB_LS; e += 256) {
    sA[e] = g.frames[LGS_A + e];
    sV[e] = g.frames[LGS_V + e];
  }
  if (tid < 32) {
    sLam[tid] = g.frames[LGS_LAM + tid];
    sD[tid] = g.frames[LGS_D + tid];
  }
  __syncthreads();
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  SmallFrags<NT, KS> f;
  load_frags<NT, KS>(f, sV, sLam, S);
  d4 M[NT][NT];
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) M[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  double rho = 0.0;
  for (int i = lane; i < S; i += 64) rho = fmax(rho, fabs(sA[i * CB_LS + i]));
  rho = 2.0 * wave_max(rho);
  double cval[NT][NT][4];
  load_counts<NT, KS>(cval, S, a.Ct + (size_t)b * S * S);
  double lossacc = 0.0;
  small_bucket<NT, KS, SMALL_LOSSGRAD>(f, S, a.t[b], cval, nullptr, a.inv_n[0], sA, sD, sV,
                                        lds + LD::TAB + wave * 384, sLam, rho, M, lossacc, nullptr);
  lossacc = wave_sum(lossacc);
  double *dst = g.Mpart + (size_t)b * 1024;
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((x * 2 + y) * 4 + r) * 64 + lane] = M[x][y][r];
  if (lane == 0) g.lpart[b] = lossacc;
}

template <int NT>
__global__ __launch_bounds__(256) void lg_finish(TrainArgs a, LgSplit g, int epoch, double bc1,
                                                 double bc2s) {
  extern __shared__ double lds[];
  using LD = SmallLds<4>;
  double *sA = lds + LD::A, *sG = lds + LD::G, *sV = lds + LD::V, *sD = lds + LD::D;
  double *sPi = lds + LD::TOTAL, *sGd = sPi + 32, *sFlag = sGd + 32;
  const int S = a.S, B = a.B, tid = threadIdx.x;
  for (int e = tid; e < 32 * CB_LS; e += 256) {
    sA[e] = g.frames[LGS_A + e];
    sV[e] = g.frames[LGS_V + e];
    sG[e] = 0.0;
  }
  if (tid < 32) {
    sD[tid] = g.frames[LGS_D + tid];
    sPi[tid] = g.frames[LGS_PI + tid];
  }
  __syncthreads();
  // M = sum over buckets in a fixed order; slot (x, y, r, lane) -> M[16x + hi + 4r][16y + lo].
  // Loads are issued 8 at a time (independent) -- one dependent load per iteration would
  // cost B global round trips.
  for (int e = tid; e < NT * NT * 256; e += 256) {
    const int slot = e >> 6, lane = e & 63;
    const int x = (slot >> 2) / NT, y = (slot >> 2) % NT, r = slot & 3;
    const double *src = g.Mpart + ((x * 2 + y) * 4 + r) * 64 + lane;
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= B; b += 8) {
      double v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(b + i) * 1024];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    for (; b < B; ++b) acc += src[(size_t)b * 1024];
    const int ra = 16 * x + (lane >> 4) + 4 * r, c = 16 * y + (lane & 15);
    sG[ra * CB_LS + c] = acc;
  }
  // loss partials through LDS (scratch RED), then one thread adds them in order
  double *sL = lds + LD::RED;
  for (int b = tid; b < B; b += 256) sL[b] = g.lpart[b];
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) tot += sL[b];
    double dir = 0.0;
    for (int k = 0; k < S; ++k) dir = fma(log(sD[k]), a.dirsum[k], dir);
    lds[LD::LOSSTOT] = (tot - dir) * a.inv_n[0];
  }
  __syncthreads();
  small_dA_from_M(S, sG, sV, lds + LD::RED);
  tr_update(a, 0, epoch, lds[LD::LOSSTOT], bc1, bc2s, sA, sG, sD, sPi, sGd, sFlag, lds + LD::RED, g.best);
}

// ---------------------------------------------------------------------------------------
// Site-parallel split trainer (S <= 24, any L, both parameterisations).  In the one-kernel trainer a
// workgroup owns a site for all epochs: every epoch its wave 0 spends ~70 us in the eigensolver
// (31 % of the epoch at 20 states) while the other waves wait, and its count streaming cannot
// overlap anybody else's serial phases.  Here an epoch is three launches over all sites, enqueued
// back to back without host synchronisation:
//   sp_prepare : theta -> A, eigh (warm), frames to global     1 workgroup / site, 26 KB of LDS, so
//                                                              six eigensolves overlap per CU
//   sp_bank    : 4x4-tile quads (small_quad), partial M, loss  (site, chunk) workgroups
//   sp_finish  : sum the partials (fixed order), dA, gradients, best iterate, Adam   1 workgroup / site
// State between the launches lives in HBM / L2 (17.7 KB of frames per site).
struct SpSplit {
  double *frames;  // [L][LGS_TOTAL]   A | V | lam | d | pi
  double *Mpart;   // [L][nchunk][576] partial M of a chunk, [tile][4 q + r]
  double *lpart;   // [L][nchunk]
  double *best;    // [L] best loss so far
  int nchunk, quads_per_chunk;
};
#define SP_ROWS 24   // S <= 24: frames of 24 rows (stride CB_LS) suffice
#define SPP_A 0
#define SPP_G (SP_ROWS * CB_LS)
#define SPP_V (2 * SP_ROWS * CB_LS)
#define SPP_LAM (3 * SP_ROWS * CB_LS)
#define SPP_D (SPP_LAM + 32)
#define SPP_PI (SPP_D + 32)
#define SPP_X (SPP_PI + 32)                     // first-order sweeps: X frame, then the diagonal of Gamma
#define SPP_DG (SPP_X + SP_ROWS * CB_LS)
#define SPP_TOTAL (SPP_DG + 32)

// one wavefront per workgroup: only the eigensolver's wave has work for most of the kernel, and
// 26 KB of LDS lets six sites overlap per CU.  Every epoch after the first is warm-started from the previous
// eigenvectors and solved by first-order sweeps on the matrix cores (jacobi_wave.hip.h: a solve late in an
// optimisation is ~8 20 x 20 products instead of two or three 19-round Jacobi sweeps)
// NTH = 64 for many sites (above); a few sites (one LG-sized bank) take 256 threads: theta -> A, the frame copies and
// the Q_last write are then spread over four waves (the single wave spent more time there than in the eigensolver)
#ifdef CB_SP_STAMPS   // build-time experiment: phase stamps of site 0 at epoch 300 into g.best[1 + i] (100 MHz ticks)
#define SP_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && epoch == 300) ((unsigned long long *)g.best)[1 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif
template <int NTH>
__device__ __forceinline__ void sp_prepare_body(const TrainArgs &a, const SpSplit &g, int epoch, double *lds, bool v_resident) {
  double *sA = lds + SPP_A, *sG = lds + SPP_G, *sV = lds + SPP_V, *sLam = lds + SPP_LAM, *sD = lds + SPP_D,
         *sPi = lds + SPP_PI;
  const int S = a.S, tid = threadIdx.x, l = blockIdx.x;
  double *fr = g.frames + (size_t)l * LGS_TOTAL;
  SP_STAMP(5);
  if (epoch == 0 && tid == 0) g.best[l] = INFINITY;
  if (epoch > 0 && !v_resident)  // previous eigenvectors: warm start (issued first: the loads fly while theta -> A is computed)
    for (int e = tid; e < SP_ROWS * CB_LS; e += NTH) sV[e] = fr[LGS_V + e];
  tr_build(a, l, epoch, sA, sD, sPi);
  // the 4 x 4-tile products of the warm solve read whole tiles: A is zero outside S x S (V already is)
  for (int e = tid; e < SP_ROWS * SP_ROWS; e += NTH) {
    const int i = e / SP_ROWS, j = e - i * SP_ROWS;
    if (i >= S || j >= S) sA[i * CB_LS + j] = 0.0;
  }
  __syncthreads();
  SP_STAMP(6);
  if (NTH == 64 || tid < 64) {
    if (epoch > 0) wave_eigh_rate_warm_mfma4(S, sA, sG, sV, lds + SPP_X, lds + SPP_DG, sLam, CB_LS);
    else wave_eigh_rate(S, sA, sG, sV, sLam, CB_LS, false);
  }
  __syncthreads();
  for (int e = tid; e < SP_ROWS * 32; e += NTH) {
    const int k = e >> 5, i = e & 31;
    if (k >= S || i >= S) sV[k * CB_LS + i] = 0.0;
  }
  for (int k = S + tid; k < 32; k += NTH) sLam[k] = 0.0;
  __syncthreads();
  SP_STAMP(7);
  for (int e = tid; e < SP_ROWS * CB_LS; e += NTH) {
    fr[LGS_A + e] = sA[e];
    fr[LGS_V + e] = sV[e];
  }
  if (tid < 32) {
    fr[LGS_LAM + tid] = sLam[tid];
    fr[LGS_D + tid] = sD[tid];
    fr[LGS_PI + tid] = sPi[tid];
  }
  SP_STAMP(8);
}

template <int NTH>
__global__ __launch_bounds__(NTH, 2) void sp_prepare(TrainArgs a, SpSplit g, int epoch) {
  extern __shared__ double lds[];
  sp_prepare_body<NTH>(a, g, epoch, lds, false);
}

// bank LDS: only the first 24 rows of the A / V frames are touched for S <= 20; Mw = [wave][tile][lane]
#define SPB_ROWS 24
#ifndef SPB_LS
#define SPB_LS 36      // row stride of sp_bank's OWN copy of the frames, SPB_TABS: distance of a wave's four spectral tables
#endif
#ifndef SPB_TABS
#define SPB_TABS 100   // (small_quad: both = 4 mod 32 -> conflict-free ds_read_b64 in every layout of the quad)
#endif
#define SPB_A 0
#define SPB_V (SPB_ROWS * SPB_LS)
#define SPB_LAM (2 * SPB_ROWS * SPB_LS)
#define SPB_TAB (SPB_LAM + 32)
#define SPB_MW (SPB_TAB + 4 * 4 * SPB_TABS)

// Workgroups per CU: THREE where the quad fits 168 registers once the B-layout tiles of U are re-read from LDS
// (small_quad's ULDS form) and M is accumulated in 16 slots per tile (44 KB of LDS per workgroup) -- up to 16
// states, and 20 states with symmetric counts (the SiteRM case); two elsewhere (the other forms spill 40 .. 83
// registers at 168).  CB_SPB_WGS=2 at build time restores two everywhere.
#ifndef CB_SPB_WGS
#define CB_SPB_WGS 3
#endif
__host__ __device__ constexpr bool spb_w3_ok(int TS, bool SYM) { return CB_SPB_WGS >= 3 && (TS <= 4 || (TS == 5 && SYM)); }
__host__ __device__ constexpr int spb_wgs(int TS, bool SYM, bool W3) { return (W3 && spb_w3_ok(TS, SYM)) ? 3 : 2; }
__host__ __device__ constexpr int spb_mws(int TS, bool SYM, bool W3) { return (TS <= 5 && spb_wgs(TS, SYM, W3) < 3) ? 1600 : 576; }   // M doubles per wave
// after Mw: 8 doubles of loss partials; Mw = 4 waves x spb_mws doubles ([tile][64] per wave: 1600, or [tile][16]: 576)
__host__ __device__ constexpr int spb_total(int TS, bool SYM, bool W3) { return SPB_MW + 4 * spb_mws(TS, SYM, W3) + 8 + 256; }   // + the log table
// W3: the three-workgroup form (many sites); the two-workgroup form keeps U's tiles in registers and is the faster one
// per quad, which is what counts when the grid does not fill the chip (one LG-sized bank)
template <int TS, bool SYM = false, bool W3 = false>   // SYM: all count matrices symmetric (small_quad's symmetric form)
__global__ __launch_bounds__(256, spb_wgs(TS, SYM, W3)) void sp_bank(TrainArgs a, SpSplit g) {
  extern __shared__ double lds[];
  double *sA = lds + SPB_A, *sV = lds + SPB_V, *sLam = lds + SPB_LAM;
  const int S = a.S, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, blk = (lane >> 2) & 3;
  const int l = blockIdx.x / g.nchunk, chunk = blockIdx.x - l * g.nchunk;
  const double *fr = g.frames + (size_t)l * LGS_TOTAL;
  for (int e = tid; e < SPB_ROWS * 32; e += 256) {   // the frames (row stride CB_LS in memory) re-strided
    const int k = e >> 5, i = e & 31;
    sA[k * SPB_LS + i] = fr[LGS_A + k * CB_LS + i];
    sV[k * SPB_LS + i] = fr[LGS_V + k * CB_LS + i];
  }
  if (tid < 32) sLam[tid] = fr[LGS_LAM + tid];
  constexpr bool LANEM = spb_mws(TS, SYM, W3) == 1600;   // per-lane M slots while they fit LDS (2 workgroups per CU)
  constexpr int MWS = spb_mws(TS, SYM, W3);             // doubles per wave
  constexpr int SPB_LOSS = SPB_MW + 4 * MWS;
  const double *ltab = lds + SPB_LOSS + 8;   // (16-byte aligned: every offset before it is even)
  fast_log_table_fill(lds + SPB_LOSS + 8, tid, 256);
  double *Mw = lds + SPB_MW + wave * MWS;
  for (int e = lane; e < (LANEM ? 64 : 16) * TS * TS; e += 64) Mw[e] = 0.0;
  __syncthreads();
  const int Bn = a.nlive[l], nquads = (Bn + 3) / 4;
  const size_t lb = (size_t)l * a.B;
  const double *t_l = a.t + lb, *Cq_l = a.Cq + (size_t)l * a.nq * (TS * TS * 64);
  const double inv_n = a.inv_n[l];
  double rho = 0.0;
  for (int i = lane; i < S; i += 64) rho = fmax(rho, fabs(sA[i * SPB_LS + i]));
  rho = 2.0 * wave_max(rho);
  const int q0 = chunk * g.quads_per_chunk, q1 = min(nquads, q0 + g.quads_per_chunk);
  double lossacc = 0.0;
  for (int qd = q0 + wave; qd < q1; qd += 4) {
    const int bucket = 4 * qd + blk;
    const double tb = bucket < Bn ? t_l[bucket] : 0.0;
    small_quad<TS, LANEM, SYM, (spb_wgs(TS, SYM, W3) >= 3), true, SPB_LS, SPB_TABS>(
        S, tb, Cq_l + (size_t)qd * (TS * TS * 64), inv_n, sA, sV, lds + SPB_TAB + wave * (4 * SPB_TABS), sLam, rho, Mw, lossacc, ltab);
  }
  lossacc = wave_sum(lossacc);
  if (lane == 0) lds[SPB_LOSS + wave] = lossacc;
  __syncthreads();
  double *dst = g.Mpart + ((size_t)l * g.nchunk + chunk) * 576;
  for (int e = tid; e < 16 * TS * TS; e += 256) {   // e = tile * 16 + 4 q + r: sum the (4 blocks of the) 4 waves, fixed order
    double tot = 0.0;
    if (LANEM) {
      const int slot = (e >> 4) * 64 + 16 * ((e >> 2) & 3) + (e & 3);
      for (int w = 0; w < 4; ++w) {
        const double *m = lds + SPB_MW + w * MWS + slot;
        tot += (m[0] + m[4]) + (m[8] + m[12]);
      }
    } else {
      for (int w = 0; w < 4; ++w) tot += lds[SPB_MW + w * MWS + e];
    }
    dst[e] = tot;
  }
  if (tid == 0)
    g.lpart[(size_t)l * g.nchunk + chunk] = (lds[SPB_LOSS] + lds[SPB_LOSS + 1]) + (lds[SPB_LOSS + 2] + lds[SPB_LOSS + 3]);
}

#define SPF_A 0
#define SPF_G (SP_ROWS * CB_LS)
#define SPF_V (2 * SP_ROWS * CB_LS)
#define SPF_D (3 * SP_ROWS * CB_LS)
#define SPF_PI (SPF_D + 32)
#define SPF_GD (SPF_PI + 32)
#define SPF_FLAG (SPF_GD + 32)
#define SPF_RED (SPF_FLAG + 8)
#define SPF_TOTAL (SPF_RED + 1056)

template <int TS>
__device__ __forceinline__ void sp_finish_body(const TrainArgs &a, const SpSplit &g, int epoch, double bc1, double bc2s, double *lds) {
  double *sA = lds + SPF_A, *sG = lds + SPF_G, *sV = lds + SPF_V, *sD = lds + SPF_D, *sPi = lds + SPF_PI,
         *sGd = lds + SPF_GD, *sFlag = lds + SPF_FLAG, *sRed = lds + SPF_RED;
  const int S = a.S, tid = threadIdx.x, l = blockIdx.x;
  const double *fr = g.frames + (size_t)l * LGS_TOTAL;
  SP_STAMP(0);
  for (int e = tid; e < SP_ROWS * CB_LS; e += 256) {
    sA[e] = fr[LGS_A + e];
    sV[e] = fr[LGS_V + e];
    sG[e] = 0.0;
  }
  if (tid < 32) {
    sD[tid] = fr[LGS_D + tid];
    sPi[tid] = fr[LGS_PI + tid];
  }
  __syncthreads();
  SP_STAMP(1);
  // M = sum over the chunks in a fixed order; entry e = tile * 16 + 4 i + j
  const double *Mp = g.Mpart + (size_t)l * g.nchunk * 576;
  for (int e = tid; e < 16 * TS * TS; e += 256) {
    double tot = 0.0;
#pragma unroll 4
    for (int c = 0; c < g.nchunk; ++c) tot += Mp[(size_t)c * 576 + e];
    const int tile = e >> 4, At = tile / TS, Ct = tile - At * TS;
    sG[(4 * At + ((e >> 2) & 3)) * CB_LS + 4 * Ct + (e & 3)] = tot;
  }
  if (a.sym) {   // sp_bank<TS, true> accumulated the tiles on or above the diagonal only: M is symmetric
    __syncthreads();
    for (int e = tid; e < 16 * TS * TS; e += 256) {
      const int tile = e >> 4, At = tile / TS, Ct = tile - At * TS, i = (e >> 2) & 3, j = e & 3;
      if (At > Ct) sG[(4 * At + i) * CB_LS + 4 * Ct + j] = sG[(4 * Ct + j) * CB_LS + 4 * At + i];
    }
  }
  if (tid < 64) {   // the direct pi term on S lanes (fixed-order tree sum) instead of S logarithms in a row on one thread
    const double dir = wave_sum(tid < S ? log(sD[tid]) * a.dirsum[(size_t)l * S + tid] : 0.0);
    if (tid == 0) {
      double tot = 0.0;
      for (int c = 0; c < g.nchunk; ++c) tot += g.lpart[(size_t)l * g.nchunk + c];
      sFlag[2] = (tot - dir) * a.inv_n[l];
    }
  }
  __syncthreads();
  SP_STAMP(2);
  small_dA_from_M(S, sG, sV, sRed);
  SP_STAMP(3);
  tr_update(a, l, epoch, sFlag[2], bc1, bc2s, sA, sG, sD, sPi, sGd, sFlag, sRed, g.best + l);
  SP_STAMP(4);
}

template <int TS>
__global__ __launch_bounds__(256) void sp_finish(TrainArgs a, SpSplit g, int epoch, double bc1, double bc2s) {
  extern __shared__ double lds[];
  sp_finish_body<TS>(a, g, epoch, bc1, bc2s, lds);
}

// Few sites (one LG-sized bank): sp_finish of epoch e - 1 and sp_prepare of epoch e in ONE launch -- the eigenvectors
// the finish just used are the warm start of the next solve and are already in LDS (the two frame layouts agree on A, G
// and V), a launch boundary (~3 us) and a frame read go away.  An epoch is then two launches: sp_step, sp_bank.
template <int TS>
__global__ __launch_bounds__(256, 2) void sp_step(TrainArgs a, SpSplit g, int epoch, double bc1_prev, double bc2s_prev) {
  extern __shared__ double lds[];
  static_assert(SPF_A == SPP_A && SPF_G == SPP_G && SPF_V == SPP_V, "finish and prepare share the A / G / V frames");
  sp_finish_body<TS>(a, g, epoch - 1, bc1_prev, bc2s_prev, lds);   // (ends with a barrier; parameters are updated)
  sp_prepare_body<256>(a, g, epoch, lds, true);
}
