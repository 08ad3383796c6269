// General (non-reversible Q) path for S <= 32: scaling and squaring with a
// degree-18 Taylor polynomial and its exact reverse-mode adjoint -- the
// algorithm class of torch.matrix_exp, which is what the reference runs
// (trainer.py:170-172,186) -- for rate matrices that are not reversible: the
// reference's parameterisation under a non-symmetric mask (rate.py:182) and its
// "default" / "pande" / "stationary" modes.
//
//   X = t Q / 2^s,  s = max(0, ceil(log2 |tQ|_1))          (|X|_1 <= 1: 1/19! < 1e-17)
//   H_18 = I + X/18,  H_k = I + X H_{k+1} / k,  E_0 = H_1   (Horner)
//   E_{j+1} = E_j^2,  P = E_s
//   loss -= <C, log P>;   Pbar = -C / P / n
//   Ebar_j = E_j^T Ebar_{j+1} + Ebar_{j+1} E_j^T
//   Hbar_1 = Ebar_0;  Xbar += Hbar_k H_{k+1}^T / k;  Hbar_{k+1} = X^T Hbar_k / k;  Xbar += Hbar_18/18
//   Qbar += (t / 2^s) Xbar
//
// One workgroup per site, buckets dealt to the waves; every wave owns a stack of
// 32x32 (zero padded) matrices in a global scratch buffer (L2 resident) and all
// products run on v_mfma_f64_16x16x4_f64.  This path is ~50x more arithmetic than
// the spectral one and is not tuned: it exists for completeness and parity.
#pragma once
#include "common.hip.h"

#define GN_MAXS 40                 // max squarings (|tQ|_1 < 2^40)
#define GN_DEG 18
#define GN_SLOTS (GN_DEG + GN_MAXS + 8)
#define GN_MAT 1024                // doubles per 32x32 matrix

struct GeneralArgs {
  int S, L, B;             // B = bucket stride
  const int *nlive;        // [L] buckets to visit (live first) or null = B
  const double *t, *Ct, *inv_n, *Q;
  double *loss, *dQ, *P;   // P: expm mode (or null)
  double *scratch;         // [L][NW][GN_SLOTS][1024]
  double *partial;         // [L][NW][1024 + 1]  per-wave Qbar and loss
};

__device__ __forceinline__ void gn_fence() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}

// D = alpha * op(A) op(B) [+ I on the first S diagonal entries]  or  D += alpha * op(A) op(B)
__device__ __forceinline__ void wave_mm32(double *__restrict__ D, const double *A, bool tA,
                                          const double *B, bool tB, double alpha, bool addI,
                                          bool accumulate, int S) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = 4 * s + hi;
    double av[2], bv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = 16 * x + lo;
      av[x] = tA ? A[k * 32 + i] : A[i * 32 + k];
      bv[x] = tB ? B[i * 32 + k] : B[k * 32 + i];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = mfma_f64(av[x], bv[y], acc[x][y]);
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * x + hi + 4 * r, col = 16 * y + lo;
        double v = alpha * acc[x][y][r];
        if (addI && row == col && row < S) v += 1.0;
        if (accumulate) v += D[row * 32 + col];
        D[row * 32 + col] = v;
      }
  gn_fence();
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void general_bank_kernel(GeneralArgs a) {
  __shared__ double sNorm;
  const int l = blockIdx.x, S = a.S, B = a.B;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double *Q = a.Q + (size_t)l * S * S;
  double *st = a.scratch + ((size_t)l * NW + wave) * GN_SLOTS * GN_MAT;
  // slot map
  double *mQ = st;                    // Q padded
  double *mX = st + 1 * GN_MAT;
  double *mQbar = st + 2 * GN_MAT;
  double *mXbar = st + 3 * GN_MAT;
  double *mT0 = st + 4 * GN_MAT, *mT1 = st + 5 * GN_MAT;
  double *mH = st + 6 * GN_MAT;       // H_1 .. H_18 at mH + (k-1)*MAT ; E_j = slot after
  double *mE = mH + GN_DEG * GN_MAT;  // E_1 .. E_s  (E_0 = H_1)

  // |Q|_1 (max column sum of |Q|), once per site
  if (threadIdx.x < 64) {
    double cs = 0.0;
    if (lane < S)
      for (int i = 0; i < S; ++i) cs += fabs(Q[i * S + lane]);
    cs = wave_max(cs);
    if (lane == 0) sNorm = cs;
  }
  for (int e = lane; e < GN_MAT; e += 64) {
    const int i = e >> 5, j = e & 31;
    mQ[e] = (i < S && j < S) ? Q[i * S + j] : 0.0;
    mQbar[e] = 0.0;
  }
  __syncthreads();
  const double qnorm = sNorm;
  const double inv_n = a.inv_n[l];
  double lossacc = 0.0;

  const int Bn = a.nlive ? a.nlive[l] : B;
  for (int b = wave; b < Bn; b += NW) {
    const size_t lb = (size_t)l * B + b;
    const double tb = a.t[lb];
    int s = 0;
    {
      const double nrm = tb * qnorm;
      if (nrm > 1.0) s = min(GN_MAXS, (int)ceil(log2(nrm)));
    }
    const double scale = ldexp(tb, -s);
    for (int e = lane; e < GN_MAT; e += 64) {
      mX[e] = scale * mQ[e];
      mXbar[e] = 0.0;
    }
    gn_fence();
    // ---- Horner: H_18 = I + X/18 ; H_k = I + X H_{k+1} / k -----------------------
    double *H18 = mH + (GN_DEG - 1) * GN_MAT;
    for (int e = lane; e < GN_MAT; e += 64) {
      const int i = e >> 5, j = e & 31;
      H18[e] = mX[e] * (1.0 / GN_DEG) + ((i == j && i < S) ? 1.0 : 0.0);
    }
    gn_fence();
    for (int k = GN_DEG - 1; k >= 1; --k)
      wave_mm32(mH + (k - 1) * GN_MAT, mX, false, mH + k * GN_MAT, false, 1.0 / k, true, false, S);
    // ---- squarings: E_0 = H_1, E_{j+1} = E_j^2 ----------------------------------------
    const double *Ecur = mH;
    for (int j = 0; j < s; ++j) {
      double *En = mE + (size_t)j * GN_MAT;
      wave_mm32(En, Ecur, false, Ecur, false, 1.0, false, false, S);
      Ecur = En;
    }
    // ---- P = E_s: loss and Pbar (into mT0) -------------------------------------------
    const double *Ctb = a.Ct + lb * S * S;
    for (int e = lane; e < GN_MAT; e += 64) {
      const int i = e >> 5, j = e & 31;
      double pb = 0.0;
      if (i < S && j < S) {
        const double pv = Ecur[e];
        if (a.P) {
          a.P[lb * S * S + (size_t)i * S + j] = pv;
        } else {
          const double c = Ctb[j * S + i];  // stored transposed
          if (c != 0.0) {
            lossacc = fma(-c, log(pv), lossacc);
            pb = -c * inv_n / pv;
          }
        }
      }
      mT0[e] = pb;
    }
    gn_fence();
    if (a.P || a.dQ == nullptr) continue;
    // ---- backward through the squarings: Ebar_j = E_j^T Ebar_{j+1} + Ebar_{j+1} E_j^T -----
    double *Ebar = mT0, *Etmp = mT1;
    for (int j = s - 1; j >= 0; --j) {
      const double *Ej = (j == 0) ? mH : mE + (size_t)(j - 1) * GN_MAT;
      wave_mm32(Etmp, Ej, true, Ebar, false, 1.0, false, false, S);
      wave_mm32(Etmp, Ebar, false, Ej, true, 1.0, false, true, S);
      double *sw = Ebar;
      Ebar = Etmp;
      Etmp = sw;
    }
    // ---- backward through Horner (Hbar_1 = Ebar_0) ----------------------------------------
    double *Hbar = Ebar, *Hnext = Etmp;
    for (int k = 1; k <= GN_DEG - 1; ++k) {
      // Xbar += Hbar_k H_{k+1}^T / k ; Hbar_{k+1} = X^T Hbar_k / k
      wave_mm32(mXbar, Hbar, false, mH + k * GN_MAT, true, 1.0 / k, false, true, S);
      wave_mm32(Hnext, mX, true, Hbar, false, 1.0 / k, false, false, S);
      double *sw = Hbar;
      Hbar = Hnext;
      Hnext = sw;
    }
    for (int e = lane; e < GN_MAT; e += 64)
      mQbar[e] = fma(scale, mXbar[e] + Hbar[e] * (1.0 / GN_DEG), mQbar[e]);
    gn_fence();
  }
  if (a.P) return;
  // per-wave partials -> global; wave 0 of the workgroup sums them in a fixed order
  double *part = a.partial + ((size_t)l * NW + wave) * (GN_MAT + 1);
  lossacc = wave_sum(lossacc);
  for (int e = lane; e < GN_MAT; e += 64) part[e] = mQbar[e];
  if (lane == 0) part[GN_MAT] = lossacc;
  __threadfence_block();
  __syncthreads();
  const double *p0 = a.partial + (size_t)l * NW * (GN_MAT + 1);
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < NW; ++w) tot += p0[(size_t)w * (GN_MAT + 1) + GN_MAT];
    a.loss[l] = tot * inv_n;
  }
  if (a.dQ)
    for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
      const int i = e / S, j = e - i * S;
      double acc = 0.0;
      for (int w = 0; w < NW; ++w) acc += p0[(size_t)w * (GN_MAT + 1) + i * 32 + j];
      a.dQ[(size_t)l * S * S + e] = acc;
    }
}
