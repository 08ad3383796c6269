// Block one-sided Jacobi eigensolver for the large path (S > 32, LD <= 1024).
//
// Same mathematics as jacobi_wave.hip.h -- Hestenes Jacobi on A' = A - sigma I,
// G = A' V -- but columns are grouped in blocks of 8.  One workgroup owns a
// PAIR of blocks (16 columns) per round:
//   1. stage its 16 columns of G in LDS,
//   2. Gram matrix  Gamma = G_IJ^T G_IJ  (16x16) with f64 MFMA (the A and B
//      operand of that product are the same register),
//   3. wave 0 diagonalises Gamma (Gamma = R diag R^T) with the wave solver;
//      because A' is well conditioned (kappa <= ~3) forming the Gram matrix
//      loses nothing,
//   4. G_IJ <- G_IJ R and V_IJ <- V_IJ R with MFMA (16 = exactly one tile).
// Block pairs of a round are disjoint (round-robin tournament over the LD/8
// blocks), so a sweep is LD/8 - 1 launches of LD/16 workgroups.
#pragma once
#include "common.hip.h"
#include "jacobi_wave.hip.h"

#define JB_W 8        // columns per block
#define JB_THREADS 256

__global__ void lgj_sigma(int LD, const double *A, double *sigma) {
  __shared__ double s[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < LD; i += 256) m = fmax(m, fabs(A[(size_t)i * LD + i]));
  s[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *sigma = s[0] > 0.0 ? s[0] : 1.0;
}

// Gc[k][r] = A[r][k] - sigma (r == k);  Vc = I   (column-major == row-major: A symmetric)
__global__ void lgj_init(int LD, const double *A, const double *sigma, double *Gc, double *Vc) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)LD * LD) return;
  const int k = idx / LD, r = idx - (size_t)k * LD;
  Gc[idx] = A[idx] - (r == k ? *sigma : 0.0);
  Vc[idx] = (r == k) ? 1.0 : 0.0;
}

__device__ __forceinline__ int jb_rowstride(int LD) { return LD + ((2 - LD % 32 + 32) % 32); }

__global__ __launch_bounds__(JB_THREADS) void lgj_round(int LD, int round, double *Gc, double *Vc,
                                                        unsigned long long *off_bits) {
  extern __shared__ double lds[];
  const int RS = jb_rowstride(LD);
  double *sG = lds;                 // [16][RS]
  double *sGam = sG + 16 * RS;      // [16][17]  Gram, also "A" of the wave solver
  double *sJG = sGam + 16 * 17;     // [16][17]
  double *sJV = sJG + 16 * 17;      // [16][17]  R: column c' at sJV + c'*17
  double *sJl = sJV + 16 * 17;      // [16]
  double *sPart = sJl + 16;         // [4][256] partial Gram per wave

  const int nb = LD / JB_W;
  int bi, bj;
  rr_pair(nb, round, blockIdx.x, bi, bj);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
  auto gcol = [&](int c) { return c < JB_W ? bi * JB_W + c : bj * JB_W + (c - JB_W); };

  // 1. stage G columns
  for (int c = 0; c < 16; ++c) {
    const double *src = Gc + (size_t)gcol(c) * LD;
    for (int r = threadIdx.x; r < LD; r += JB_THREADS) sG[c * RS + r] = src[r];
  }
  __syncthreads();
  // 2. Gram via MFMA: lane (lo, hi) feeds G[r = 4 s + hi][c = lo] as A and as B
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  const int nsteps = LD / 4;
  for (int s = wave; s < nsteps; s += 4) {
    const double v = sG[lo * RS + 4 * s + hi];
    acc = mfma_f64(v, v, acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sPart[wave * 256 + (hi + 4 * r) * 16 + lo] = acc[r];
  __syncthreads();
  {
    const int e = threadIdx.x;  // 256 entries
    const double v = sPart[e] + sPart[256 + e] + sPart[512 + e] + sPart[768 + e];
    sGam[(e >> 4) * 17 + (e & 15)] = v;
  }
  __syncthreads();
  // 3. off-diagonal measure + 16x16 eigenproblem (wave 0)
  if (wave == 0) {
    double off = 0.0;
    for (int e = lane; e < 256; e += 64) {
      const int p = e >> 4, q = e & 15;
      if (p < q) {
        const double den = sqrt(sGam[p * 17 + p] * sGam[q * 17 + q]);
        if (den > 0.0) off = fmax(off, fabs(sGam[p * 17 + q]) / den);
      }
    }
    off = wave_max(off);
    if (lane == 0) atomicMax(off_bits, dbl_bits(off));
    wave_jacobi(16, sGam, sJG, sJV, sJl, 17, 0.0);
    // One Newton-Schulz step R <- R (3 I - R^T R) / 2: the product of a few
    // hundred plane rotations is orthogonal only to ~3e-15 and that defect
    // would add up over the ~400 block rounds of a solve.
    for (int e = lane; e < 256; e += 64) {
      const int p = e >> 4, q = e & 15;
      double d = 0.0;
      for (int c = 0; c < 16; ++c) d = fma(sJV[p * 17 + c], sJV[q * 17 + c], d);
      sJG[p * 17 + q] = d;  // N = R^T R
    }
    wave_lds_fence();
    for (int e = lane; e < 256; e += 64) {
      const int q = e >> 4, c = e & 15;
      double d = 0.0;
      for (int p = 0; p < 16; ++p) d = fma(sJV[p * 17 + c], sJG[p * 17 + q], d);
      sGam[q * 17 + c] = 1.5 * sJV[q * 17 + c] - 0.5 * d;  // column q of R'
    }
    wave_lds_fence();
  }
  __syncthreads();
  // 4. apply R:  new^T[c'][r] = sum_c R[c][c'] old^T[c][r]
  double Rf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) Rf[s] = sGam[lo * 17 + 4 * s + hi];  // R[c = 4s+hi][c' = lo]
  const int ntiles = LD / 16;
  for (int job = wave; job < 2 * ntiles; job += 4) {
    const bool isV = job >= ntiles;
    const int r0 = (isV ? job - ntiles : job) * 16;
    d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 4 * s + hi;
      const double bv = isV ? Vc[(size_t)gcol(c) * LD + r0 + lo] : sG[c * RS + r0 + lo];
      o = mfma_f64(Rf[s], bv, o);
    }
    double *dst = isV ? Vc : Gc;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)gcol(hi + 4 * r) * LD + r0 + lo] = o[r];
  }
}

// lam_k = v_k . g_k + sigma ; Ut = Vc (as stored) ; U = Vc^T
__global__ void lgj_finish(int LD, const double *Gc, const double *Vc, const double *sigma,
                           double *lam, double *U) {
  // one wave per column k
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= LD) return;
  double d = 0.0, nn = 0.0;
  for (int r = lane; r < LD; r += 64) {
    const double v = Vc[(size_t)k * LD + r];
    d = fma(v, Gc[(size_t)k * LD + r], d);
    nn = fma(v, v, nn);
    U[(size_t)r * LD + k] = v;
  }
  d = wave_sum(d);
  nn = wave_sum(nn);
  if (lane == 0) lam[k] = d / nn + *sigma;
}
