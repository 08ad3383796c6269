// Large-state path (S > 32; co-evolution 400x400).  MFMA-bound.
//
// All internal matrices are LD x LD with LD = roundup(S, 16), zero padded, so
// every 16x16 MFMA tile is either fully inside or fully outside the matrix.
// Every product of the algorithm is brought to the single form
//
//     C[m][n] = sum_k Aop[k][m] * Bop[k][n]          ("TN", both operands k-major)
//
// by choosing which of U / U^T, G^T, T, M^T is stored, so one
// kernel template serves them all:
//   K1  Pt_b   = I + t_b A + (U^T diag(F_b))^T U^T      Aop = Ut (row-scaled), Bop = Ut
//       epilogue: loss partial, Gt_b^T = -C_b^T / Pt_b / n
//   K2  T_b    = Gt_b U                                  Aop = Gt_b^T, Bop = U
//   K3  Mt    += (T_b^T U) o Phi_b  over a chunk of b    Aop = T_b,   Bop = U
//   K4a X      = M U^T                                   Aop = Mt,    Bop = Ut
//   K4b dA     = U X  -> dQ = D^1/2 dA D^-1/2            Aop = Ut,    Bop = X
//
// Tile: 80 x 80 per workgroup of 5 wavefronts (400 = 5 * 80); wave w owns the
// 16-row strip w and five 16x16 f64 accumulators.  K is staged through LDS in
// steps of 16 with register prefetch of the next step.
#pragma once
#include "common.hip.h"

#define LG_TM 80
#define LG_TN 80
#define LG_KT 16
#define LG_THREADS 320

struct GemmOperands {
  const double *A;   // [K][lda] k-major
  const double *B;   // [K][ldb]
  int lda, ldb;
  int M, N, K;       // all multiples of 16
  const double *kscale;  // optional [K] multiplier applied to rows of Aop
};

// Loads one K-step (16 rows) of an 80-wide panel into registers:
// 16 rows x 40 16-byte chunks = 640 chunks, 2 per thread.
// SCALE is a template parameter and `sc` a reference to a register array: a `double *sc` that may be
// null made the compiler keep the two scales in SCRATCH memory (a store and a load per K-step on the
// critical path of K1's prefetch; 0.415 -> see DESIGN for the time after the change).
template <bool SCALE>
__device__ __forceinline__ void lg_load_panel(const double *__restrict__ P, int ld, int rows_total,
                                              int cols_total, int k0, int c0, double2 (&reg)[2],
                                              const double *__restrict__ kscale, double (&sc)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int chunk = threadIdx.x + u * LG_THREADS;  // 0..639
    const int kr = chunk / 40, cc = (chunk - kr * 40) * 2;
    const int k = k0 + kr, c = c0 + cc;
    if (k < rows_total && c < cols_total)
      reg[u] = *reinterpret_cast<const double2 *>(P + (size_t)k * ld + c);
    else
      reg[u] = double2{0.0, 0.0};
    // the row scale travels with the panel (fetching it at LDS-store time exposed a
    // global-load latency in every K-step)
    if (SCALE) sc[u] = (k < rows_total) ? kscale[k] : 0.0;
  }
}

template <bool SCALE>
__device__ __forceinline__ void lg_store_panel(double *s, const double2 (&reg)[2], const double (&sc)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int chunk = threadIdx.x + u * LG_THREADS;
    const int kr = chunk / 40, cc = (chunk - kr * 40) * 2;
    double2 v = reg[u];
    if (SCALE) {
      v.x *= sc[u];
      v.y *= sc[u];
    }
    *reinterpret_cast<double2 *>(s + kr * LG_TM + cc) = v;
  }
}

// acc[j] (j = 0..4): tile rows m0 + 16*wave + (l>>4) + 4r, cols n0 + 16*j + (l&15)
// sA / sB hold TWO K-steps each (double buffer): one barrier per K-step; the
// global loads of step k+1 are in flight during the MFMAs of step k.
template <bool SCALE = false>
__device__ __forceinline__ void lg_gemm_tile(const GemmOperands &g, int m0, int n0, double *sA,
                                             double *sB, d4 (&acc)[5]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int j = 0; j < 5; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  double2 ra[2], rb[2];
  double sc[2] = {1.0, 1.0}, one[2] = {1.0, 1.0};
  const int nk = g.K / LG_KT;
  lg_load_panel<SCALE>(g.A, g.lda, g.K, g.M, 0, m0, ra, g.kscale, sc);
  lg_load_panel<false>(g.B, g.ldb, g.K, g.N, 0, n0, rb, nullptr, one);
  __syncthreads();  // the previous tile's readers of buffer 0 are done
  lg_store_panel<SCALE>(sA, ra, sc);
  lg_store_panel<false>(sB, rb, one);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const double *cA = sA + (kt & 1) * (LG_KT * LG_TM), *cB = sB + (kt & 1) * (LG_KT * LG_TN);
    if (kt + 1 < nk) {
      lg_load_panel<SCALE>(g.A, g.lda, g.K, g.M, (kt + 1) * LG_KT, m0, ra, g.kscale, sc);
      lg_load_panel<false>(g.B, g.ldb, g.K, g.N, (kt + 1) * LG_KT, n0, rb, nullptr, one);
    }
#pragma unroll
    for (int s = 0; s < LG_KT / 4; ++s) {
      const double av = cA[(4 * s + hi) * LG_TM + 16 * wave + lo];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const double bv = cB[(4 * s + hi) * LG_TN + 16 * j + lo];
        acc[j] = mfma_f64(av, bv, acc[j]);
      }
    }
    if (kt + 1 < nk) {
      // buffer (kt+1)&1 was last read in step kt-1; every wave passed the barrier since
      lg_store_panel<SCALE>(sA + ((kt + 1) & 1) * (LG_KT * LG_TM), ra, sc);
      lg_store_panel<false>(sB + ((kt + 1) & 1) * (LG_KT * LG_TN), rb, one);
    }
    __syncthreads();
  }
}

// ---- the same tile with FOUR waves per workgroup, one per SIMD ------------------------------------
// A 5-wave workgroup puts two of its waves on one SIMD (which one rotates from workgroup to workgroup,
// profiles/tools/simd_probe), so with four workgroups on a CU the SIMDs carry 4..7 waves and the busiest
// one sets the pace.  Here the 25 MFMA tiles of the 80 x 80 tile are dealt 7/6/6/6: wave w owns strip w
// (acc[0..4]) and tile (4, w) of the fifth strip (ax0); wave 0 also tile (4, 4) (ax1).  Every workgroup
// then loads the four SIMDs equally: K2 0.378 -> 0.330 ms at LD = 400.  (Tile (4, 4) itself is split
// over K among the four waves, see the loop.)
#define LG4_THREADS 256
// Panel loads: thread -> (row tid / 16 of the 16-row K-step, 16-byte chunks tid % 16, + 16, + 32 of that
// row's 40), so the three loads of a thread share ONE row pointer and ONE row scale.  (A flat
// chunk = tid + 256 u mapping needed three pointers and three scales per operand: 18 loop-invariant
// registers, which the compiler spilled to scratch and re-loaded in every K-step: +170 MB of traffic
// in K1.)
template <bool SCALE>
__device__ __forceinline__ void lg4_load_panel(const double *__restrict__ P, int ld, int rows_total, int cols_total,
                                               int k0, int c0, double2 (&reg)[3],
                                               const double *__restrict__ kscale, double &sc) {
  const int kr = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const int k = k0 + kr;
  const double *row = P + (size_t)k * ld + c0 + 2 * cq;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int c = c0 + 2 * (cq + 16 * u);
    if ((u < 2 || cq < 8) && k < rows_total && c < cols_total)
      reg[u] = *reinterpret_cast<const double2 *>(row + 32 * u);
    else
      reg[u] = double2{0.0, 0.0};
  }
  if (SCALE) sc = k < rows_total ? kscale[k] : 0.0;
}
template <bool SCALE>
__device__ __forceinline__ void lg4_store_panel(double *s, const double2 (&reg)[3], double sc) {
  const int kr = threadIdx.x >> 4, cq = threadIdx.x & 15;
  double *row = s + kr * LG_TM + 2 * cq;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    double2 v = reg[u];
    if (SCALE) {
      v.x *= sc;
      v.y *= sc;
    }
    if (u < 2 || cq < 8) *reinterpret_cast<double2 *>(row + 32 * u) = v;
  }
}
template <bool SCALE = false>
__device__ __forceinline__ void lg4_gemm_tile(const GemmOperands &g, int m0, int n0, double *sA, double *sB,
                                              d4 (&acc)[5], d4 &ax0, d4 &ax1) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int j = 0; j < 5; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  ax0 = d4{0.0, 0.0, 0.0, 0.0};
  ax1 = ax0;
  double2 ra[3], rb[3];
  double sc = 1.0, one = 1.0;
  const int nk = g.K / LG_KT;
  lg4_load_panel<SCALE>(g.A, g.lda, g.K, g.M, 0, m0, ra, g.kscale, sc);
  lg4_load_panel<false>(g.B, g.ldb, g.K, g.N, 0, n0, rb, nullptr, one);
  __syncthreads();  // the previous tile's readers of buffer 0 are done
  lg4_store_panel<SCALE>(sA, ra, sc);
  lg4_store_panel<false>(sB, rb, one);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const double *cA = sA + (kt & 1) * (LG_KT * LG_TM), *cB = sB + (kt & 1) * (LG_KT * LG_TN);
    if (kt + 1 < nk) {
      lg4_load_panel<SCALE>(g.A, g.lda, g.K, g.M, (kt + 1) * LG_KT, m0, ra, g.kscale, sc);
      lg4_load_panel<false>(g.B, g.ldb, g.K, g.N, (kt + 1) * LG_KT, n0, rb, nullptr, one);
    }
#pragma unroll
    for (int s = 0; s < LG_KT / 4; ++s) {
      const double av = cA[(4 * s + hi) * LG_TM + 16 * wave + lo];
      const double a4 = cA[(4 * s + hi) * LG_TM + 64 + lo];
      double bv[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) bv[j] = cB[(4 * s + hi) * LG_TN + 16 * j + lo];
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = mfma_f64(av, bv[j], acc[j]);
      // tile (4, wave): column block `wave` (a wave-uniform choice among registers)
      const double bx = wave == 0 ? bv[0] : wave == 1 ? bv[1] : wave == 2 ? bv[2] : bv[3];
      ax0 = mfma_f64(a4, bx, ax0);
      // tile (4, 4): its K range is dealt round-robin to the four waves (6.25 MFMA tiles each
      // instead of 7/6/6/6); the partial sums meet in wave 0 below
      if ((kt & 3) == wave) ax1 = mfma_f64(a4, bv[4], ax1);
    }
    if (kt + 1 < nk) {
      lg4_store_panel<SCALE>(sA + ((kt + 1) & 1) * (LG_KT * LG_TM), ra, sc);
      lg4_store_panel<false>(sB + ((kt + 1) & 1) * (LG_KT * LG_TN), rb, one);
    }
    __syncthreads();
  }
  if (wave != 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[(wave - 1) * 256 + r * 64 + lane] = ax1[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ax1[r] += (sA[r * 64 + lane] + sA[256 + r * 64 + lane]) + sA[512 + r * 64 + lane];
  }
  __syncthreads();   // sA is free again (callers reuse it)
}

// f(row, col, value) for every element of the tile this lane owns (either wave layout)
template <int NW, typename F>
__device__ __forceinline__ void lg_for_each(int m0, int n0, const d4 (&acc)[5], const d4 &ax0, const d4 &ax1, F &&f) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 16 * wave + hi + 4 * r;
#pragma unroll
    for (int j = 0; j < 5; ++j) f(row, n0 + 16 * j + lo, acc[j][r]);
    if (NW == 4) {
      const int row4 = m0 + 64 + hi + 4 * r;
      f(row4, n0 + 16 * wave + lo, ax0[r]);
      if (wave == 0) f(row4, n0 + 64 + lo, ax1[r]);
    }
  }
}
template <int NW, bool SCALE>
__device__ __forceinline__ void lg_tile(const GemmOperands &g, int m0, int n0, double *sA, double *sB, d4 (&acc)[5],
                                        d4 &ax0, d4 &ax1) {
  if (NW == 4) lg4_gemm_tile<SCALE>(g, m0, n0, sA, sB, acc, ax0, ax1);
  else lg_gemm_tile<SCALE>(g, m0, n0, sA, sB, acc);
}

// XCD-aware block id: hardware deals consecutive workgroup ids round-robin over the 8
// XCDs (ids i and i + 8 share an L2).  Remap so that each XCD walks a CONTIGUOUS range
// of virtual ids: the 25 tiles of one bucket then run on one XCD and share its L2 for
// the operand panels (measured before: 5.4x the algorithmic bytes left L2).  Bijective
// for any grid size (guide 5.5 T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_swizzle(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// ------------------------------------------------------------------ K1
struct K1Args {
  int S, LD, B;
  const double *Ut;      // [LD][LD]  Ut[k][i] = U[i][k]
  const double *A;       // [LD][LD]  symmetric
  const double *t;       // [B]
  const double *F;       // [B][LD]   phi2(t_b lam_k) (split) or exp(t_b lam_k)
  const double *sigma;   // max |A_ii|: bucket uses the split form iff 2 sigma t_b <= 1
  const double *Ct;      // [B][LD][LD] transposed counts (padded)
  double *Gt;            // [B][LD][LD] out: Gt^T
  double *loss_part;     // [B * tiles] out
  double inv_n;
  const double *dsq;     // [LD] sqrt(pi) (expm mode)
  double *P;             // [B][S][S] (expm mode) or null
};

// Pt_b is symmetric: only the tilesN (tilesN + 1) / 2 tiles with tm <= tn run the main loop; an
// off-diagonal tile serves both (row, col) and (col, row) in its epilogue (same Pt value, its own
// count and its own Gt^T entry).  40 % fewer MFMAs than the full grid at LD = 400.
template <int NW>   // 5: one wave per 16-row strip; 4: one wave per SIMD (lg4_gemm_tile)
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 5) void k1_pt_loss_gt(K1Args a) {  // four workgroups per CU (96 / 128 VGPRs)
  __shared__ double sA[2 * LG_KT * LG_TM];
  __shared__ double sB[2 * LG_KT * LG_TN];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN, tiles = tilesN * (tilesN + 1) / 2;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles;
  int tile = vid - b * tiles, tm = 0;
  while (tile >= tilesN - tm) {  // row tm of the upper triangle holds tilesN - tm tiles
    tile -= tilesN - tm;
    ++tm;
  }
  const int tn = tm + tile;
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t boff = (size_t)b * a.LD * a.LD;
  GemmOperands g{a.Ut, a.Ut, a.LD, a.LD, a.LD, a.LD, a.LD, a.F + (size_t)b * a.LD};
  d4 acc[5], ax0, ax1;
  lg_tile<NW, true>(g, m0, n0, sA, sB, acc, ax0, ax1);

  const double tb = a.t[b];
  const bool split = tb * 2.0 * (*a.sigma) <= 1.0;  // see small_bank.hip.h
  const bool mirror = tm != tn;
  double lossacc = 0.0;
  auto emit = [&](int row, int col, double pt) {
    const size_t idx = (size_t)row * a.LD + col;
    if (a.P) {
      if (row < a.S && col < a.S)
        a.P[(size_t)b * a.S * a.S + (size_t)row * a.S + col] = pt * a.dsq[col] / a.dsq[row];
    } else {
      const double c = a.Ct[boff + idx];
      const bool nz = c != 0.0;
      lossacc = fma(-c, fast_log(nz ? pt : 1.0), lossacc);
      a.Gt[boff + idx] = nz ? -c * a.inv_n * fast_rcp(pt) : 0.0;
    }
  };
  lg_for_each<NW>(m0, n0, acc, ax0, ax1, [&](int row, int col, double pt) {
    if (row < a.LD && col < a.LD) {
      if (split) pt += tb * a.A[(size_t)row * a.LD + col] + (row == col ? 1.0 : 0.0);
      else if (row >= a.S || col >= a.S) pt = 1.0;  // pad (never used: C = 0 there)
      emit(row, col, pt);
      if (mirror) emit(col, row, pt);
    }
  });
  if (a.P) return;
  lossacc = wave_sum(lossacc);
  // sA is free after the K loop (the tile routine ends with a barrier)
  if (lane == 0) sA[wave] = lossacc;
  __syncthreads();
  if (threadIdx.x == 0) a.loss_part[vid] = sA[0] + sA[1] + sA[2] + sA[3] + (NW == 5 ? sA[4] : 0.0);
}

// ------------------------------------------------------------------ K2
struct K2Args {
  int LD;
  const double *Gt;  // [B][LD][LD]
  const double *U;   // [LD][LD]
  double *T;         // [B][LD][LD]
};

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 5) void k2_t_eq_g_u(K2Args a) {
  __shared__ double sA[2 * LG_KT * LG_TM];
  __shared__ double sB[2 * LG_KT * LG_TN];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN, tiles = tilesN * tilesN;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles, tile = vid - b * tiles;
  const int tm = tile / tilesN, tn = tile - tm * tilesN;
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  const size_t boff = (size_t)b * a.LD * a.LD;
  GemmOperands g{a.Gt + boff, a.U, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
  d4 acc[5], ax0, ax1;
  lg_tile<NW, false>(g, m0, n0, sA, sB, acc, ax0, ax1);
  lg_for_each<NW>(m0, n0, acc, ax0, ax1, [&](int row, int col, double v) {
    if (row < a.LD && col < a.LD) a.T[boff + (size_t)row * a.LD + col] = v;
  });
}

// ------------------------------------------------------------------ K3
// Wt_b[c][a] = Phi_b[c][a] * sum_i T_b[i][c] U[i][a], one (bucket, tile) per workgroup,
// written over Gt_b (dead once K2 has produced T_b); k3_reduce then sums the buckets
// in a fixed order.  (A version that kept the running sum over a chunk of buckets in
// registers needed 2 accumulator sets: 256 VGPRs, one workgroup per CU.)
struct K3Args {
  int LD, B;
  const double *T;       // [B][LD][LD]
  const double *U;       // [LD][LD]
  const double *t;       // [B]
  const double *lam;     // [LD]
  const double *E;       // [B][LD] exp(t lam)
  const double *H;       // [B][LD] exp(t lam / 2)
  double *W;             // [B][LD][LD] out (aliases the Gt buffer)
  int sym;               // counts symmetric => Gt_b, hence W_b, symmetric: upper-triangular tiles only
};

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 5) void k3_w_phi(K3Args a) {
  __shared__ double sA[2 * LG_KT * LG_TM];
  __shared__ double sB[2 * LG_KT * LG_TN];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN;
  const int tiles = a.sym ? tilesN * (tilesN + 1) / 2 : tilesN * tilesN;
  const int vid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = vid / tiles;
  int tile = vid - b * tiles, tm, tn;
  if (a.sym) {
    tm = 0;
    while (tile >= tilesN - tm) {
      tile -= tilesN - tm;
      ++tm;
    }
    tn = tm + tile;
  } else {
    tm = tile / tilesN;
    tn = tile - tm * tilesN;
  }
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  const size_t boff = (size_t)b * a.LD * a.LD;
  GemmOperands g{a.T + boff, a.U, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
  d4 acc[5], ax0, ax1;
  lg_tile<NW, false>(g, m0, n0, sA, sB, acc, ax0, ax1);
  const double tb = a.t[b];
  const double *Eb = a.E + (size_t)b * a.LD, *Hb = a.H + (size_t)b * a.LD;
  lg_for_each<NW>(m0, n0, acc, ax0, ax1, [&](int row, int col, double v) {
    if (row < a.LD && col < a.LD) {
      const double ph = divdiff_fast(tb, a.lam[row], a.lam[col], Eb[row], Eb[col], Hb[row], Hb[col]);
      a.W[boff + (size_t)row * a.LD + col] = v * ph;   // (symmetric case: k3_reduce mirrors the sum, not every bucket)
    }
  });
}

// Mt = sum over chunks (fixed order => bitwise reproducible).  sym (LD > 0): only the 80x80 tiles on
// or above the diagonal were written by k3_w_phi; sum those and mirror the SUM into the lower tiles
// (40 % less to read, and no mirrored stores per bucket).
__global__ void k3_reduce(const double *part, int nchunks, size_t n, double *out, int LD = 0) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int row = 0, col = 0;
  if (LD > 0) {
    row = (int)(i / LD);
    col = (int)(i - (size_t)row * LD);
    if (row / LG_TM > col / LG_TN) return;
  }
  double s0 = 0.0, s1 = 0.0;   // two interleaved partial sums: twice the loads in flight
  int c = 0;
  for (; c + 1 < nchunks; c += 2) {
    s0 += part[(size_t)c * n + i];
    s1 += part[(size_t)(c + 1) * n + i];
  }
  if (c < nchunks) s0 += part[(size_t)c * n + i];
  const double s = s0 + s1;
  out[i] = s;
  if (LD > 0 && row / LG_TM < col / LG_TN) out[(size_t)col * LD + row] = s;
}

// ------------------------------------------------------------------ K4 (plain / dQ epilogue)
struct K4Args {
  int S, LD;
  const double *Aop, *Bop;  // [LD][LD] each
  double *out;              // [LD][LD] or dQ [S][S]
  const double *dsq;        // non-null => out = dQ[i][j] = d_i * acc / d_j, unpadded S x S
  const double *sub;        // non-null => out = acc - (*sub_scale) * sub[row][col]
  const double *sub_scale;
  double *outT = nullptr;   // sg_gemm only: also write the transpose of the result
  double *diag = nullptr;   // sg_gemm only: also write the diagonal of the result
  // sg_gemm only: operands chosen ON THE DEVICE -- when *sel != 0 the non-null alternatives replace
  // Aop / Bop.  Lets the host enqueue the products of a first-order sweep before it knows which X
  // (all pairs or far pairs only) lgx_build decided on.
  const unsigned long long *sel = nullptr;
  const double *Aalt = nullptr, *Balt = nullptr;
};

__global__ __launch_bounds__(LG_THREADS) void k4_gemm(K4Args a) {
  __shared__ double sA[2 * LG_KT * LG_TM];
  __shared__ double sB[2 * LG_KT * LG_TN];
  const int tilesN = (a.LD + LG_TN - 1) / LG_TN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
  const int m0 = tm * LG_TM, n0 = tn * LG_TN;
  GemmOperands g{a.Aop, a.Bop, a.LD, a.LD, a.LD, a.LD, a.LD, nullptr};
  d4 acc[5];
  lg_gemm_tile(g, m0, n0, sA, sB, acc);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * wave + hi + 4 * r, col = n0 + 16 * j + lo;
      if (a.dsq) {
        if (row < a.S && col < a.S)
          a.out[(size_t)row * a.S + col] = a.dsq[row] * acc[j][r] / a.dsq[col];
      } else if (row < a.LD && col < a.LD) {
        const size_t idx = (size_t)row * a.LD + col;
        a.out[idx] = a.sub ? acc[j][r] - (*a.sub_scale) * a.sub[idx] : acc[j][r];
      }
    }
}

// Single-matrix products (K4a, K4b, the warm-start G0 = A' U_prev, the first-order eigen
// correction): one LD^3 GEMM is only (LD/80)^2 = 25 of the 80x80 tiles, i.e. 25 of 256 CUs and
// a 25-step serial K loop per tile (43 us at LD = 400).  Here a workgroup owns a 16 x 80 strip
// and its four waves split K (k-step s goes to wave s mod 4); operand fragments are read
// straight from L2 in MFMA layout (no LDS staging: the matrices are 1.3 MB), the four partial
// strips are summed through LDS in a fixed order.  (LD/16) x ceil(LD/80) = 125 workgroups.
// Same argument block and epilogues as k4_gemm, plus:
//   ns == 1 :  out = (row == col) + sub[row][col] - acc / 2      (R = I + X + X^2/2, X^2 = -X^T X)
//   ns == 2 :  out = alpha * acc + beta * sub[row][col]          (polynomial / Newton-Schulz steps of the
//                                                                 first-order sweeps, jacobi_block.hip.h)
// NJ = 16-column tiles per workgroup (strip 16 x 16 NJ): fewer -> more workgroups and shorter waves
// (these single 400^3 products are latency bound, not MFMA bound).
template <int NW, int UU, int NJ = 5>
__global__ __launch_bounds__(NW * 64) void sg_gemm(K4Args a, int ns, double alpha, double beta) {
  __shared__ double sRed[4][NJ][256];
  const int LD = a.LD, tilesN = (LD + 16 * NJ - 1) / (16 * NJ);
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
  const int m0 = tm * 16, n0 = tn * 16 * NJ;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  d4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  int ncol[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ncol[j] = min(n0 + 16 * j + lo, LD - 1);  // clamped: tiles past LD are discarded
  const int nsteps = LD / 4;
  const bool alt = a.sel && *a.sel != 0ull;
  const double *Ap = (alt && a.Aalt ? a.Aalt : a.Aop) + m0 + lo, *Bp = alt && a.Balt ? a.Balt : a.Bop;
  for (int s0 = wave; s0 < nsteps; s0 += UU * NW) {   // UU k-steps of this wave in flight
    double av[UU], bv[UU][NJ];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const int s = min(s0 + NW * u, nsteps - 1);
      const size_t krow = (size_t)(4 * s + hi) * LD;
      av[u] = Ap[krow];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[u][j] = Bp[krow + ncol[j]];
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      if (s0 + NW * u < nsteps) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = mfma_f64(av[u], bv[u][j], acc[j]);
      }
    }
  }
  // K was split over the NW waves: fold the upper waves into the lower four, then sum those
  for (int half = NW / 2; half >= 4; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sRed[wave - half][j][r * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] += sRed[wave][j][r * 64 + lane];
    }
    __syncthreads();
  }
  if (wave < 4) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sRed[wave][j][r * 64 + lane] = acc[j][r];
  }
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int t = threadIdx.x, r = t >> 6, l = t & 63;
  const int row = m0 + (l >> 4) + 4 * r;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + 16 * j + (l & 15);
    if (col >= LD) continue;
    const double v = (sRed[0][j][t] + sRed[1][j][t]) + (sRed[2][j][t] + sRed[3][j][t]);
    if (a.dsq) {
      if (row < a.S && col < a.S) a.out[(size_t)row * a.S + col] = a.dsq[row] * v / a.dsq[col];
    } else {
      const size_t idx = (size_t)row * LD + col;
      double o;
      if (ns == 1) o = (row == col ? 1.0 : 0.0) + a.sub[idx] - 0.5 * v;
      else if (ns == 2) o = fma(alpha, v, beta * a.sub[idx]);
      else o = a.sub ? v - (*a.sub_scale) * a.sub[idx] : v;
      a.out[idx] = o;
      if (a.outT) a.outT[(size_t)col * LD + row] = o;       // transposed copy (the squarings need R^T)
      if (a.diag && row == col) a.diag[row] = o;
    }
  }
}

// ------------------------------------------------------------------ small helpers
// A = sym(D^1/2 Q D^-1/2) into padded LD x LD, dsq = sqrt(pi) (1 on the pad)
__global__ void lg_build_A(int S, int LD, const double *Q, const double *pi, double *A,
                           double *dsq) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < LD) dsq[idx] = idx < S ? sqrt(pi[idx]) : 1.0;
  if (idx >= LD * LD) return;
  const int i = idx / LD, j = idx - i * LD;
  double v = 0.0;
  if (i < S && j < S) {
    const double di = sqrt(pi[i]), dj = sqrt(pi[j]);
    v = 0.5 * (di * Q[(size_t)i * S + j] / dj + dj * Q[(size_t)j * S + i] / di);
  }
  A[idx] = v;
}

// spectral tables F = phi2(t lam), E = exp(t lam), H = exp(t lam / 2): [B][LD]
__global__ void lg_tables(int LD, int B, const double *t, const double *lam,
                          const double *sigma, double *F, double *E, double *H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * LD) return;
  const int b = idx / LD, k = idx - b * LD;
  const double x = t[b] * lam[k];
  const bool split = t[b] * 2.0 * (*sigma) <= 1.0;
  F[idx] = split ? phi2(x) : exp(x);
  E[idx] = exp(x);
  H[idx] = exp(0.5 * x);
}

// loss = (sum of partials - direct term) * inv_n, single thread-block, fixed order
__global__ void lg_finish_loss(const double *part, int nparts, int S, const double *dsq,
                               const double *dirsum, double inv_n, double *loss) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  double dir = 0.0;
  for (int k = threadIdx.x; k < S; k += 256) dir = fma(log(dsq[k]), dirsum[k], dir);
  s[threadIdx.x] = acc - dir;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = s[0] * inv_n;
}

// flag[0] |= 1 when some live bucket has C_b != C_b^T
__global__ void lg_sym_check(int LD, const double *Ct, int *flag) {
  const size_t boff = (size_t)blockIdx.z * LD * LD;
  const int i = blockIdx.y * blockDim.y + threadIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < LD && j < i && Ct[boff + (size_t)i * LD + j] != Ct[boff + (size_t)j * LD + i]) atomicOr(flag, 1);
}

// pad + transpose counts at create time: Ct[b][j][i] = C[b][i][j]
// destination bucket z holds source bucket src[z] (live buckets only, see cb_create)
__global__ void lg_transpose_pad(int S, int LD, const double *C, double *Ct, const int *src) {
  __shared__ double tile[32][33];
  const int b = src[blockIdx.z];
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int i = i0 + r, j = j0 + threadIdx.x;
    tile[r][threadIdx.x] = (i < S && j < S) ? C[(size_t)b * S * S + (size_t)i * S + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int j = j0 + r, i = i0 + threadIdx.x;
    if (j < LD && i < LD) Ct[(size_t)blockIdx.z * LD * LD + (size_t)j * LD + i] = tile[threadIdx.x][r];
  }
}
