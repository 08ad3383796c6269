// Block one-sided Jacobi eigensolver for the large path (S > 32, LD <= 1024).
//
// Same mathematics as jacobi_wave.hip.h -- Hestenes Jacobi on A' = A - sigma I,
// columns of G = A' V made mutually orthogonal, V never formed,
// U = -normalised(G) -- but columns are grouped in blocks of 8.  One workgroup
// owns a PAIR of blocks (16 columns) per round:
//   1. stage its 16 columns of G in LDS,
//   2. Gram matrix  Gamma = G_IJ^T G_IJ  (16x16) with f64 MFMA (the A and B
//      operand of that product are the same register),
//   3. wave 0 finds the orthogonal R diagonalising Gamma with the wave solver
//      (because A' is well conditioned, kappa <= ~3, forming the Gram matrix
//      loses nothing) and polishes R with one Newton-Schulz step,
//   4. G_IJ <- G_IJ R with MFMA (16 columns = exactly one tile).
// Block pairs of a round are disjoint (round-robin tournament over the LD/8
// blocks), so a sweep is LD/8 - 1 launches of LD/16 workgroups.
#pragma once
#include "common.hip.h"
#include "jacobi_wave.hip.h"

#define JB_W 8        // columns per block
#define JB_THREADS 512
#define JB_WAVES (JB_THREADS / 64)

__global__ void lgj_sigma(int LD, const double *A, double *sigma, unsigned long long *zero64 = nullptr) {
  __shared__ double s[256];
  if (zero64 && threadIdx.x < 64) zero64[threadIdx.x] = 0ull;   // the solve's state words (saves a memset launch)
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

// Gc[k][r] = A[r][k] - sigma (r == k)   (column-major == row-major: A symmetric)
__global__ void lgj_init(int LD, const double *A, const double *sigma, double *Gc) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)LD * LD) return;
  const int k = idx / LD, r = idx - (size_t)k * LD;
  Gc[idx] = A[idx] - (r == k ? *sigma : 0.0);
}

__device__ __forceinline__ int jb_rowstride(int LD) { return LD + ((2 - LD % 32 + 32) % 32); }

// round >= 0: tournament round `round`, cross-block pairs only (inner_sweeps == 0) or
//             full sweeps of all 120 pairs (inner_sweeps > 0);
// round <  0: the "within" pass: workgroup w takes blocks 2w, 2w+1 and rotates only
//             the pairs inside each block.  One sweep = the LD/8 - 1 rounds + 1 within
//             pass visits every column pair exactly once.
// round <= -10: banded round of the hybrid sweep (large_eigh): code = -round - 10, k = code / 2 + 2,
//             parity = code & 1: block pairs (i, i + k) with (i / k) % 2 == parity, cross pairs once
//             (the eigen-columns are kept sorted, so near-degenerate columns are a few blocks apart).
// must_zero / must_nonzero (planned solves, eigh_planned.hip.h): words of the device-side control block that decide whether
// this launch runs (the stall word; "the last sweep rotated far pairs only").
__global__ __launch_bounds__(JB_THREADS) void lgj_round(int LD, int round, int inner_sweeps,
                                                        double *Gc, unsigned long long *off_bits,
                                                        const unsigned long long *must_zero = nullptr,
                                                        const unsigned long long *must_nonzero = nullptr) {
  // off_bits[0]: running max cosine of this sweep; off_bits[1]: solve finished (set by lgj_check):
  // sweeps are enqueued speculatively, the surplus launches return at once
#ifdef CB_EIGH_STAMPS
  if (must_zero && blockIdx.x == 0 && threadIdx.x == 0) {   // (planned solves: must_zero is the control block's first word)
    unsigned long long *c = const_cast<unsigned long long *>(must_zero);
    const unsigned long long i = atomicAdd(c + 95, 1ull);
    if (i < 96ull) c[96 + i] = ((unsigned long long)__builtin_amdgcn_s_memrealtime() << 8) | 3ull;
  }
#endif
  {
    // the three words that decide whether this launch runs, in ONE batch of scalar loads (constant address space); as plain
    // loads in the short-circuit order of the tests they were three dependent round trips behind a kernel boundary
#ifdef CB_PLAIN_WORDS
    typedef const unsigned long long *const_words;
#else
    typedef const __attribute__((address_space(4))) unsigned long long *const_words;
#endif
    const_words p0 = (const_words)off_bits, p1 = (const_words)(must_zero ? must_zero : off_bits + 1),
                p2 = (const_words)(must_nonzero ? must_nonzero : off_bits + 1);
    const unsigned long long w0 = p0[1], w1 = *p1, w2 = *p2;
    if (w0 != 0ull) return;
    if ((must_zero && w1 != 0ull) || (must_nonzero && w2 == 0ull)) return;
  }
  extern __shared__ double lds[];
  const int RS = jb_rowstride(LD);
  double *sG = lds;                 // [16][RS]
  double *sGam = sG + 16 * RS;      // [16][17]  Gram -> orthogonalised columns
  double *sR = sGam + 16 * 17;      // [16][17]  R: column c' at sR + c'*17
  double *sN = sR + 16 * 17;        // [16][17]  R^T R, then polished R
  double *sPart = sN + 16 * 17;     // [JB_WAVES][256] partial Gram per wave

  const int nb = LD / JB_W;
  int bi, bj;
  if (round >= 0) {
    rr_pair(nb, round, blockIdx.x, bi, bj);
  } else if (round <= -10) {
    const int code = -round - 10, k = code / 2 + 2, par = code & 1, w = blockIdx.x;
    bi = (w / k) * 2 * k + par * k + (w % k);
    bj = bi + k;
    if (bj >= nb) return;
  } else if (round == -1) {
    bi = 2 * blockIdx.x;
    bj = bi + 1;
  } else {  // -2: groups shifted by one block, so clusters straddling a group edge get their turn
    bi = 2 * blockIdx.x + 1;
    bj = (bi + 1) % nb;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, hi = lane >> 4;
  auto gcol = [&](int c) { return c < JB_W ? bi * JB_W + c : bj * JB_W + (c - JB_W); };

  // 1. stage G columns: all 16 loads of a row slab in flight before the first LDS store
  // (round 6 measured the loads issued BEFORE the words that decide whether the launch runs are looked at, as lge_gemm does with
  // its operands: + 0.3 us per launch -- 25 workgroups do not queue behind a cold L2 the way 625 do; not kept)
  {
    const double *src_i = Gc + (size_t)bi * JB_W * LD, *src_j = Gc + (size_t)bj * JB_W * LD;
    for (int r = threadIdx.x; r < LD; r += JB_THREADS) {
      double v[16];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v[c] = src_i[(size_t)c * LD + r];
        v[8 + c] = src_j[(size_t)c * LD + r];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) sG[c * RS + r] = v[c];
    }
  }
  __syncthreads();
  // 2. Gram via MFMA: lane (lo, hi) feeds G[r = 4 s + hi][c = lo] as A and as B.
  //    Four independent accumulators: a dependent f64 MFMA chain costs ~300 cycles a link.
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  {
    // this wave's k-steps are s = wave, wave + JB_WAVES, ...; batches of 8: all 8 LDS reads
    // in flight, then 8 MFMAs on 4 independent accumulators
    d4 a0 = acc, a1 = acc, a2 = acc, a3 = acc;
    const int nsteps = LD / 4;
    for (int s0 = wave; s0 < nsteps; s0 += 8 * JB_WAVES) {
      double v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int st = s0 + JB_WAVES * i;
        const double x = sG[lo * RS + 4 * min(st, nsteps - 1) + hi];
        v[i] = st < nsteps ? x : 0.0;
      }
      a0 = mfma_f64(v[0], v[0], a0);
      a1 = mfma_f64(v[1], v[1], a1);
      a2 = mfma_f64(v[2], v[2], a2);
      a3 = mfma_f64(v[3], v[3], a3);
      a0 = mfma_f64(v[4], v[4], a0);
      a1 = mfma_f64(v[5], v[5], a1);
      a2 = mfma_f64(v[6], v[6], a2);
      a3 = mfma_f64(v[7], v[7], a3);
    }
    acc = (a0 + a1) + (a2 + a3);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sPart[wave * 256 + (hi + 4 * r) * 16 + lo] = acc[r];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int e = threadIdx.x;  // 256 entries
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < JB_WAVES; ++w) v += sPart[w * 256 + e];
    sGam[(e >> 4) * 17 + (e & 15)] = v;
  }
  __syncthreads();
  // 3. 16x16 rotation (wave 0); the convergence measure comes out of the rotation loop
  if (wave == 0) {
    double off;
    if (inner_sweeps > 0) {
      off = 0.0;
      for (int e = lane; e < 256; e += 64) {
        const int p = e >> 4, q = e & 15;
        if (p < q) {
          const double den2 = sGam[p * 17 + p] * sGam[q * 17 + q];
          if (den2 > 0.0) off = fmax(off, fabs(sGam[p * 17 + q]) * fast_rsqrt(den2));
        }
      }
      off = wave_max(off);
      if (inner_sweeps >= 100) {
        wave_rotation_spd16(sGam, sR, 17, inner_sweeps - 100);   // (the 4-lanes-per-pair cyclic solver, for comparison)
      } else {
        // a sweep over all 120 pairs = the 64 cross pairs + the 2 x 28 pairs inside the blocks, 8 lanes
        // per pair (all 64 lanes busy; the cyclic solver above keeps 32 busy and takes 1.6x as long)
        for (int e = lane; e < 256; e += 64) sR[(e >> 4) * 17 + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        wave_lds_fence();
        for (int s = 0; s < inner_sweeps; ++s) {
          const double o1 = wave_rotation_spd16_blockpairs(sGam, sR, 17, true);
          const double o2 = wave_rotation_spd16_blockpairs(sGam, sR, 17, false);
          if (fmax(o1, o2) <= CB_JAC_STOP * CB_JAC_STOP) break;
        }
      }
    } else {
      double off2;
      if (round >= 0 || round <= -10) {
        off2 = wave_rotation_cross16_regs(sGam, sR, 17);
      } else {
        for (int e = lane; e < 256; e += 64) sR[(e >> 4) * 17 + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        wave_lds_fence();
        off2 = wave_rotation_spd16_blockpairs(sGam, sR, 17, false);
      }
      off = off2 > 0.0 ? off2 * fast_rsqrt(off2) : 0.0;
    }
    if (lane == 0) {
      atomicMax(off_bits, dbl_bits(off));
    }
  }
  __syncthreads();
  // One Newton-Schulz step R <- R (3 I - R^T R) / 2 (all four waves, one entry per thread):
  // a product of plane rotations is orthogonal only to a few 1e-16 and that defect would
  // add up over the ~300 block rounds of a solve.
  if (threadIdx.x < 256) {
    const int e = threadIdx.x, p = e >> 4, q = e & 15;
    double d = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) d = fma(sR[p * 17 + c], sR[q * 17 + c], d);
    sGam[p * 17 + q] = d;  // N = R^T R  (symmetric)
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int e = threadIdx.x, q = e >> 4, c = e & 15;
    double d = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) d = fma(sR[p * 17 + c], sGam[p * 17 + q], d);
    sN[q * 17 + c] = 1.5 * sR[q * 17 + c] - 0.5 * d;  // column q of R'
  }
  __syncthreads();
  // 4. apply R:  new^T[c'][r] = sum_c R[c][c'] old^T[c][r]
  double Rf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) Rf[s] = sN[lo * 17 + 4 * s + hi];  // R[c = 4s+hi][c' = lo]
  const int ntiles = LD / 16;
  int job = wave;
  for (; job + JB_WAVES < ntiles; job += 2 * JB_WAVES) {  // two row tiles per trip
    const int r0 = job * 16, r1 = (job + JB_WAVES) * 16;
    d4 o0 = {0.0, 0.0, 0.0, 0.0}, o1 = o0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      o0 = mfma_f64(Rf[s], sG[(4 * s + hi) * RS + r0 + lo], o0);
      o1 = mfma_f64(Rf[s], sG[(4 * s + hi) * RS + r1 + lo], o1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double *dst = Gc + (size_t)gcol(hi + 4 * r) * LD + lo;
      dst[r0] = o0[r];
      dst[r1] = o1[r];
    }
  }
  for (; job < ntiles; job += JB_WAVES) {
    const int r0 = job * 16;
    d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) o = mfma_f64(Rf[s], sG[(4 * s + hi) * RS + r0 + lo], o);
#pragma unroll
    for (int r = 0; r < 4; ++r) Gc[(size_t)gcol(hi + 4 * r) * LD + r0 + lo] = o[r];
  }
}

// End of a sweep: state[0] = largest cosine met before its rotation during the sweep.  Quadratic
// convergence: a sweep that STARTS below `tol` ends at rounding level, so it was the last one.
// state[1] = done (1) / non-finite input (2); state[2] = sweeps run; state[8 + k] = cosine of sweep k.
__global__ void lgj_check(unsigned long long *state, double tol) {
  if (threadIdx.x != 0 || state[1] != 0ull) return;
  const unsigned long long bits = state[0];
  double off;
  memcpy(&off, &bits, sizeof off);
  const unsigned long long k = state[2];
  if (k < 48) state[8 + k] = bits;
  state[2] = k + 1;
  state[0] = 0ull;
  if (!(off == off) || off > 1e300) state[1] = 2ull;   // NaN / inf
  else if (off <= tol) state[1] = 1ull;
}

// ---- first-order final sweep ------------------------------------------------------------
// Once every cosine is <= 1e-8 the remaining rotations are tiny and commute to first order, so
// the last ("verification") sweep -- LD/8 + 1 launches -- is replaced by
//   Gamma = G^T G (GEMM);  X_ij = Gamma_ij / (Gamma_jj - Gamma_ii)  (the small-angle limit of the
//   Jacobi rotation of pair (i, j); X is antisymmetric);  R = I + X + X^2 / 2 = I + X - X^T X / 2
//   (GEMM; orthogonal to O(|X|^3));  G <- G R (GEMM).
// lgx_build also measures what decides whether this is legitimate: the largest cosine c and
// max_i sum_j |X_ij| >= |X|_2 (near-degenerate neighbours amplify angles by ~ sigma / gap):
//   |X| <= 1e-5: R as above;  |X| <= 2e-3: R = exp(X) to 4th order (two more GEMMs, orthogonal to
//   |X|^5 / 120 < 1e-15);  larger: the caller runs an ordinary Jacobi sweep instead.
// The sweep is the last one when it started with c <= 1e-8 (it ends at rounding level, like a
// Jacobi sweep); otherwise the residual is ~ |X| c and another first-order sweep follows.
__global__ void lgx_transpose(int LD, const double *Gc, double *Gr, unsigned long long *zero4 = nullptr) {
  __shared__ double tile[32][33];
  // (also clears the four statistics words lgx_build accumulates into: saves the memset launches)
  if (zero4 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0ull;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int i = i0 + r, j = j0 + threadIdx.x;
    tile[r][threadIdx.x] = (i < LD && j < LD) ? Gc[(size_t)i * LD + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int j = j0 + r, i = i0 + threadIdx.x;
    if (j < LD && i < LD) Gr[(size_t)j * LD + i] = tile[threadIdx.x][r];
  }
}

// One wavefront per row i (4 rows per workgroup); state[4] = max cosine (bits), state[5] = max row sum (bits).
// Xf = X with the pairs at block distance <= band zeroed (the hybrid sweep rotates those exactly,
// by banded Jacobi rounds); state[6] = its max row sum.  state[7] counts finished workgroups: the
// last one publishes the three statistics + `seq` to `poll` (coherent pinned host memory), where the
// host is spinning -- a hipStreamSynchronize round trip costs ~30 us of idle GPU per sweep.
__global__ __launch_bounds__(256) void lgx_build(int LD, const double *Gam, const double *dg, double *X, double *Xf,
                                                 int band, unsigned long long *state,
                                                 volatile unsigned long long *poll, unsigned long long seq,
                                                 int hybrid_ok, double trigger, unsigned long long *part) {
  __shared__ double s0[256], s1[256], s2[256];
  __shared__ int s_last;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  double mc2 = 0.0, rs = 0.0, rsf = 0.0;
  if (i < LD) {
    const double gii = dg[i];   // diagonal of Gam, written by the Gram product's epilogue
    const int bi = i / JB_W;
    for (int j0 = lane; j0 < LD; j0 += 256) {   // four column chunks in flight
      double g[4], gj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + 64 * u, LD - 1);
        g[u] = Gam[(size_t)i * LD + j];
        gj[u] = dg[j];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + 64 * u;
        if (j >= LD) continue;
        double x = 0.0;
        const double g2 = g[u] * g[u], ab = gii * gj[u];
        if (!(g2 == g2) || !(ab == ab)) mc2 = INFINITY;   // NaN in G: reported as an infinite cosine (fmax drops NaNs)
        if (j != i && g2 > ab * (CB_JAC_SKIP * CB_JAC_SKIP)) {
          mc2 = fmax(mc2, g2 * fast_rcp(ab));
          const double d = gj[u] - gii;
          // (fast_rcp is odd in its argument, so X stays exactly antisymmetric)
          x = d != 0.0 ? g[u] * fast_rcp(d) : (g[u] > 0.0 ? 1.0 : -1.0);  // exactly degenerate and coupled: refuse (huge row sum)
          rs += fabs(x);
        }
        X[(size_t)i * LD + j] = x;
        const int bd = bi - j / JB_W;
        const bool far = bd > band || -bd > band;
        Xf[(size_t)i * LD + j] = far ? x : 0.0;
        rsf += far ? fabs(x) : 0.0;
      }
    }
  }
  mc2 = wave_max(mc2);
  rs = wave_sum(rs);
  rsf = wave_sum(rsf);
  if (lane == 0) {
    s0[wave] = mc2;
    s1[wave] = rs;
    s2[wave] = rsf;
  }
  __syncthreads();
  // Per-workgroup maxima go to `part` (plain stores), ONE atomic per workgroup counts them in, and the
  // last workgroup reduces the array with all its threads (three atomicMax per workgroup on the same
  // words were serialised across the XCDs: most of this kernel's 12 us).
  if (threadIdx.x == 0) {
    part[3 * blockIdx.x + 0] = dbl_bits(fmax(fmax(s0[0], s0[1]), fmax(s0[2], s0[3])));
    part[3 * blockIdx.x + 1] = dbl_bits(fmax(fmax(s1[0], s1[1]), fmax(s1[2], s1[3])));
    part[3 * blockIdx.x + 2] = dbl_bits(fmax(fmax(s2[0], s2[1]), fmax(s2[2], s2[3])));
    __threadfence();
    s_last = atomicAdd(state + 7, 1ull) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double m0 = 0.0, m1 = 0.0, m2 = 0.0;
  for (int w = threadIdx.x; w < (int)gridDim.x; w += 256) {   // agent-scope loads: the other XCDs' stores
    const unsigned long long b0 = __hip_atomic_load(part + 3 * w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b1 = __hip_atomic_load(part + 3 * w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b2 = __hip_atomic_load(part + 3 * w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double v0, v1, v2;
    memcpy(&v0, &b0, 8);
    memcpy(&v1, &b1, 8);
    memcpy(&v2, &b2, 8);
    m0 = fmax(m0, v0);
    m1 = fmax(m1, v1);
    m2 = fmax(m2, v2);
  }
  s0[threadIdx.x] = m0;
  s1[threadIdx.x] = m1;
  s2[threadIdx.x] = m2;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) {
      s0[threadIdx.x] = fmax(s0[threadIdx.x], s0[threadIdx.x + st]);
      s1[threadIdx.x] = fmax(s1[threadIdx.x], s1[threadIdx.x + st]);
      s2[threadIdx.x] = fmax(s2[threadIdx.x], s2[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cosmax = sqrt(s0[0]), rowsum = s1[0];
    // which X the sweep rotates with (the host applies the same rule to the same numbers): far
    // pairs only while the state is too far for the small-angle limit to hold for near neighbours
    const unsigned long long sel = (hybrid_ok && (cosmax > trigger || rowsum > 0.5)) ? 1ull : 0ull;
    state[3] = sel;              // read by the products enqueued behind this kernel (K4Args::sel)
    state[4] = dbl_bits(cosmax);   // (the copy route of CB_NO_POLL reads these three)
    state[5] = dbl_bits(rowsum);
    state[6] = dbl_bits(s2[0]);
    state[7] = 0ull;             // the counter, ready for the next sweep
    if (poll) {
      poll[1] = dbl_bits(cosmax);
      poll[2] = dbl_bits(rowsum);
      poll[3] = dbl_bits(s2[0]);
      poll[4] = sel;
      __threadfence_system();
      poll[0] = seq;
      __threadfence_system();
    }
  }
}

// exp(Y), Y = sc X, to 8th order by Paterson-Stockmeyer:  exp(Y) ~ lo + hi Y^4,
//   lo = I + Y + Y^2/2 + Y^3/6,  hi = I/4! + Y/5! + Y^2/6! + Y^3/7! + Y^4/8!.
// In: X (antisymmetric), P2 = X^T X = -X^2, P3 = X^T P2 = X^3, P4 = P2^T P2 = X^4.
// Out: lo and hi^T (X, P3 change sign under transposition, P2, P4 do not); the caller forms
// R = lo + sc^4 (hi^T)^T P4 with one more sg_gemm.  |Y| <= 0.075 keeps the remainder |Y|^9/9! < 1e-15.
__global__ void lgx_poly8(int LD, double sc, const double *X, const double *P2, const double *P3, const double *P4,
                          double *lo, double *hiT) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)LD * LD) return;
  const int i = idx / LD, j = idx - (size_t)i * LD;
  const double dl = i == j ? 1.0 : 0.0;
  const double s2 = sc * sc;
  const double y1 = sc * X[idx], y2 = -s2 * P2[idx], y3 = s2 * sc * P3[idx], y4 = s2 * s2 * P4[idx];
  lo[idx] = dl + y1 + 0.5 * y2 + y3 * (1.0 / 6.0);
  hiT[idx] = dl * (1.0 / 24.0) - y1 * (1.0 / 120.0) + y2 * (1.0 / 720.0) - y3 * (1.0 / 5040.0) + y4 * (1.0 / 40320.0);
}

// R = I + X - P2/2 + P3/6 + P4/24  (= exp(X) to 4th order: P2 = X^T X = -X^2, P3 = X^T P2 = X^3,
// P4 = P2^T P2 = X^4); P3 == nullptr: second order, R = I + X - P2/2.
__global__ void lgx_combine(int LD, const double *X, const double *P2, const double *P3, const double *P4, double *R) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)LD * LD) return;
  const int i = idx / LD, j = idx - (size_t)i * LD;
  double v = (i == j ? 1.0 : 0.0) + X[idx] - 0.5 * P2[idx];
  if (P3) v += P3[idx] * (1.0 / 6.0) + P4[idx] * (1.0 / 24.0);
  R[idx] = v;
}

// |g_k| per column (one wave per column)
__global__ void lgj_norms(int LD, const double *Gc, double *nrm) {
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= LD) return;
  double nn = 0.0;
  for (int r = lane; r < LD; r += 64) {
    const double v = Gc[(size_t)k * LD + r];
    nn = fma(v, v, nn);
  }
  nn = wave_sum(nn);
  if (lane == 0) nrm[k] = sqrt(nn);
}

// Columns re-ordered by descending norm (= ascending eigenvalue once the columns are nearly orthogonal):
// a cold solve switches from tournament sweeps to hybrid sweeps through this, because the hybrid
// sweep finds the near-degenerate pairs by their distance in that order.
__global__ void lgj_sort_columns(int LD, const double *Gc, const double *nrm, double *out) {
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= LD) return;
  const double mine = nrm[k];
  int before = 0;
  for (int j = lane; j < LD; j += 64) {
    const double o = nrm[j];
    before += (o > mine || (o == mine && j < k)) ? 1 : 0;
  }
  const int pos = (int)wave_sum((double)before);
  for (int r = lane; r < LD; r += 64) out[(size_t)pos * LD + r] = Gc[(size_t)k * LD + r];
}

// lam = sigma - |g_k| ; Ut[pos][r] = U[r][pos] = -g_k[r] / |g_k|   (A' negative definite), with
// pos = rank of |g_k| in descending order (eigenvalues ascending).  Sorted output keeps
// near-degenerate eigenvectors in neighbouring columns, i.e. inside one 16-column group of the
// next (warm-started) solve, where the within pass resolves the whole cluster at once.
__global__ void lgj_finish(int LD, const double *Gc, const double *nrm, const double *sigma, double *lam,
                           double *U, double *Ut) {
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= LD) return;
  const double mine = nrm[k];
  int before = 0;
  for (int j = lane; j < LD; j += 64) {
    const double o = nrm[j];
    before += (o > mine || (o == mine && j < k)) ? 1 : 0;
  }
  before = (int)wave_sum((double)before);
  const int pos = before;
  const double inv = -1.0 / mine;
  for (int r = lane; r < LD; r += 64) {
    const double v = Gc[(size_t)k * LD + r] * inv;
    Ut[(size_t)pos * LD + r] = v;
    U[(size_t)r * LD + pos] = v;
  }
  if (lane == 0) lam[pos] = *sigma - mine;
}
