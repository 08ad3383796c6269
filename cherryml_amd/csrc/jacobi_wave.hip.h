// One-wavefront symmetric eigensolver for n <= 32, matrices in LDS.
//
// One-sided (Hestenes) Jacobi on the SHIFTED matrix A' = A - sigma I with
// sigma >= max |A_ii| (> 0): a symmetrised rate matrix is negative
// semidefinite with a zero eigenvalue, so A' is negative definite with
// condition number <= ~3 and every column of G = A' V keeps a healthy norm.
//
// V is never formed: when the columns of G are mutually orthogonal, V
// diagonalises A'^2, hence (A' definite) A' itself, so g_k = lambda'_k v_k and
//     U[:,k] = sign * g_k / |g_k|,   lambda_k = sign * |g_k| + sigma
// (sign = -1 for a negative definite A', +1 for a positive definite one).
// The orthogonality of U is then the orthogonality the sweeps converged to,
// not an accumulated product of rotations.
//
// Layout: Gc[k*LS + r] holds COLUMN k (component r).  The n/2 disjoint pairs
// of a round (round-robin tournament) are handled by 4 lanes each (lanes
// 4p .. 4p+3 take rows sub, sub+4, ...; at most 8 rows per lane).
#pragma once
#include "common.hip.h"

// A rotation is skipped when the pair is already orthogonal to rounding; the
// sweeps stop after the first sweep whose largest PRE-rotation cosine was
// below CB_JAC_STOP (quadratic convergence: that sweep leaves ~CB_JAC_STOP^2 x sigma/gap; with
// near-degenerate sites 1e-8 was measured to leave 1e-12 and to drift trajectories by 1e-10).
#define CB_JAC_SKIP 1e-16
#define CB_JAC_STOP 1e-11
#define CB_JAC_MAX_SWEEPS 40

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// circle-method pairing: round r in [0, m-1), slot i in [0, m/2), m even.
__device__ __forceinline__ void rr_pair(int m, int r, int i, int &p, int &q) {
  const int mm = m - 1;
  if (i == 0) {
    p = mm;
    q = r % mm;
  } else {
    p = (r + i) % mm;
    q = (r - i + mm) % mm;
  }
}

// Orthogonalises the columns of Gc in place.  RPL = rows per lane (ceil(n/4)),
// a compile-time bound so that the column pieces stay in registers.
// Returns the number of sweeps used.
// If Vc != nullptr the same rotations are applied to the columns of Vc (the
// caller initialises it, e.g. to I), so a truncated run still yields an exactly
// orthogonal accumulated transform.
template <int RPL, bool ACC>
__device__ int wave_jacobi_columns(int n, double *Gc, double *Vc, int LS, int max_sweeps) {
  const int lane = threadIdx.x & 63;
  const int m = (n + 1) & ~1;  // even number of players (one dummy when n is odd)
  const int mm = m - 1;
  const int slot = lane >> 2, sub = lane & 3;
  const bool active = slot < (m >> 1);
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    double off = 0.0;
    // positions advance by one per round: p = (r + slot) % mm, q = (r - slot) % mm
    int p = slot % mm, q = (mm - slot) % mm;
    for (int r = 0; r < mm; ++r) {
      const int pp = slot == 0 ? mm : p, qq = slot == 0 ? r : q;
      const bool real = active && pp < n && qq < n;
      // unconditional loads from clamped (always valid) addresses + selects:
      // guarded loads make hipcc emit one exec-mask branch region per load
      double x[RPL], y[RPL], u[RPL], v[RPL];
      double a = 0.0, b = 0.0, g = 0.0;
      const int pbase = (real ? pp : 0) * LS, qbase = (real ? qq : 0) * LS;
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const int row = min(sub + 4 * i, n - 1);
        x[i] = Gc[pbase + row];
        y[i] = Gc[qbase + row];
        if (ACC) {
          u[i] = Vc[pbase + row];
          v[i] = Vc[qbase + row];
        }
      }
#pragma unroll
      for (int i = 0; i < RPL; ++i) {
        const bool ok = real && (sub + 4 * i < n);
        x[i] = ok ? x[i] : 0.0;
        y[i] = ok ? y[i] : 0.0;
        a = fma(x[i], x[i], a);
        b = fma(y[i], y[i], b);
        g = fma(x[i], y[i], g);
      }
      a = quad_sum(a);
      b = quad_sum(b);
      g = quad_sum(g);
      if (real) {
        const double ab = a * b;
        // cosine^2 = g^2 / (a b); compare squares, no sqrt / div on this path
        const double g2 = g * g;
        if (g2 > ab * (CB_JAC_STOP * CB_JAC_STOP)) off = 1.0;  // "not converged" flag
        if (g2 > ab * (CB_JAC_SKIP * CB_JAC_SKIP)) {
          // half-angle form (|theta| <= pi/4): cos 2theta = |d| / h, sin 2theta = sign(d) 2 g / h, d = b - a,
          // c = sqrt((1 + cos 2theta) / 2), s = sin 2theta / (2 c): two dependent rsqrt
          const double d = b - a;
          const double hh = fma(d, d, 4.0 * g2);   // > 0 here
          const double rh = fast_rsqrt(hh);
          const double xx = fma(0.5 * fabs(d), rh, 0.5);
          const double rx = fast_rsqrt(xx);
          const double c = xx * rx;
          const double s = copysign(g * rh * rx, g * d);
#pragma unroll
          for (int i = 0; i < RPL; ++i) {
            const int row = sub + 4 * i;
            if (row < n) {
              Gc[pp * LS + row] = c * x[i] - s * y[i];
              Gc[qq * LS + row] = s * x[i] + c * y[i];
              if (ACC) {
                Vc[pp * LS + row] = c * u[i] - s * v[i];
                Vc[qq * LS + row] = s * u[i] + c * v[i];
              }
            }
          }
        }
      }
      wave_lds_fence();
      p = (p + 1 == mm) ? 0 : p + 1;
      q = (q + 1 == mm) ? 0 : q + 1;
    }
    off = wave_max(off);
    if (off == 0.0) {
      ++sweeps;
      break;
    }
  }
  return sweeps;
}

__device__ __forceinline__ int wave_jacobi_columns_n(int n, double *Gc, int LS, int max_sweeps) {
  if (n <= 4) return wave_jacobi_columns<1, false>(n, Gc, nullptr, LS, max_sweeps);
  if (n <= 8) return wave_jacobi_columns<2, false>(n, Gc, nullptr, LS, max_sweeps);
  if (n <= 16) return wave_jacobi_columns<4, false>(n, Gc, nullptr, LS, max_sweeps);
  if (n <= 20) return wave_jacobi_columns<5, false>(n, Gc, nullptr, LS, max_sweeps);
  if (n <= 24) return wave_jacobi_columns<6, false>(n, Gc, nullptr, LS, max_sweeps);
  return wave_jacobi_columns<8, false>(n, Gc, nullptr, LS, max_sweeps);
}

// Eigendecomposition of a symmetric matrix A (row-major, stride LS, read only)
// that is negative semidefinite up to rounding (symmetrised rate matrix):
//   Uc[k*LS + i] = component i of eigenvector k,  lam[k].
// Gc is scratch (n x LS).  Returns the sweeps used.
// warm = true: Uc holds an orthonormal basis (the eigenvectors of a nearby matrix,
// e.g. the previous optimiser epoch): the sweeps start from G0 = A' Uc, whose columns
// are already nearly orthogonal (2-3 sweeps instead of 6-8).
__device__ int wave_eigh_rate(int n, const double *A, double *Gc, double *Uc, double *lam,
                              int LS, bool warm = false) {
  const int lane = threadIdx.x & 63;
  double mx = 0.0;
  for (int i = lane; i < n; i += 64) mx = fmax(mx, fabs(A[i * LS + i]));
  double sigma = wave_max(mx);
  if (!(sigma > 0.0)) sigma = 1.0;
  for (int e = lane; e < n * n; e += 64) {
    const int k = e / n, r = e - k * n;
    double v;
    if (warm) {  // (A' u_k)[r] = sum_j A[r][j] u_k[j] - sigma u_k[r]
      v = -sigma * Uc[k * LS + r];
      for (int j = 0; j < n; ++j) v = fma(A[r * LS + j], Uc[k * LS + j], v);
    } else {
      v = A[r * LS + k] - (r == k ? sigma : 0.0);
    }
    Gc[k * LS + r] = v;
  }
  wave_lds_fence();
  const int sweeps = wave_jacobi_columns_n(n, Gc, LS, CB_JAC_MAX_SWEEPS);
  // normalise: 4 lanes per column
  for (int k0 = 0; k0 < n; k0 += 16) {
    const int k = k0 + (lane >> 2), sub = lane & 3;
    double nn = 0.0;
    if (k < n)
      for (int r = sub; r < n; r += 4) nn = fma(Gc[k * LS + r], Gc[k * LS + r], nn);
    nn += __shfl_xor(nn, 1, 64);
    nn += __shfl_xor(nn, 2, 64);
    if (k < n) {
      const double nrm = sqrt(nn);
      const double inv = -1.0 / nrm;  // A' negative definite: g_k = lambda'_k v_k, lambda'_k < 0
      for (int r = sub; r < n; r += 4) Uc[k * LS + r] = Gc[k * LS + r] * inv;
      if (sub == 0) lam[k] = sigma - nrm;
    }
  }
  wave_lds_fence();
  return sweeps;
}

// Orthogonal R (16 x 16) that (approximately, after `max_sweeps` Jacobi sweeps)
// diagonalises the symmetric positive definite Gamma: columns of R at
// Rc[k*LS + i].  R is the accumulated product of the plane rotations, hence
// orthogonal however few sweeps are run.  Gamma is destroyed.
__device__ int wave_rotation_spd16(double *Gam, double *Rc, int LS, int max_sweeps) {
  const int lane = threadIdx.x & 63;
  for (int e = lane; e < 256; e += 64) Rc[(e >> 4) * LS + (e & 15)] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
  wave_lds_fence();
  return wave_jacobi_columns<4, true>(16, Gam, Rc, LS, max_sweeps);
}

// One pass over a chosen family of pairs of the 16 columns (same rotation arithmetic as
// wave_jacobi_columns):
//   cross = true : the 64 pairs (i, 8 + j) between the two 8-column blocks, as 8
//                  perfect matchings  i <-> 8 + (i + r) % 8
//   cross = false: the 2 x 28 pairs inside each block, 7 tournament rounds each
// All 64 lanes work: 8 lanes per pair, 2 of the 16 rows per lane (rows sub, sub + 8).
// Returns (wave-uniform) the largest squared cosine g^2 / (a b) met BEFORE its rotation.
__device__ double wave_rotation_spd16_blockpairs(double *Gc, double *Vc, int LS, bool cross) {
  const int lane = threadIdx.x & 63;
  const int slot = lane >> 3, sub = lane & 7;
  const int nrounds = cross ? 8 : 7;
  double off2 = 0.0;
  for (int r = 0; r < nrounds; ++r) {
    int pp, qq;
    if (cross) {
      pp = slot;
      qq = 8 + ((slot + r) & 7);
    } else {
      // slots 0-3: block 0, slots 4-7: block 1; 8 players each, circle method
      const int i = slot & 3, base = (slot & 4) ? 8 : 0;
      int a, b;
      rr_pair(8, r, i, a, b);
      pp = base + a;
      qq = base + b;
    }
    const int pbase = pp * LS + sub, qbase = qq * LS + sub;
    const double x0 = Gc[pbase], x1 = Gc[pbase + 8], y0 = Gc[qbase], y1 = Gc[qbase + 8];
    const double u0 = Vc[pbase], u1 = Vc[pbase + 8], v0 = Vc[qbase], v1 = Vc[qbase + 8];
    const double a = oct_sum(fma(x1, x1, x0 * x0));
    const double b = oct_sum(fma(y1, y1, y0 * y0));
    const double g = oct_sum(fma(x1, y1, x0 * y0));
    const double g2 = g * g, ab = a * b;
    if (g2 > ab * (CB_JAC_SKIP * CB_JAC_SKIP)) {
      off2 = fmax(off2, g2 * fast_rcp(ab));
      const double d = b - a;
      const double hh = fma(d, d, 4.0 * g2);
      const double h = hh * fast_rsqrt(hh);
      const double den = d + copysign(h, d);
      const double t = 2.0 * g * copysign(fast_rcp(fabs(den)), den);
      const double c = fast_rsqrt(fma(t, t, 1.0));
      const double s = c * t;
      Gc[pbase] = c * x0 - s * y0;
      Gc[pbase + 8] = c * x1 - s * y1;
      Gc[qbase] = s * x0 + c * y0;
      Gc[qbase + 8] = s * x1 + c * y1;
      Vc[pbase] = c * u0 - s * v0;
      Vc[pbase + 8] = c * u1 - s * v1;
      Vc[qbase] = s * u0 + c * v0;
      Vc[qbase + 8] = s * u1 + c * v1;
    }
    wave_lds_fence();
  }
  return wave_max(off2);
}

// The same 8 cross matchings as wave_rotation_spd16_blockpairs(cross = true), but as a TWO-sided
// Jacobi on Gamma held in registers (Brent-Luk systolic layout): lane 8 I + J owns the 2 x 2 block
// (rows p_I, q_I) x (columns p_J, q_J) of Gamma, p_I = I, q_I = 8 + (I + r) % 8 in matching r.  The
// diagonal lanes (I == J) see [[a, g], [g, b]] of their pair and compute its rotation; (c, s)
// travel along the block row and block column by lane shuffles; every lane updates its block
// from both sides; then the q rows / columns move one block up / left, which is the next
// matching.  No LDS round trip and no reduction inside the loop (the one-sided form needs three
// 8-lane reductions and an LDS fence per matching).  R (rows 2I, 2I+1 x columns p_J, q_J per lane)
// accumulates the column rotations.  After 8 matchings the layout is back where it started.
// Gamma is well conditioned here (kappa(A')^2 <= ~9), so working on it directly loses nothing.
// Returns (wave-uniform) the largest squared cosine g^2 / (a b) met before its rotation.
__device__ double wave_rotation_cross16_regs(const double *Gam, double *Rc, int LS) {
  const int lane = threadIdx.x & 63;
  const int I = lane >> 3, J = lane & 7;
  double g00 = Gam[I * LS + J], g01 = Gam[I * LS + 8 + J];
  double g10 = Gam[(8 + I) * LS + J], g11 = Gam[(8 + I) * LS + 8 + J];
  // R = identity: rows 2I, 2I+1; columns J, 8 + J
  double r00 = (2 * I == J) ? 1.0 : 0.0, r01 = (2 * I == 8 + J) ? 1.0 : 0.0;
  double r10 = (2 * I + 1 == J) ? 1.0 : 0.0, r11 = (2 * I + 1 == 8 + J) ? 1.0 : 0.0;
  const int diagI = 9 * I, diagJ = 9 * J;
  const int right = (lane & ~7) | ((J + 1) & 7);                 // (I, J+1)
  const int down = (((I + 1) & 7) << 3) | J;                     // (I+1, J)
  const int diag = (((I + 1) & 7) << 3) | ((J + 1) & 7);         // (I+1, J+1)
  double off2 = 0.0;
#pragma unroll 1
  for (int r = 0; r < 8; ++r) {
    // rotation of "my" pair -- meaningful on the diagonal lanes only
    const double a = g00, b = g11, g = g01;
    const double g2 = g * g, ab = a * b;
    double c = 1.0, s = 0.0;
    if (g2 > ab * (CB_JAC_SKIP * CB_JAC_SKIP)) {
      if (I == J) off2 = fmax(off2, g2 * fast_rcp(ab));
      // half-angle form (|theta| <= pi/4): cos 2theta = |d| / h, sin 2theta = sign(d) 2g / h,
      // c = sqrt((1 + cos 2theta) / 2), s = sin 2theta / (2c): two dependent rsqrt instead of
      // rsqrt -> rcp -> rsqrt
      const double d = b - a;
      const double hh = fma(d, d, 4.0 * g2);
      const double rh = fast_rsqrt(hh);
      const double x = fma(0.5 * fabs(d), rh, 0.5);
      const double rx = fast_rsqrt(x);
      c = x * rx;
      s = copysign(g * rh * rx, g * d);
    }
    const double cI = __shfl(c, diagI), sI = __shfl(s, diagI);
    const double cJ = __shfl(c, diagJ), sJ = __shfl(s, diagJ);
    // columns:  [p' q'] = [p q] [[c, s], [-s, c]]
    double x00 = cJ * g00 - sJ * g01, x01 = sJ * g00 + cJ * g01;
    double x10 = cJ * g10 - sJ * g11, x11 = sJ * g10 + cJ * g11;
    const double n00 = cJ * r00 - sJ * r01, n01 = sJ * r00 + cJ * r01;
    const double n10 = cJ * r10 - sJ * r11, n11 = sJ * r10 + cJ * r11;
    // rows: the transposed rotation from the left
    g00 = cI * x00 - sI * x10;
    const double y01 = cI * x01 - sI * x11;
    const double y10 = sI * x00 + cI * x10;
    const double y11 = sI * x01 + cI * x11;
    // next matching: q rows come from the block row below, q columns from the block column right
    g01 = __shfl(y01, right);
    g10 = __shfl(y10, down);
    g11 = __shfl(y11, diag);
    r00 = n00;
    r10 = n10;
    r01 = __shfl(n01, right);
    r11 = __shfl(n11, right);
  }
  Rc[J * LS + 2 * I] = r00;
  Rc[J * LS + 2 * I + 1] = r10;
  Rc[(8 + J) * LS + 2 * I] = r01;
  Rc[(8 + J) * LS + 2 * I + 1] = r11;
  wave_lds_fence();
  return wave_max(off2);
}


// ---------------------------------------------------------------------------------------------------------------
// Warm-started FIRST-ORDER sweeps on the matrix cores (one wavefront, n <= 24).  An optimiser step perturbs A a little,
// so the columns of G = A' U_prev are already nearly orthogonal; a whole Jacobi sweep then spends n - 1 dependent
// rounds (LDS round trip, three reductions, two rsqrt each: ~30k cycles at n = 20) on rotations whose angles are all
// small.  Small angles commute to first order, so ALL pairs are rotated at once:
//     Gamma = G^T G,     X_ij = Gamma_ij / (Gamma_jj - Gamma_ii)   (the small-angle limit of every pair's rotation;
//     G <- G exp(X)       antisymmetric),  exp(X) applied to G term by term: T_0 = G, T_k = T_{k-1} X / k
// with as many terms as the bound ||X||_2 <= ||X||_F needs for a remainder below 1e-17 -- so the implicit
// V <- V exp(X) stays orthogonal to rounding.  Each term is ONE n x n product on the matrix cores from LDS.  An
// iteration squares the cosines (like a Jacobi sweep); the one that STARTED below CB_JAC_STOP is the last.  Pairs that
// are nearly degenerate make X large (||X|| > 0.5: the first epochs of an optimisation, equal-rate models): that
// iteration is an exact Jacobi sweep instead.  Layouts: Gc / Tc column-major (column k at k * LS), X row-major.

// smallest Taylor degree m with x^(m+1) / (m+1)! < 1e-17
__device__ __forceinline__ int fo_taylor_degree(double x) {
  const double lim[15] = {4.5e-9, 3.9e-6, 1.24e-4, 1.04e-3, 4.4e-3, 1.26e-2, 2.82e-2, 5.34e-2,
                          9.03e-2, 0.14, 0.2025, 0.279, 0.369, 0.4725, 0.588};
  int m = 1;
#pragma unroll
  for (int i = 0; i < 15; ++i) m += x > lim[i] ? 1 : 0;
  return m;
}

#define CB_FO_MAX_NORM 0.5
#define CB_FO_MAX_ITERS 48

// The products run on v_mfma_f64_4x4x4 (FOUR independent 4 x 4 x 4 products per instruction, 16 cycles): 5 x 5 tiles
// of 4 x 4 at 20 states = 125 tile products = 35 instructions (7 output slots x 5 k-steps, slot s / block b = tile
// 4 s + b) = 560 cycles of matrix pipe per product.  (The first version multiplied a frame padded to 32 x 32 with
// v_mfma_f64_16x16x4: 24 instructions of 64 cycles = 1536 cycles, a third of them useful -- 12.0 us of eigensolver per
// LG epoch late in an optimisation; this form: LG epoch 50.0 -> 45.9 us, SiteRM 1.41 -> 1.36 ms.)
// Lane = 16 q + 4 b + r holds A_b[i = r][k = q], B_b[k = q][j = r], D_b[i = q][j = r] (common.hip.h).
// REQUIRES: every frame (A, Gc, Uc, X) zero outside n x n up to 4 TS rows / columns (no bounds tests on operands).
template <int TS>
struct Mm4 {
  static constexpr int NT = TS * TS, NS = (NT + 3) / 4;
};

template <int TS>
__device__ __forceinline__ void wave_mm4(const double *Af, int ams, int aks, const double *Bf, int bks, int bns,
                                         const int (&ti)[Mm4<TS>::NS], const int (&tj)[Mm4<TS>::NS],
                                         double (&acc)[Mm4<TS>::NS]) {
  constexpr int NS = Mm4<TS>::NS;
  const int lane = threadIdx.x & 63, q = lane >> 4, r = lane & 3;
  double a[NS][TS], b[NS][TS];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int K = 0; K < TS; ++K) {
      a[s][K] = Af[(4 * ti[s] + r) * ams + (4 * K + q) * aks];
      b[s][K] = Bf[(4 * K + q) * bks + (4 * tj[s] + r) * bns];
    }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double c = 0.0;
#pragma unroll
    for (int K = 0; K < TS; ++K) c = mfma4_f64(a[s][K], b[s][K], c);
    acc[s] = c;
  }
}

template <int TS>
__device__ int wave_orthogonalise_first_order4(int n, const double *A, double sigma, double *Gc, double *Uc, double *X,
                                               double *dg, int LS) {
  constexpr int NS = Mm4<TS>::NS, NT = Mm4<TS>::NT;
  const int lane = threadIdx.x & 63, q = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
  int ti[NS], tj[NS], row[NS], col[NS];
  bool live[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int t = 4 * s + blk;
    live[s] = t < NT;
    const int tc = live[s] ? t : NT - 1;   // a dead block repeats the last tile (its result is ignored)
    ti[s] = tc / TS;
    tj[s] = tc - ti[s] * TS;
    row[s] = 4 * ti[s] + q;
    col[s] = 4 * tj[s] + r;
  }
  double acc[NS];
  // ---- G0 = A' U = A U - sigma U
  wave_mm4<TS>(A, LS, 1, Uc, 1, LS, ti, tj, acc);
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (live[s]) Gc[col[s] * LS + row[s]] = fma(-sigma, Uc[col[s] * LS + row[s]], acc[s]);
  wave_lds_fence();
  int it = 0;
  for (; it < CB_FO_MAX_ITERS; ++it) {
    double gam[NS];
    wave_mm4<TS>(Gc, LS, 1, Gc, 1, LS, ti, tj, gam);
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s] && row[s] == col[s]) dg[row[s]] = gam[s];
    wave_lds_fence();
    double cos2 = 0.0, fro2 = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double di = dg[row[s]], dj = dg[col[s]];
      const bool on = live[s] && row[s] < n && col[s] < n && row[s] != col[s];
      const double g = gam[s], g2 = g * g, dd = di * dj;
      const double x = (on && g2 > dd * (CB_JAC_SKIP * CB_JAC_SKIP)) ? g * fast_rcp(dj - di) : 0.0;
      if (on) cos2 = fmax(cos2, g2 * fast_rcp(dd));
      fro2 = fma(x, x, fro2);
      if (live[s]) X[row[s] * LS + col[s]] = x;
    }
    fro2 = wave_sum(fro2);                        // ||X||_2 <= ||X||_F (NaN / inf propagate: the test below fails)
    const double nrm = sqrt(fro2);
    cos2 = wave_max(cos2);
    wave_lds_fence();
    const bool last = cos2 < CB_JAC_STOP * CB_JAC_STOP;
    // (a NaN norm -- an exactly degenerate pair divides by zero above, or the matrix itself is non-finite -- takes the exact
    // branch below: it is what degenerate pairs need, and on a non-finite matrix the iteration count bounds the loop and the
    // loss of the epoch comes out NaN through fast_log_table's argument check)
    if (!(nrm <= CB_FO_MAX_NORM)) {   // near-degenerate pairs: exact rotations for this iteration
      if (n <= 4) wave_jacobi_columns<1, false>(n, Gc, nullptr, LS, 1);
      else if (n <= 8) wave_jacobi_columns<2, false>(n, Gc, nullptr, LS, 1);
      else if (n <= 16) wave_jacobi_columns<4, false>(n, Gc, nullptr, LS, 1);
      else if (n <= 20) wave_jacobi_columns<5, false>(n, Gc, nullptr, LS, 1);
      else wave_jacobi_columns<6, false>(n, Gc, nullptr, LS, 1);
      if (last) {
        ++it;
        break;
      }
      continue;
    }
    const int deg = fo_taylor_degree(nrm);
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = Gc[col[s] * LS + row[s]];
    const double *Tsrc = Gc;
    for (int k = 1; k <= deg; ++k) {
      double t[NS];
      wave_mm4<TS>(Tsrc, 1, LS, X, LS, 1, ti, tj, t);
      const double ik = 1.0 / (double)k;
      wave_lds_fence();   // every lane has read the previous term before it is overwritten
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const double v = t[s] * ik;
        acc[s] += v;
        if (k < deg && live[s]) Uc[col[s] * LS + row[s]] = v;
      }
      wave_lds_fence();
      Tsrc = Uc;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s]) Gc[col[s] * LS + row[s]] = acc[s];
    wave_lds_fence();
    if (last) {
      ++it;
      break;
    }
  }
  // insurance: the iteration count ran out without a sweep that started below CB_JAC_STOP (never observed) -- finish with
  // the plain Jacobi sweeps, which stop by the same rule
  if (it >= CB_FO_MAX_ITERS) it += wave_jacobi_columns_n(n, Gc, LS, CB_JAC_MAX_SWEEPS);
  return it;
}

// n <= 24, frames of at least 4 ceil(n / 4) rows, ZERO outside n x n (the caller pads: see sp_prepare_body).
// (inlined: as a call it saved and restored 103 callee-saved registers through scratch memory, 416 bytes per lane, on the
// one path of the LG epoch that is pure latency)
__device__ __forceinline__ int wave_eigh_rate_warm_mfma4(int n, const double *A, double *Gc, double *Uc, double *X, double *dg, double *lam,
                                         int LS) {
  const int lane = threadIdx.x & 63;
  double mx = 0.0;
  for (int i = lane; i < n; i += 64) mx = fmax(mx, fabs(A[i * LS + i]));
  double sigma = wave_max(mx);
  if (!(sigma > 0.0)) sigma = 1.0;
  int its;
  if (n <= 4) its = wave_orthogonalise_first_order4<1>(n, A, sigma, Gc, Uc, X, dg, LS);
  else if (n <= 8) its = wave_orthogonalise_first_order4<2>(n, A, sigma, Gc, Uc, X, dg, LS);
  else if (n <= 12) its = wave_orthogonalise_first_order4<3>(n, A, sigma, Gc, Uc, X, dg, LS);
  else if (n <= 16) its = wave_orthogonalise_first_order4<4>(n, A, sigma, Gc, Uc, X, dg, LS);
  else if (n <= 20) its = wave_orthogonalise_first_order4<5>(n, A, sigma, Gc, Uc, X, dg, LS);
  else its = wave_orthogonalise_first_order4<6>(n, A, sigma, Gc, Uc, X, dg, LS);
  for (int k0 = 0; k0 < n; k0 += 16) {   // normalise: 4 lanes per column (as wave_eigh_rate)
    const int k = k0 + (lane >> 2), sub = lane & 3;
    double nn = 0.0;
    if (k < n)
      for (int r = sub; r < n; r += 4) nn = fma(Gc[k * LS + r], Gc[k * LS + r], nn);
    nn += __shfl_xor(nn, 1, 64);
    nn += __shfl_xor(nn, 2, 64);
    if (k < n) {
      const double nrm = sqrt(nn);
      const double inv = -1.0 / nrm;
      for (int r = sub; r < n; r += 4) Uc[k * LS + r] = Gc[k * LS + r] * inv;
      if (sub == 0) lam[k] = sigma - nrm;
    }
  }
  wave_lds_fence();
  return its;
}
