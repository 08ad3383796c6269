// The 400-state bank in a TIME BASIS (round 5): the buckets are not independent -- every quantity the bank needs from bucket b
// is a SMOOTH function of its branch length t_b evaluated on the spectrum, and over the quantisation grid those functions span
// a space of dimension ~14 .. 30, not B = 129.
//
//   forward   P_b = I + t_b A + t_b^2 U psi(t_b Lambda) U^T,   psi(t, lam) = phi2(t lam) / t^2           (short branches)
//             b -> psi(t_b, lam) lies, to 1e-17, in the span of its values at ns ~ 14 SKELETON buckets:
//                 psi(t_b, .) = sum_r Ls[b][r] psi(t_{s_r}, .)        =>      P_b = I + t_b A + t_b^2 sum_r Ls[b][r] Psi_r
//             with Psi_r = U psi(t_{s_r} Lambda) U^T: ns products instead of one per bucket, and P_b(i, j) an ELEMENTWISE
//             combination of the Psi_r(i, j) -- relative accuracy of the O(t^2) entries as in the per-bucket form (the factor
//             t_b^2 stays outside the sum).  Long branches (t_b rho_max > 8: P_b = U e^{t_b Lambda} U^T has no tiny entries
//             left, and the family e^{t lam} on that range has nearly full rank) keep their own product: "direct" buckets.
//   backward  M = sum_b (U^T G_b U) o Phi_b,   Phi_b,ij = t_b * mean of e^{t_b mu} over mu in [lam_j, lam_i]:
//             b -> Phi_b,ij / t_b lies in the span of {b -> e^{t_b mu}}, whose skeleton has ng ~ 24 .. 30 buckets:
//                 Phi_b / t_b = sum_r Lg[b][r] Phi_{g_r} / t_{g_r}
//             =>  M = sum_r (U^T Gh_r U) o Phi_{g_r},      Gh_r = sum_b (Lg[b][r] t_b / t_{g_r}) G_b
//             i.e. the two products per bucket of the gradient (K2, K3) run on ng VIRTUAL buckets Gh_r at the skeleton times.
//
// The interpolation matrices Ls, Lg depend on the grid t and on a bound rho_max of the spectral radius only -- not on the
// spectrum: they are an interpolative decomposition (pivoted Gram-Schmidt on the rows of the sampled family, long double) built
// on the HOST, once per optimisation (cb_tb_build; rebuilt when 2 max|Q_ii| >= rho, the Gershgorin bound, leaves the range it was
// built for).  Between the products sits ONE elementwise kernel (tb_ew): element (i, j) of all B buckets per thread --
// P_b(i, j), its logarithm and reciprocal, the loss term, G_b(i, j) and its ng accumulations -- reading the counts once.
//
// MFMA tiles of an evaluation: (ns + nd) 15 + ng 40 instead of 40 B (bench bank, B = 129: ~1 900 instead of 5 160); accuracy
// of dL/dA against the per-bucket form 6e-13 .. 5e-12 (both are equally far, 7e-13 .. 7e-11, from a long-double evaluation of P_b:
// the eigendecomposition's own backward error dominates; profiles/tools/tb_prototype.py, EXPERIMENTS section 13).
#pragma once
#include "cb_internal.hip.h"
#include "common.hip.h"
#include <hip/hip_ext.h>

// ------------------------------------------------------------------ host: the interpolative decomposition
namespace tbasis {

typedef long double ld;

// phi2(x) / x^2 = sum_k x^k / (k + 2)!   (x <= 0)
static inline ld g_phi2(ld x) {
  if (fabsl(x) < 0.5L) {
    ld p = 1.0L;
    for (int k = 24; k >= 3; --k) p = 1.0L + p * x / (ld)k;
    return 0.5L * p;
  }
  return (expm1l(x) - x) / (x * x);
}

// dot product with four independent partial sums (x87 adds are 3-5 cycles deep: one running sum is a latency chain)
static inline ld dot4(const ld *a, const ld *b, int n) {
  ld s0 = 0.0L, s1 = 0.0L, s2 = 0.0L, s3 = 0.0L;
  int i = 0;
  for (; i + 3 < n; i += 4) {
    s0 += a[i] * b[i];
    s1 += a[i + 1] * b[i + 1];
    s2 += a[i + 2] * b[i + 2];
    s3 += a[i + 3] * b[i + 3];
  }
  for (; i < n; ++i) s0 += a[i] * b[i];
  return (s0 + s1) + (s2 + s3);
}

// Rows of F [B][N] -> skeleton rows and L [B][R] with F ~ L F[skel]: pivoted modified Gram-Schmidt on the rows (each new
// direction re-orthogonalised twice), stopped when the largest remaining row is below tol * the largest row.  Returns the rank,
// or -1 when Rmax rows do not reach the tolerance.
static int id_rows(int B, int N, const std::vector<ld> &F, int Rmax, ld tol, std::vector<int> &skel, std::vector<ld> &L, ld &resid) {
  std::vector<ld> W(F), Q((size_t)Rmax * N, 0.0L), Cq((size_t)B * Rmax, 0.0L), q(N);
  skel.clear();
  ld scale = 0.0L;
  int R = 0;
  bool done = false;
  for (int r = 0; r <= Rmax; ++r) {
    int p = -1;
    ld best = -1.0L;
    for (int b = 0; b < B; ++b) {
      const ld *w = &W[(size_t)b * N];
      const ld s = dot4(w, w, N);
      if (s > best) best = s, p = b;
    }
    if (r == 0) scale = sqrtl(best);
    if (!(scale > 0.0L) || sqrtl(best) <= tol * scale) {
      done = true;
      break;
    }
    if (r == Rmax) break;
    const ld inv = 1.0L / sqrtl(best);
    for (int i = 0; i < N; ++i) q[i] = W[(size_t)p * N + i] * inv;
    for (int pass = 0; pass < 2; ++pass) {
      for (int k = 0; k < r; ++k) {
        const ld *qk = &Q[(size_t)k * N];
        const ld d = dot4(qk, q.data(), N);
        for (int i = 0; i < N; ++i) q[i] -= d * qk[i];
      }
      const ld s = dot4(q.data(), q.data(), N);
      const ld in = 1.0L / sqrtl(s);
      for (int i = 0; i < N; ++i) q[i] *= in;
    }
    for (int i = 0; i < N; ++i) Q[(size_t)r * N + i] = q[i];
    for (int b = 0; b < B; ++b) {
      ld *w = &W[(size_t)b * N];
      const ld c = dot4(w, q.data(), N);
      Cq[(size_t)b * Rmax + r] = c;
      for (int i = 0; i < N; ++i) w[i] -= c * q[i];
    }
    skel.push_back(p);
    R = r + 1;
  }
  if (!done) return -1;
  // F[skel] = T Q with T[k][r] = Cq[skel[k]][r] (zero for r > k);  L T = Cq, last column first
  L.assign((size_t)B * std::max(R, 1), 0.0L);
  for (int r = R - 1; r >= 0; --r) {
    const ld trr = Cq[(size_t)skel[r] * Rmax + r];
    for (int b = 0; b < B; ++b) {
      ld s = Cq[(size_t)b * Rmax + r];
      for (int k = r + 1; k < R; ++k) s -= L[(size_t)b * R + k] * Cq[(size_t)skel[k] * Rmax + r];
      L[(size_t)b * R + r] = s / trr;
    }
  }
  for (int r = 0; r < R; ++r)   // a skeleton row is itself
    for (int k = 0; k < R; ++k) L[(size_t)skel[r] * R + k] = k == r ? 1.0L : 0.0L;
  resid = 0.0L;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < N; ++i) {
      ld s = F[(size_t)b * N + i];
      for (int k = 0; k < R; ++k) s -= L[(size_t)b * R + k] * F[(size_t)skel[k] * N + i];
      resid = std::max(resid, fabsl(s));
    }
  return R;
}

}  // namespace tbasis

#ifndef CB_TB_SAMPLES
#define CB_TB_SAMPLES 512      // log-spaced sample points of the spectrum in [-rho_max, 0) (+ the point 0)
#endif
#ifndef CB_TB_TOL_G
#define CB_TB_TOL_G 2e-15L     // largest remaining row of the gradient family / largest row at which its skeleton is complete: entries of
                               // Phi_b / t_b in (0, 1] are then interpolated to ~1e-14, two orders below what dL/dQ carries anyway
                               // (1e-12 on the bench bank); at 1e-16 the bench bank needs 31 virtual buckets instead of 30 -- 775
                               // tiles for the first gradient product, a fourth tile on eight of the 256 CUs, which set its pace
#endif
#ifndef CB_TB_X_SMALL
#define CB_TB_X_SMALL 8.0      // a bucket is expanded in the psi family while t_b rho_max <= this (|t A| <= 8: one digit of
                               // cancellation in I + t A + t^2 Psi at the very end of the basis' range, none where it was built)
#endif

bool cb_tb_build(int B, const double *t, double rho_max, CbTimeBasisHost &out) {
  using namespace tbasis;
  out = CbTimeBasisHost{};
  if (B < 1 || !(rho_max > 0.0) || !std::isfinite(rho_max)) return false;
  for (int b = 0; b < B; ++b)
    if (!(t[b] > 0.0) || !std::isfinite(t[b])) return false;
  const int N = CB_TB_SAMPLES + 1;
  std::vector<ld> mu(N);
  mu[0] = 0.0L;
  for (int i = 1; i < N; ++i) mu[i] = -(ld)rho_max * powl(10.0L, -6.0L + 6.0L * (ld)(i - 1) / (ld)(N - 2));
  out.B = B;
  out.rho_max = rho_max;
  for (int b = 1; b < B; ++b)   // (an ascending grid -- the reference's is: the short-branch buckets are then a prefix)
    if (!(t[b] >= t[b - 1])) return false;
  out.kind.assign(B, -1);
  std::vector<int> small_idx;
  for (int b = 0; b < B; ++b) {
    if (t[b] * rho_max <= CB_TB_X_SMALL) small_idx.push_back(b);
    else {
      out.kind[b] = (int)out.direct.size();
      out.direct.push_back(b);
    }
  }
  out.Ls.assign((size_t)B * CB_TB_RS_MAX, 0.0);
  out.Lg.assign((size_t)B * CB_TB_RG_MAX, 0.0);
  // forward family of the short branches: psi(t, mu) = phi2(t mu) / t^2 = mu^2 g(t mu)
  const int nsm = (int)small_idx.size();
  if (nsm > 0) {
    std::vector<ld> F((size_t)nsm * N);
    for (int k = 0; k < nsm; ++k)
      for (int i = 0; i < N; ++i) F[(size_t)k * N + i] = mu[i] * mu[i] * g_phi2((ld)t[small_idx[k]] * mu[i]);
    std::vector<int> sk;
    std::vector<ld> L;
    ld res = 0.0L;
    const int R = id_rows(nsm, N, F, std::min(CB_TB_RS_MAX, nsm), 1e-17L, sk, L, res);
    if (R < 0) return false;
    out.ns = R;
    out.res_s = (double)res;
    for (int r = 0; r < R; ++r) out.skel_s.push_back(small_idx[sk[r]]);
    for (int k = 0; k < nsm; ++k)
      for (int r = 0; r < R; ++r) out.Ls[(size_t)small_idx[k] * CB_TB_RS_MAX + r] = (double)L[(size_t)k * R + r];
  }
  out.nd = (int)out.direct.size();
  // backward family, all buckets: e^{t mu}; Lg carries the factor t_b / t_skeleton (see the header comment)
  {
    std::vector<ld> F((size_t)B * N);
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < N; ++i) F[(size_t)b * N + i] = expl((ld)t[b] * mu[i]);
    std::vector<int> sk;
    std::vector<ld> L;
    ld res = 0.0L;
    const int R = id_rows(B, N, F, std::min(CB_TB_RG_MAX, B), CB_TB_TOL_G, sk, L, res);
    if (R < 0) return false;
    out.ng = R;
    out.res_g = (double)res;
    out.skel_g = sk;
    for (int b = 0; b < B; ++b)
      for (int r = 0; r < R; ++r)
        out.Lg[(size_t)b * CB_TB_RG_MAX + r] = (double)(L[(size_t)b * R + r] * (ld)t[b] / (ld)t[sk[r]]);
  }
  for (int r = 0; r < out.ns; ++r) out.tf.push_back(t[out.skel_s[r]]);
  for (int r = 0; r < out.nd; ++r) out.tf.push_back(t[out.direct[r]]);
  for (int r = 0; r < out.ng; ++r) out.tg.push_back(t[out.skel_g[r]]);
  return true;
}

// ------------------------------------------------------------------ device
// spectral tables of the virtual buckets: F [ns + nd][LD] (psi(t lam) for the skeleton of the short branches, e^{t lam} for
// the direct buckets), E / H [ng][LD] (e^{t lam}, e^{t lam / 2} at the gradient skeleton, what k3_w_phi's divided
// differences read)
__global__ void tb_tables(int LD, int ns, int nd, int ng, const double *__restrict__ tf, const double *__restrict__ tg,
                          const double *__restrict__ lam, double *__restrict__ F, double *__restrict__ E, double *__restrict__ H,
                          const unsigned long long *skip) {
  if (skip && *skip != 0ull) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = ns + nd;
  if (idx >= (nf + ng) * LD) return;
  const int r = idx / LD, k = idx - r * LD;
  if (r < nf) {
    const double t = tf[r], x = t * lam[k];
    F[idx] = r < ns ? phi2(x) / (t * t) : exp(x);
  } else {
    const double x = tg[r - nf] * lam[k];
    E[idx - nf * LD] = exp(x);
    H[idx - nf * LD] = exp(0.5 * x);
  }
}

// The elementwise kernel.  A wavefront owns 16 consecutive elements (one row, columns c0 .. c0 + 15) of ALL buckets; lane
// (j = l & 15, k = l >> 4) works on element j and, in every step of 16 buckets b0 .. b0 + 15, on the four buckets b0 + k + 4 m.
// That is the register layout of v_mfma_f64_16x16x4, so the two contractions over the basis run on the matrix pipe while the
// vector pipe does what is left per (element, bucket) -- P, log P, 1 / P, the loss term, G:
//   forward   acc[b][e] = sum_r Ls[b][r] Psi_r(e):   A = Ls[b0 + (l & 15)][4 q + (l >> 4)], B = Psi_{4 q + k}(e_j) (RS / 4 registers
//             per lane, loaded once), D register m = acc of bucket b0 + k + 4 m at element j -- exactly the lane's four pairs;
//   backward  Gh_r(e) += sum_b Lg[b][r] G_b(e):      for m = 0 .. 3 the K-step {b0 + 4 m + k}: B = the lane's G of register m (no
//             shuffle), A = Lg[b0 + 4 m + (l >> 4)][16 rho + (l & 15)], D tile rho register q = Gh of row 16 rho + k + 4 q at element j.
// (As 48 v_fma_f64 per pair with the coefficients in SGPRs -- one element per thread, 96 accumulator VGPRs, the scalar loads of
// every bucket's 48 coefficients waited for in place -- the kernel took 0.20 ms.  On gfx950 the f64 MFMA runs on the f64 vector
// pipe: the MFMA form saves instruction issue, 80 registers and the scalar-load waits, not pipe time: EXPERIMENTS section 13.)
// Ls / Lg / t live in LDS (zero rows behind the last bucket: a step that hangs over B adds nothing), shared by the waves of
// a workgroup.  The short-branch buckets are a PREFIX of the (ascending) grid: steps behind it skip the forward product and
// read P_b of the long-branch buckets.  Loss partial per workgroup in a fixed order.
template <int RS, int RG>
struct TbEwLds {
  static constexpr int SS = RS + 1, SG = RG + 2;   // row strides (doubles)
  int BP, LS, LG, TT, LT, RED, TOTAL;              // bucket rows (B rounded up to whole steps) and the offsets (doubles)
  __host__ __device__ explicit TbEwLds(int B) {
    BP = (B + 15) & ~15;
    LS = 0;
    LG = LS + BP * SS;
    TT = LG + BP * SG;
    LT = TT + BP;
    RED = LT + 256;
    TOTAL = RED + 16;
  }
};

// One step of 16 buckets for the lane's four (element, bucket) pairs.  A step whose buckets are all short-branch ones skips the
// P_b loads, one whose buckets are all long-branch ones the forward product (wave-uniform branches); the step that holds the
// boundary does both and chooses per lane.
// The logarithm is the table form WITHOUT its argument checks (it is finite for every bit pattern) and P is clamped at 1e-300
// from below (v_max_f64: a NaN becomes 1e-300 as well): a pair with C = 0 then contributes 0 * finite = 0 to the loss and G = 0
// whatever P is (a tiny entry rounded to <= 0, the pad, a bucket past the end) without a select; a pair with C != 0 whose P is
// not a positive number sets `bad`, and the wave's loss partial becomes NaN.
template <int RS, int RG>
__device__ __forceinline__ void tb_ew_step(int b0, int j, int k, int nsm, const double *sLs, const double *sLg, const double *sT,
                                           const double *ltab, const double (&psi)[RS / 4], double a_e, double dl, double inv_n,
                                           const double (&c)[4], const double (&pd)[4], d4 (&D)[RG / 16], double &loss, int &bad) {
  typedef TbEwLds<RS, RG> Lds;
  const bool any_small = b0 < nsm, any_direct = b0 + 16 > nsm;   // (wave-uniform)
  double P[4];
  if (any_small) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#ifndef TB_NO_FWD
#pragma unroll
    for (int q = 0; q < RS / 4; ++q) acc = mfma_f64(sLs[(b0 + j) * Lds::SS + 4 * q + k], psi[q], acc);
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double t = sT[b0 + k + 4 * m];
      P[m] = fma(t, fma(t, acc[m], a_e), dl);   // I + t A + t^2 sum_r Ls Psi_r: the t^2 outside keeps the O(t^2) entries' digits
    }
    if (any_direct) {
#pragma unroll
      for (int m = 0; m < 4; ++m) P[m] = (b0 + k + 4 * m < nsm) ? P[m] : pd[m];
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m) P[m] = pd[m];
  }
  double g[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double cv = c[m];
    bad |= (!(P[m] >= 1e-300) && cv != 0.0) ? 1 : 0;
    const double Pu = fmax(P[m], 1e-300);
#ifdef TB_NO_LOG
    loss = fma(-cv, Pu, loss);
    g[m] = (-cv * inv_n) * Pu;
#else
    loss = fma(-cv, fast_log_table_unchecked(Pu, ltab), loss);
    g[m] = (-cv * inv_n) * fast_rcp(Pu);
#endif
  }
#ifdef TB_NO_BWD
#pragma unroll
  for (int m = 0; m < 4; ++m) D[0][m] += g[m];
#else
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double *w = sLg + (b0 + 4 * m + k) * Lds::SG + j;
#pragma unroll
    for (int q = 0; q < RG / 16; ++q) D[q] = mfma_f64(w[16 * q], g[m], D[q]);
  }
#endif
}

// SYM (the bank's counts are symmetric -- the only banks the time basis serves --, so is every P_b, G_b and Gh_r): only the
// 16 x 16 blocks (I, J >= I) of the upper block triangle are visited -- item = (block, row of the block), workgroup = eight rows
// of a block (two workgroups per CU) --, the loss terms of an off-diagonal block count twice, and tb_mirror copies the blocks of
// Gh_r across the diagonal afterwards: 52 % of the pairs, reads and products of the full matrix.
template <int RS, int RG, int NWAVE, bool SYM>
__global__ __launch_bounds__(NWAVE * 64, 4) void tb_ew(CbTbEwArgs a) {   // (four waves per SIMD: 128 registers)
  constexpr int NT = NWAVE * 64;
  typedef TbEwLds<RS, RG> Lds;
  extern __shared__ double smem[];
  if (a.skip && *a.skip != 0ull) return;
  const Lds o(a.B);
  double *sLs = smem + o.LS, *sLg = smem + o.LG, *sT = smem + o.TT, *ltab = smem + o.LT, *sred = smem + o.RED;
  const int tid = threadIdx.x;
  for (int i = tid; i < o.BP * RS; i += NT) {
    const int b = i / RS, r = i - b * RS;
    sLs[b * Lds::SS + r] = b < a.B ? a.Ls[(size_t)b * CB_TB_RS_MAX + r] : 0.0;
  }
  for (int i = tid; i < o.BP * RG; i += NT) {
    const int b = i / RG, r = i - b * RG;
    sLg[b * Lds::SG + r] = (b < a.B && r < CB_TB_RG_MAX) ? a.Lg[(size_t)b * CB_TB_RG_MAX + r] : 0.0;
  }
  for (int i = tid; i < o.BP; i += NT) sT[i] = i < a.B ? a.t[i] : 0.0;
  fast_log_table_fill(ltab, tid, NT);
  __syncthreads();
  const size_t LL = (size_t)a.LD * a.LD;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 15, k = lane >> 4;
  size_t e;
  bool live;
  double lossw = 1.0;
  if (SYM) {
    // item -> (block (I, J >= I) of the upper block triangle, row of the block): row I of the triangle holds nb - I blocks
    const int nb = a.LD / 16, item = (int)blockIdx.x * NWAVE + wave, blk = item >> 4, rr = item & 15;
    int I = 0, rest = blk;
    while (I < nb && rest >= nb - I) {
      rest -= nb - I;
      ++I;
    }
    live = I < nb;
    const int J = I + rest;
    e = live ? ((size_t)(16 * I + rr) * a.LD + 16 * J + j) : 0;
    lossw = J > I ? 2.0 : 1.0;
  } else {
    e = ((size_t)blockIdx.x * NWAVE + wave) * 16 + j;     // (LD % 16 == 0: the 16 elements share a row)
    live = e < LL;                                       // (a whole wave: LL % 16 == 0)
  }
  const size_t ec = live ? e : 0;
  const int row = (int)(ec / a.LD), col = (int)(ec - (size_t)row * a.LD);
  const double *__restrict__ Psi = a.Psi + ec;
  double psi[RS / 4];
#pragma unroll
  for (int q = 0; q < RS / 4; ++q) psi[q] = (4 * q + k < a.ns) ? Psi[(size_t)(4 * q + k) * LL] : 0.0;
  const double a_e = a.A[ec], dl = row == col ? 1.0 : 0.0;
  d4 D[RG / 16];
#pragma unroll
  for (int q = 0; q < RG / 16; ++q) D[q] = d4{0.0, 0.0, 0.0, 0.0};
  double loss = 0.0;
  int bad = 0;
  const int nsm = a.nsmall, B = a.B;
  // counts and long-branch P_b through BUFFER loads: the lane's byte offsets inside a step are fixed (element + bucket k + 4 m of
  // the step) and the step is a scalar offset -- no address arithmetic in the loop.  Only the step that hangs over the end of
  // the bank (and, for P_b, the one that holds the first long-branch bucket) checks its buckets, behind a wave-uniform branch.
  // (LD <= 1024 and (B + 32) 8 LD^2 < 2^31 keep the offsets inside 31 bits: cb_tb_launch_ew checks)
  const unsigned plane = (unsigned)(LL * sizeof(double));
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.Ct), 0, 0x7fffffff, 0x00027000);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.Psi + (size_t)a.ns * LL), 0, 0x7fffffff, 0x00027000);
  const int eoff = (int)(ec * sizeof(double));
  int voff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) voff[m] = (int)((unsigned)(k + 4 * m) * plane) + eoff;
  auto load = [&](int b0, double (&cc)[4], double (&pp)[4]) {
    if (b0 >= B) return;
#ifdef TB_NO_LOAD
    for (int m = 0; m < 4; ++m) cc[m] = 1.0 + b0 + m, pp[m] = 0.5;
    return;
#endif
    if (b0 + 16 <= B) {   // (wave-uniform)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        cc[m] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, voff[m], (int)((unsigned)b0 * plane), 0));
      if (b0 >= nsm) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          pp[m] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rp, voff[m], (int)((unsigned)(b0 - nsm) * plane), 0));
        return;
      }
      if (b0 + 16 <= nsm) return;   // short-branch buckets only: no P_b
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool in = b0 + k + 4 * m < B;
        const double v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, in ? voff[m] + (int)((unsigned)b0 * plane) : eoff, 0, 0));
        cc[m] = in ? v : 0.0;
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {   // the boundary step, or the last one: every bucket checked
      const int b = b0 + k + 4 * m;
      const bool in = b >= nsm && b < B;
      const double v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rp, (int)((unsigned)(in ? b - nsm : 0) * plane) + eoff, 0, 0));
      pp[m] = in ? v : 1.0;
    }
  };
  // two steps per iteration: the loads of the next step in flight during a step's arithmetic, ping-pong registers
  double c0[4] = {0.0, 0.0, 0.0, 0.0}, c1[4] = {0.0, 0.0, 0.0, 0.0}, p0[4] = {1.0, 1.0, 1.0, 1.0}, p1[4] = {1.0, 1.0, 1.0, 1.0};
  load(0, c0, p0);
  for (int b0 = 0; b0 < B; b0 += 32) {
    load(b0 + 16, c1, p1);
    tb_ew_step<RS, RG>(b0, j, k, nsm, sLs, sLg, sT, ltab, psi, a_e, dl, a.inv_n, c0, p0, D, loss, bad);
    if (b0 + 16 >= B) break;
    load(b0 + 32, c0, p0);
    tb_ew_step<RS, RG>(b0 + 16, j, k, nsm, sLs, sLg, sT, ltab, psi, a_e, dl, a.inv_n, c1, p1, D, loss, bad);
  }
  if (live) {
    if (a.Gh32) {   // (CB_MIXED: Gh_r rounded to float32 once, here; the two gradient products run on the float32 MFMA)
      float *__restrict__ Gh = a.Gh32 + e;
#pragma unroll
      for (int q = 0; q < RG / 16; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * q + k + 4 * r;
          if (rr < a.ng) Gh[(size_t)rr * LL] = (float)D[q][r];
        }
    } else {
      double *__restrict__ Gh = a.Gh + e;
#pragma unroll
      for (int q = 0; q < RG / 16; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * q + k + 4 * r;
          if (rr < a.ng) Gh[(size_t)rr * LL] = D[q][r];
        }
    }
  } else {
    loss = 0.0;
    bad = 0;
  }
  if (bad) loss = NAN;
  loss = wave_sum(loss) * lossw;
  if (lane == 0) sred[wave] = loss;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < NWAVE; ++w) s += sred[w];
    a.loss_part[blockIdx.x] = s;
  }
}

// Gh_r(16 J + c, 16 I + r) = Gh_r(16 I + r, 16 J + c) for the blocks I < J: one workgroup per (block, virtual bucket), the block
// through LDS so that both sides move in 128-byte rows.
template <typename T>
__global__ __launch_bounds__(256) void tb_mirror(int LD, T *__restrict__ Gh, const unsigned long long *skip) {
  __shared__ T tile[16][17];
  if (skip && *skip != 0ull) return;
  const int nb = LD / 16;
  int I = 0, rest = (int)blockIdx.x;   // off-diagonal blocks only: row I of the strict triangle holds nb - 1 - I
  while (rest >= nb - 1 - I) {
    rest -= nb - 1 - I;
    ++I;
  }
  const int J = I + 1 + rest, r = threadIdx.x >> 4, c = threadIdx.x & 15;
  T *G = Gh + (size_t)blockIdx.y * LD * LD;
  tile[r][c] = G[(size_t)(16 * I + r) * LD + 16 * J + c];
  __syncthreads();
  G[(size_t)(16 * J + r) * LD + 16 * I + c] = tile[c][r];
}

int cb_tb_launch_tables(int LD, int ns, int nd, int ng, const double *tf, const double *tg, const double *lam, double *F, double *E,
                        double *H, const unsigned long long *skip, hipStream_t stream) {
  const int n = (ns + nd + ng) * LD;
  hipLaunchKernelGGL(tb_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, LD, ns, nd, ng, tf, tg, lam, F, E, H, skip);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// what tb_ew can serve: the interpolation matrices of B buckets in its LDS, the buffer loads' 31-bit offsets
bool cb_tb_supported(int B, int LD, int ns, int ng) {
  return ns <= CB_TB_RS_MAX && ng <= CB_TB_RG_MAX && LD % 16 == 0 && cb_tb_ew_lds_bytes(B, ns, ng) <= CB_TB_LDS_MAX &&
         (double)(B + 32) * (double)LD * (double)LD * 8.0 < 2147483648.0;
}

size_t cb_tb_ew_lds_bytes(int B, int ns, int ng) {
  const int RS = ns <= 16 ? 16 : 24, RG = ng <= 32 ? 32 : 48;
  const int BP = (B + 15) & ~15;
  return ((size_t)BP * (RS + 1 + RG + 2 + 1) + 256 + 16) * sizeof(double);
}

// The dynamic-LDS limit of the tb_ew instantiation that serves (ns, ng), raised to what B buckets need.  The attribute belongs to
// the DEVICE the caller has made current: `cache` (four words, one per instantiation) lives in the handle, so a second handle
// on another GPU of the same process sets it there too (round 5 kept one word per process: ADVICE r5).
int cb_tb_prepare_ew(int B, int ns, int ng, size_t *cache) {
#define TB_PREP(RS_, RG_, slot_)                                                                                         \
  do {                                                                                                                   \
    const size_t lds = (size_t)TbEwLds<RS_, RG_>(B).TOTAL * sizeof(double);                                              \
    if (lds > cache[slot_]) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(tb_ew<RS_, RG_, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                                         \
        return -1;                                                                                                       \
      }                                                                                                                  \
      cache[slot_] = lds;                                                                                                \
    }                                                                                                                    \
  } while (0)
  if (ns <= 16 && ng <= 32) TB_PREP(16, 32, 0);
  else if (ns <= 16) TB_PREP(16, 48, 1);
  else if (ng <= 32) TB_PREP(24, 32, 2);
  else TB_PREP(24, 48, 3);
#undef TB_PREP
  return 0;
}

int cb_tb_launch_ew(const CbTbEwArgs &a, hipStream_t stream, hipEvent_t stop, int *nparts) {
  if (!cb_tb_supported(a.B, a.LD, a.ns, a.ng)) return -1;
  // the symmetric form: eight-wave workgroups (two per CU), items of the upper block triangle, then the mirror copies
  const int nb = a.LD / 16, nblk = nb * (nb + 1) / 2;
  const unsigned grid = (unsigned)(nblk * 2);   // 16 rows per block, 8 per workgroup
  if (nparts) *nparts = (int)grid;
  // (the dynamic-LDS attribute was set for this device when the basis was installed: cb_tb_prepare_ew)
#define TB_GO(RS_, RG_)                                                                                                  \
  do {                                                                                                                   \
    const size_t lds = (size_t)TbEwLds<RS_, RG_>(a.B).TOTAL * sizeof(double);                                            \
    hipLaunchKernelGGL((tb_ew<RS_, RG_, 8, true>), dim3(grid), dim3(512), lds, stream, a);                               \
  } while (0)
  if (a.ns <= 16 && a.ng <= 32) TB_GO(16, 32);
  else if (a.ns <= 16) TB_GO(16, 48);
  else if (a.ng <= 32) TB_GO(24, 32);
  else TB_GO(24, 48);
#undef TB_GO
  // (the mirrored entries written straight from tb_ew -- eight-byte stores LD apart, no second launch -- measured: 0.069 against 0.066 ms)
  if (nb > 1) {
    const dim3 mg((unsigned)(nb * (nb - 1) / 2), (unsigned)a.ng);
    if (a.Gh32) {
      if (stop) hipExtLaunchKernelGGL(tb_mirror<float>, mg, dim3(256), 0, stream, nullptr, stop, 0, a.LD, a.Gh32, a.skip);
      else hipLaunchKernelGGL(tb_mirror<float>, mg, dim3(256), 0, stream, a.LD, a.Gh32, a.skip);
    } else {
      if (stop) hipExtLaunchKernelGGL(tb_mirror<double>, mg, dim3(256), 0, stream, nullptr, stop, 0, a.LD, a.Gh, a.skip);
      else hipLaunchKernelGGL(tb_mirror<double>, mg, dim3(256), 0, stream, a.LD, a.Gh, a.skip);
    }
  } else if (stop) {
    (void)hipEventRecord(stop, stream);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
