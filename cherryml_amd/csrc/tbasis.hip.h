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
// of dL/dA against the per-bucket form 3e-13 .. 4e-12 (both are 7e-11 from a long-double evaluation: the eigendecomposition's
// own backward error dominates; /tmp prototype recorded in EXPERIMENTS section 13).
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
      ld s = 0.0L;
      const ld *w = &W[(size_t)b * N];
      for (int i = 0; i < N; ++i) s += w[i] * w[i];
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
        ld d = 0.0L;
        for (int i = 0; i < N; ++i) d += qk[i] * q[i];
        for (int i = 0; i < N; ++i) q[i] -= d * qk[i];
      }
      ld s = 0.0L;
      for (int i = 0; i < N; ++i) s += q[i] * q[i];
      const ld in = 1.0L / sqrtl(s);
      for (int i = 0; i < N; ++i) q[i] *= in;
    }
    for (int i = 0; i < N; ++i) Q[(size_t)r * N + i] = q[i];
    for (int b = 0; b < B; ++b) {
      ld *w = &W[(size_t)b * N];
      ld c = 0.0L;
      for (int i = 0; i < N; ++i) c += w[i] * q[i];
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
    const int R = id_rows(B, N, F, std::min(CB_TB_RG_MAX, B), 1e-16L, sk, L, res);
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

// One element (row, col) of ALL buckets per thread.  RS / RG: compile-time bounds of ns / ng (the arrays live in registers).
// Loss partial per workgroup (fixed order inside: thread sums over b, wave sum, four waves), summed by lg_finish_loss_body.
template <int RS, int RG>
__global__ __launch_bounds__(256) void tb_ew(CbTbEwArgs a) {
  __shared__ double ltab[256];
  __shared__ double sred[4];
  if (a.skip && *a.skip != 0ull) return;
  fast_log_table_fill(ltab, threadIdx.x, 256);
  __syncthreads();
  const size_t LL = (size_t)a.LD * a.LD, e = (size_t)blockIdx.x * 256 + threadIdx.x;   // (LD % 16 == 0: LL % 256 == 0)
  const int row = (int)(e / a.LD), col = (int)(e - (size_t)row * a.LD);
  const double *__restrict__ Ct = a.Ct + e;
  const double *__restrict__ Psi = a.Psi + e;
  const double *__restrict__ Ls = a.Ls;
  const double *__restrict__ Lg = a.Lg;
  const double *__restrict__ tb = a.t;
  const int *__restrict__ kind = a.kind;
  double psi[RS], G[RG];
#pragma unroll
  for (int r = 0; r < RS; ++r) psi[r] = r < a.ns ? Psi[(size_t)r * LL] : 0.0;
#pragma unroll
  for (int r = 0; r < RG; ++r) G[r] = 0.0;
  const double a_e = a.A[e], dl = row == col ? 1.0 : 0.0;
  const double *__restrict__ Pd = Psi + (size_t)a.ns * LL;   // the direct buckets' P_b
  double loss = 0.0;
  constexpr int U = 4;   // buckets per group; the next group's counts (and direct P_b) are loaded before this group's arithmetic
  double c[U], pd[U], cn[U], pn[U];
  auto load = [&](int b0, double (&cc)[U], double (&pp)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = min(b0 + u, a.B - 1);
      cc[u] = Ct[(size_t)b * LL];
      const int k = kind[b];   // (wave-uniform)
      pp[u] = k >= 0 ? Pd[(size_t)k * LL] : 0.0;
    }
  };
  load(0, c, pd);
  for (int b0 = 0; b0 < a.B; b0 += U) {
    if (b0 + U < a.B) load(b0 + U, cn, pn);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u;
      if (b < a.B) {   // (wave-uniform)
        double P;
        if (kind[b] < 0) {
          const double *__restrict__ l = Ls + (size_t)b * CB_TB_RS_MAX;
          double s0 = 0.0, s1 = 0.0;   // two chains
#pragma unroll
          for (int r = 0; r + 1 < RS; r += 2) {
            s0 = fma(l[r], psi[r], s0);
            s1 = fma(l[r + 1], psi[r + 1], s1);
          }
          if (RS & 1) s0 = fma(l[RS - 1], psi[RS - 1], s0);
          const double t = tb[b];
          P = fma(t, fma(t, s0 + s1, a_e), dl);   // I + t A + t^2 sum_r Ls Psi_r: the t^2 outside keeps the O(t^2) entries' digits
        } else {
          P = pd[u];
        }
        const double cv = c[u];
        const bool nz = cv != 0.0;   // (P <= 0 only where C = 0: rounding of a tiny entry, or the pad)
        const double lg = fast_log_table(P, ltab), rc = fast_rcp(P);
        loss = fma(-cv, nz ? lg : 0.0, loss);
        const double g = nz ? -cv * a.inv_n * rc : 0.0;
        const double *__restrict__ w = Lg + (size_t)b * CB_TB_RG_MAX;
#pragma unroll
        for (int r = 0; r < RG; ++r) G[r] = fma(w[r], g, G[r]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c[u] = cn[u];
      pd[u] = pn[u];
    }
  }
  double *__restrict__ Gh = a.Gh + e;
#pragma unroll
  for (int r = 0; r < RG; ++r)
    if (r < a.ng) Gh[(size_t)r * LL] = G[r];
  loss = wave_sum(loss);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) a.loss_part[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

int cb_tb_launch_tables(int LD, int ns, int nd, int ng, const double *tf, const double *tg, const double *lam, double *F, double *E,
                        double *H, const unsigned long long *skip, hipStream_t stream) {
  const int n = (ns + nd + ng) * LD;
  hipLaunchKernelGGL(tb_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, LD, ns, nd, ng, tf, tg, lam, F, E, H, skip);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int cb_tb_launch_ew(const CbTbEwArgs &a, hipStream_t stream, hipEvent_t stop) {
  const unsigned grid = (unsigned)(((size_t)a.LD * a.LD) / 256);
  if (a.ns > CB_TB_RS_MAX || a.ng > CB_TB_RG_MAX) return -1;
#define TB_GO(RS_, RG_)                                                                                        \
  do {                                                                                                         \
    if (stop) hipExtLaunchKernelGGL((tb_ew<RS_, RG_>), dim3(grid), dim3(256), 0, stream, nullptr, stop, 0, a); \
    else hipLaunchKernelGGL((tb_ew<RS_, RG_>), dim3(grid), dim3(256), 0, stream, a);                           \
  } while (0)
#define TB_RG(RS_)                  \
  do {                              \
    if (a.ng <= 24) TB_GO(RS_, 24); \
    else if (a.ng <= 28) TB_GO(RS_, 28); \
    else if (a.ng <= 32) TB_GO(RS_, 32); \
    else if (a.ng <= 36) TB_GO(RS_, 36); \
    else TB_GO(RS_, 40);            \
  } while (0)
  if (a.ns <= 16) TB_RG(16);
  else TB_RG(24);
#undef TB_RG
#undef TB_GO
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
