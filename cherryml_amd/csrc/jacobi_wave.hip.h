// One-wavefront symmetric eigensolver for n <= 32, matrices in LDS.
//
// One-sided (Hestenes) Jacobi on the SHIFTED matrix A' = A - sigma I with
// sigma >= max |A_ii| (> 0): a symmetrised rate matrix is negative
// semidefinite with a zero eigenvalue, so A' is negative definite with
// condition number <= ~3, every column of G = A' V keeps a healthy norm and
// the sweep count is small and predictable.  V accumulates the rotations
// (orthogonal by construction), on exit
//     A = V diag(lam) V^T,   lam_k = v_k . g_k + sigma.
//
// Layout: Gc[k*LS + r] / Vc[k*LS + r] hold COLUMN k (component r).
// Work split: the n/2 disjoint pairs of a round (round-robin tournament) are
// handled by 4 lanes each (lanes 4p .. 4p+3 take rows sub, sub+4, ...).
#pragma once
#include "common.hip.h"

// A rotation is skipped when the pair is already orthogonal to rounding; the
// sweeps stop after the first sweep whose largest PRE-rotation cosine was
// below CB_JAC_STOP (quadratic convergence: that sweep leaves ~CB_JAC_STOP^2).
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

// Returns the number of sweeps used (CB_JAC_MAX_SWEEPS + 1 if not converged).
// A (row-major, stride LS) is read only; sigma_shift < 0 means "choose".
__device__ int wave_jacobi(int n, const double *A, double *Gc, double *Vc, double *lam,
                           int LS, double sigma_shift) {
  const int lane = threadIdx.x & 63;
  // shift
  double sigma = sigma_shift;
  if (sigma_shift < 0.0) {
    double m = 0.0;
    for (int i = lane; i < n; i += 64) m = fmax(m, fabs(A[i * LS + i]));
    sigma = wave_max(m);
    if (!(sigma > 0.0)) sigma = 1.0;
  }
  for (int e = lane; e < n * n; e += 64) {
    const int k = e / n, r = e - k * n;
    Gc[k * LS + r] = A[r * LS + k] - (r == k ? sigma : 0.0);
    Vc[k * LS + r] = (r == k) ? 1.0 : 0.0;
  }
  wave_lds_fence();

  const int m = (n + 1) & ~1;  // even number of players (one dummy when n is odd)
  const int slot = lane >> 2, sub = lane & 3;
  int sweeps = 0;
  for (; sweeps < CB_JAC_MAX_SWEEPS; ++sweeps) {
    double off = 0.0;
    for (int r = 0; r < m - 1; ++r) {
      int p = 0, q = 0;
      const bool active = slot < (m >> 1);
      if (active) rr_pair(m, r, slot, p, q);
      const bool real = active && p < n && q < n;
      double a = 0.0, b = 0.0, g = 0.0;
      if (real) {
        for (int row = sub; row < n; row += 4) {
          const double x = Gc[p * LS + row], y = Gc[q * LS + row];
          a = fma(x, x, a);
          b = fma(y, y, b);
          g = fma(x, y, g);
        }
      }
      a += __shfl_xor(a, 1, 64);
      b += __shfl_xor(b, 1, 64);
      g += __shfl_xor(g, 1, 64);
      a += __shfl_xor(a, 2, 64);
      b += __shfl_xor(b, 2, 64);
      g += __shfl_xor(g, 2, 64);
      if (real) {
        const double denom = sqrt(a * b);
        const double rel = (denom > 0.0) ? fabs(g) / denom : 0.0;
        off = fmax(off, rel);
        if (rel > CB_JAC_SKIP) {
          const double zeta = (b - a) / (2.0 * g);
          const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
          const double c = 1.0 / sqrt(fma(tt, tt, 1.0));
          const double s = c * tt;
          for (int row = sub; row < n; row += 4) {
            const double x = Gc[p * LS + row], y = Gc[q * LS + row];
            Gc[p * LS + row] = c * x - s * y;
            Gc[q * LS + row] = s * x + c * y;
            const double u = Vc[p * LS + row], v = Vc[q * LS + row];
            Vc[p * LS + row] = c * u - s * v;
            Vc[q * LS + row] = s * u + c * v;
          }
        }
      }
      wave_lds_fence();
    }
    off = wave_max(off);
    if (off <= CB_JAC_STOP) {
      ++sweeps;
      break;
    }
  }
  for (int k = lane; k < n; k += 64) {
    double d = 0.0;
    for (int r = 0; r < n; ++r) d = fma(Vc[k * LS + r], Gc[k * LS + r], d);
    lam[k] = d + sigma;
  }
  wave_lds_fence();
  return sweeps;
}
