// FastCherries branch-length / site-rate estimation on the GPU: per-cherry and
// per-site log-likelihood gathers from the bank logP[t][r][x][y] = log expm(grid_t rate_r Q)
// (reference: cherryml/phylogeny_estimation/FastCherries/branch_length_estimation.cpp:60-144) and
// the SiteRM site-rate gather (cherryml/_siterm/fast_site_rates.pyx:8-47).
// The bank (T x R x S x S doubles, 8 MB for T = 129, R = 20, S = 20) is L2 resident; the kernels
// are gather + wavefront reductions, one wavefront per cherry (lanes over sites) or per site
// (lanes over cherries).  The bisections are wave-uniform: every lane sees the same sums.
#pragma once
#include "common.hip.h"

// out[c] = argmax_t sum_sites logP[t][rate(site)][x][y] + logP[t][rate(site)][y][x], found by the
// reference's bisection (compare mid against mid + 1; `>` keeps the lower index).
// changed[0] is set when out[c] differs from prev[c] (prev may be null).
__global__ __launch_bounds__(256) void ble_branch_lengths_kernel(
    int S, int T, int R, int n, int L, const double *__restrict__ logP, const int8_t *__restrict__ cx,
    const int8_t *__restrict__ cy, const int *__restrict__ site_to_rate, const int *__restrict__ prev,
    int *__restrict__ out, int *__restrict__ changed) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= n) return;
  const int8_t *x = cx + (size_t)c * L, *y = cy + (size_t)c * L;
  const size_t SS = (size_t)S * S, RSS = (size_t)R * SS;
  int low = 0, high = T - 1;
  while (low < high) {
    const int mid = low + (high - low) / 2;
    const double *Pm = logP + (size_t)mid * RSS;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < L; i += 64) {
      const int xi = x[i], yi = y[i];
      if (xi < 0 || yi < 0) continue;
      const double *M = Pm + (size_t)site_to_rate[i] * SS;
      a += M[xi * S + yi] + M[yi * S + xi];
      b += M[RSS + xi * S + yi] + M[RSS + yi * S + xi];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (a > b) high = mid;
    else low = mid + 1;
  }
  if (lane == 0) {
    out[c] = low;
    if (prev && prev[c] != low) atomicOr(changed, 1);
  }
}

// out[s] = rate category maximising prior[r] + sum_cherries (logP[len(c)][r][x][y] + logP[..][y][x]).
// cxT / cyT are the SITE-major copies [L][n] of the cherries, so that the lanes (consecutive
// cherries of one site) read consecutive bytes.
__global__ __launch_bounds__(256) void ble_site_rates_kernel(
    int S, int /*T*/, int R, int n, int L, const double *__restrict__ logP, const int8_t *__restrict__ cxT,
    const int8_t *__restrict__ cyT, const int *__restrict__ lengths_index, const double *__restrict__ priors,
    int *__restrict__ out) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= L) return;
  const size_t SS = (size_t)S * S, RSS = (size_t)R * SS;
  int low = 0, high = R - 1;
  while (low < high) {
    const int mid = low + (high - low) / 2;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < n; i += 64) {
      const int xi = cxT[(size_t)s * n + i], yi = cyT[(size_t)s * n + i];
      if (xi < 0 || yi < 0) continue;
      const double *M = logP + (size_t)lengths_index[i] * RSS + (size_t)mid * SS;
      a += M[xi * S + yi] + M[yi * S + xi];
      b += M[SS + xi * S + yi] + M[SS + yi * S + xi];
    }
    a = wave_sum(a) + priors[mid];
    b = wave_sum(b) + priors[mid + 1];
    if (a > b) high = mid;
    else low = mid + 1;
  }
  if (lane == 0) out[s] = low;
}

// fast_site_rates.pyx: best[s] = first r maximising log_prior[r] + sum_c tens[r][c][x_cs][y_cs]
__global__ __launch_bounds__(256) void site_rate_gather_kernel(int S, int R, int n, int L,
                                                               const double *__restrict__ tens,
                                                               const int8_t *__restrict__ cx,
                                                               const int8_t *__restrict__ cy,
                                                               const double *__restrict__ log_prior,
                                                               int *__restrict__ best) {
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= L) return;
  const size_t SS = (size_t)S * S;
  int arg = 0;
  double top = -INFINITY;
  for (int r = 0; r < R; ++r) {
    double a = 0.0;
    for (int c = lane; c < n; c += 64) {
      const int xi = cx[(size_t)c * L + s], yi = cy[(size_t)c * L + s];
      a += tens[((size_t)r * n + c) * SS + xi * S + yi];
    }
    a = wave_sum(a) + log_prior[r];
    if (a > top || r == 0) {
      top = a;
      arg = r;
    }
  }
  if (lane == 0) best[s] = arg;
}
