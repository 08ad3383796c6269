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
  // The lane's first BLE_HOIST sites do not change over the ~7 steps of the bisection: their two bank offsets are formed ONCE
  // (the state bytes and the site's rate were three dependent loads in front of every gather of every step); a step is then
  // 4 BLE_HOIST independent gathers per lane, all in flight together.  Same sums in the same order as the plain loop.
  constexpr int BLE_HOIST = 8;   // x 64 lanes = 512 sites in registers; longer alignments take the plain loop for the rest
  int oxy[BLE_HOIST], oyx[BLE_HOIST];
#pragma unroll
  for (int k = 0; k < BLE_HOIST; ++k) {
    const int i = lane + 64 * k;
    oxy[k] = -1;
    oyx[k] = -1;
    if (i < L) {
      const int xi = x[i], yi = y[i];
      if (xi >= 0 && yi >= 0) {
        const int base = site_to_rate[i] * (int)SS;
        oxy[k] = base + xi * S + yi;
        oyx[k] = base + yi * S + xi;
      }
    }
  }
  int low = 0, high = T - 1;
  while (low < high) {
    const int mid = low + (high - low) / 2;
    const double *Pm = logP + (size_t)mid * RSS;
    double a = 0.0, b = 0.0;
    double va[BLE_HOIST][2], vb[BLE_HOIST][2];
#pragma unroll
    for (int k = 0; k < BLE_HOIST; ++k) {
      const int o1 = oxy[k] < 0 ? 0 : oxy[k], o2 = oyx[k] < 0 ? 0 : oyx[k];   // (an unobserved site reads entry 0 and adds nothing)
      va[k][0] = Pm[o1];
      va[k][1] = Pm[o2];
      vb[k][0] = Pm[RSS + o1];
      vb[k][1] = Pm[RSS + o2];
    }
#pragma unroll
    for (int k = 0; k < BLE_HOIST; ++k)
      if (oxy[k] >= 0) {
        a += va[k][0] + va[k][1];
        b += vb[k][0] + vb[k][1];
      }
    for (int i = lane + 64 * BLE_HOIST; i < L; i += 64) {
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
// NW = waves per site: 1 -- four sites per workgroup, one wave each (many sites); 4 -- ONE site per workgroup, its cherries
// in four contiguous ranges, partial sums met in LDS in a fixed order (a family of a few hundred sites is a few hundred
// waves: one per CU and SIMD at best, each walking 32 cherries per lane through five dependent bisection steps).
template <int NW>
__global__ __launch_bounds__(256) void ble_site_rates_kernel(
    int S, int /*T*/, int R, int n, int L, const double *__restrict__ logP, const int8_t *__restrict__ cxT,
    const int8_t *__restrict__ cyT, const int *__restrict__ lengths_index, const double *__restrict__ priors,
    int *__restrict__ out) {
  __shared__ double part[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = NW == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (s >= L) return;   // (NW == 1: a wave of its own; NW == 4: the whole workgroup)
  const size_t SS = (size_t)S * S, RSS = (size_t)R * SS;
  // this wave's cherries: all of them (NW == 1) or the wave's quarter, in whole groups of 64
  const int per = NW == 1 ? n : (((n + NW - 1) / NW + 63) / 64) * 64;
  const int i_lo = NW == 1 ? 0 : min(n, wave * per), i_hi = NW == 1 ? n : min(n, (wave + 1) * per);
  // (as in ble_branch_lengths_kernel: every load of a group of BLE_SR_HOIST x 64 cherries in flight together; same sums in
  // the same order as the plain loop)
  constexpr int BLE_SR_HOIST = 8;
  int low = 0, high = R - 1;
  const int nfull = i_lo + (i_hi - i_lo) / (64 * BLE_SR_HOIST) * (64 * BLE_SR_HOIST);
  while (low < high) {
    const int mid = low + (high - low) / 2;
    double a = 0.0, b = 0.0;
    for (int c0 = i_lo; c0 < nfull; c0 += 64 * BLE_SR_HOIST) {
      long long o1[BLE_SR_HOIST], o2[BLE_SR_HOIST];
      bool ok[BLE_SR_HOIST];
#pragma unroll
      for (int k = 0; k < BLE_SR_HOIST; ++k) {
        const int i = c0 + lane + 64 * k;
        const int xi = cxT[(size_t)s * n + i], yi = cyT[(size_t)s * n + i];
        ok[k] = xi >= 0 && yi >= 0;
        const long long base = (long long)lengths_index[i] * (long long)RSS + (long long)mid * (long long)SS;
        o1[k] = base + (ok[k] ? xi * S + yi : 0);
        o2[k] = base + (ok[k] ? yi * S + xi : 0);
      }
      double va[BLE_SR_HOIST][2], vb[BLE_SR_HOIST][2];
#pragma unroll
      for (int k = 0; k < BLE_SR_HOIST; ++k) {
        va[k][0] = logP[o1[k]];
        va[k][1] = logP[o2[k]];
        vb[k][0] = logP[SS + o1[k]];
        vb[k][1] = logP[SS + o2[k]];
      }
#pragma unroll
      for (int k = 0; k < BLE_SR_HOIST; ++k)
        if (ok[k]) {
          a += va[k][0] + va[k][1];
          b += vb[k][0] + vb[k][1];
        }
    }
    for (int i = nfull + lane; i < i_hi; i += 64) {
      const int xi = cxT[(size_t)s * n + i], yi = cyT[(size_t)s * n + i];
      if (xi < 0 || yi < 0) continue;
      const double *M = logP + (size_t)lengths_index[i] * RSS + (size_t)mid * SS;
      a += M[xi * S + yi] + M[yi * S + xi];
      b += M[SS + xi * S + yi] + M[SS + yi * S + xi];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (NW > 1) {   // the four quarters, summed in wave order by everybody: every wave takes the same decision
      __syncthreads();   // (the previous step's reads of `part` are done)
      if (lane == 0) {
        part[0][wave] = a;
        part[1][wave] = b;
      }
      __syncthreads();
      a = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
      b = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    }
    a += priors[mid];
    b += priors[mid + 1];
    if (a > b) high = mid;
    else low = mid + 1;
  }
  if (lane == 0 && (NW == 1 || wave == 0)) out[s] = low;
}
// the form by the number of sites: a wave per site fills the chip from ~2000 sites on
static inline void ble_launch_site_rates(int S, int T, int R, int n, int L, const double *logP, const int8_t *cxT, const int8_t *cyT,
                                         const int *lengths_index, const double *priors, int *out, hipStream_t stream = 0) {
  if (L < 2048)
    hipLaunchKernelGGL(ble_site_rates_kernel<4>, dim3(L), dim3(256), 0, stream, S, T, R, n, L, logP, cxT, cyT, lengths_index, priors, out);
  else
    hipLaunchKernelGGL(ble_site_rates_kernel<1>, dim3((L + 3) / 4), dim3(256), 0, stream, S, T, R, n, L, logP, cxT, cyT, lengths_index, priors, out);
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


// ---- what the host did per call before round 6, as kernels (cb_ble_bank_run: the device-resident FastCherries entry) -----------
// [rows][cols] int8 -> [cols][rows] through a 64 x 65 LDS tile; also raises *bad when a code is >= S (the range check of
// ble_check, which read every byte on the host)
__global__ __launch_bounds__(256) void ble_transpose_check_kernel(int rows, int cols, int S, const int8_t *__restrict__ in,
                                                                  int8_t *__restrict__ out, int *bad) {
  __shared__ int8_t tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  int flag = 0;
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, cc = c0 + tx;
    int8_t v = -1;
    if (rr < rows && cc < cols) v = in[(size_t)rr * cols + cc];
    flag |= v >= S ? 1 : 0;
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int cc = c0 + c, rr = r0 + tx;
    if (cc < cols && rr < rows) out[(size_t)cc * rows + rr] = tile[tx][c];
  }
  if (flag) atomicOr(bad, 1);
}

// initial site-rate bins, first half (branch_length_estimation.cpp:10-34): per site the number of sequence PAIRS that differ,
// total[j] = sum_k (non_missing_j - cnt_jk) cnt_jk with cnt_jk = sequences holding state k at site j.  One workgroup per 32
// sites: 32 x (S + 1) LDS counters, eight sequence rows per step (32 consecutive bytes of a row per wave quarter).
__global__ __launch_bounds__(256) void ble_site_totals_kernel(int n_seqs, int L, int S, const int8_t *__restrict__ seqs,
                                                              long long *__restrict__ total, int *bad) {
  extern __shared__ int cnt[];   // [32][S]
  const int j0 = blockIdx.x * 32, sx = threadIdx.x & 31, sy = threadIdx.x >> 5;
  for (int e = threadIdx.x; e < 32 * S; e += 256) cnt[e] = 0;
  __syncthreads();
  int flag = 0;
  if (j0 + sx < L)
    for (int i = sy; i < n_seqs; i += 8) {
      const int v = seqs[(size_t)i * L + j0 + sx];
      if (v >= S) flag = 1;
      else if (v >= 0) atomicAdd(&cnt[sx * S + v], 1);
    }
  if (flag) atomicOr(bad, 1);
  __syncthreads();
  if (threadIdx.x < 32 && j0 + (int)threadIdx.x < L) {
    long long nm = 0, t = 0;
    for (int k = 0; k < S; ++k) nm += cnt[threadIdx.x * S + k];
    for (int k = 0; k < S; ++k) t += (nm - cnt[threadIdx.x * S + k]) * (long long)cnt[threadIdx.x * S + k];
    total[j0 + threadIdx.x] = t;
  }
}
