// libcherrybank: FastCherries branch lengths / site rates.
#include "cb_internal.hip.h"
#include "common.hip.h"
#include "ble.hip.h"

// ----------------------------------------------------------------- FastCherries BLE (8f #3)
extern "C" int cb_ble_log_bank(int device, int S, int T, int R, const double *Q, const double *pi,
                               const double *grid, const double *rates, double *logP) {
  if (!Q || !grid || !rates || !logP) return fail(CB_EINVAL, "cb_ble_log_bank: NULL argument");
  if (S < 2 || T < 1 || R < 1) return fail(CB_EINVAL, "cb_ble_log_bank: bad sizes");
  const size_t nb = (size_t)T * R, SS = (size_t)S * S;
  std::vector<double> tt(nb);
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < R; ++r) tt[(size_t)t * R + r] = grid[t] * rates[r];   // as io_helpers.cpp:161
  cb_handle h = nullptr;
  // a counts-free handle: the bank alone (with dummy counts the handle staged 8 MB and ran the count preparation -- 1.2 ms of
  // kernels for nothing, more than the thirteen bisection passes of a family)
  int rc = cb_create(device, S, 1, (int)nb, CB_F64, tt.data(), nullptr, CB_EXPM_ONLY, &h);
  if (rc != CB_OK) return rc;
  rc = cb_expm_bank(h, Q, pi, 0, logP);
  cb_destroy(h);
  if (rc != CB_OK) return rc;
  for (size_t i = 0; i < nb * SS; ++i) logP[i] = std::log(logP[i]);
  return CB_OK;
}

namespace {
std::vector<int8_t> ble_transposed(const int8_t *c, int n, int L) {
  std::vector<int8_t> t((size_t)n * L);
  for (int i = 0; i < n; ++i)
    for (int s = 0; s < L; ++s) t[(size_t)s * n + i] = c[(size_t)i * L + s];
  return t;
}
int ble_check(int device, int S, int T, int R, int n, int L, const int8_t *cx, const int8_t *cy) {
  if (S < 2 || S > 127 || T < 1 || R < 1 || n < 1 || L < 1) return fail(CB_EINVAL, "ble: bad sizes");
  const int ndev = cb_device_count();
  if (ndev <= 0) return fail(CB_EHIP, "ble: no HIP device visible");
  if (device < 0 || device >= ndev) return fail(CB_EINVAL, "ble: device %d out of range", device);
  for (size_t i = 0; i < (size_t)n * L; ++i)
    if (cx[i] >= S || cy[i] >= S) return fail(CB_EINVAL, "ble: state code out of range");
  return CB_OK;
}
}  // namespace

extern "C" int cb_ble_branch_lengths(int device, int S, int T, int R, const double *logP, const int8_t *cx,
                                     const int8_t *cy, int n, int L, const int *site_to_rate, int *lengths_index) {
  if (!logP || !cx || !cy || !site_to_rate || !lengths_index) return fail(CB_EINVAL, "cb_ble_branch_lengths: NULL argument");
  int rc = ble_check(device, S, T, R, n, L, cx, cy);
  if (rc != CB_OK) return rc;
  for (int i = 0; i < L; ++i)
    if (site_to_rate[i] < 0 || site_to_rate[i] >= R) return fail(CB_EINVAL, "cb_ble_branch_lengths: rate index out of range");
  HIP_TRY(hipSetDevice(device));
  CbDevBufs d;
  const double *dP = d.up(logP, (size_t)T * R * S * S, rc);
  const int8_t *dx = d.up(cx, (size_t)n * L, rc), *dy = d.up(cy, (size_t)n * L, rc);
  const int *ds = d.up(site_to_rate, L, rc);
  int *dout = d.up<int>(nullptr, n, rc);
  if (rc != CB_OK) return rc;
  hipLaunchKernelGGL(ble_branch_lengths_kernel, dim3((n + 3) / 4), dim3(256), 0, 0, S, T, R, n, L, dP, dx, dy, ds,
                     (const int *)nullptr, dout, (int *)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(lengths_index, dout, n * sizeof(int), hipMemcpyDeviceToHost));
  return CB_OK;
}

extern "C" int cb_ble_site_rates(int device, int S, int T, int R, const double *logP, const int8_t *cx,
                                 const int8_t *cy, int n, int L, const int *lengths_index, const double *priors,
                                 int *rate_index) {
  if (!logP || !cx || !cy || !lengths_index || !priors || !rate_index) return fail(CB_EINVAL, "cb_ble_site_rates: NULL argument");
  int rc = ble_check(device, S, T, R, n, L, cx, cy);
  if (rc != CB_OK) return rc;
  for (int i = 0; i < n; ++i)
    if (lengths_index[i] < 0 || lengths_index[i] >= T) return fail(CB_EINVAL, "cb_ble_site_rates: length index out of range");
  HIP_TRY(hipSetDevice(device));
  CbDevBufs d;
  const std::vector<int8_t> xT = ble_transposed(cx, n, L), yT = ble_transposed(cy, n, L);
  const double *dP = d.up(logP, (size_t)T * R * S * S, rc), *dpr = d.up(priors, R, rc);
  const int8_t *dxT = d.up(xT.data(), (size_t)n * L, rc), *dyT = d.up(yT.data(), (size_t)n * L, rc);
  const int *dl = d.up(lengths_index, n, rc);
  int *dout = d.up<int>(nullptr, L, rc);
  if (rc != CB_OK) return rc;
  HIP_TRY(hipStreamSynchronize(0));  // xT / yT are locals
  ble_launch_site_rates(S, T, R, n, L, dP, dxT, dyT, dl, dpr, dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(rate_index, dout, L * sizeof(int), hipMemcpyDeviceToHost));
  return CB_OK;
}

// initial site-rate bins (branch_length_estimation.cpp:10-58): sites ordered by the number of
// differing sequence pairs (ties: site index); the i-th site of that order gets category rc,
// rc advancing while i >= round(weights[rc] * L)
static int ble_initial_bins(const int8_t *all_seqs, int n_seqs, int L, int S, int R, const double *weights, int *s2r) {
  std::vector<long long> cnt((size_t)L * S, 0);
  for (int i = 0; i < n_seqs; ++i)
    for (int j = 0; j < L; ++j) {
      const int v = all_seqs[(size_t)i * L + j];
      if (v >= S) return fail(CB_EINVAL, "cb_ble: state code out of range");
      if (v >= 0) cnt[(size_t)j * S + v] += 1;
    }
  std::vector<std::pair<long long, int>> order(L);
  for (int j = 0; j < L; ++j) {
    long long non_missing = 0, total = 0;
    for (int k = 0; k < S; ++k) non_missing += cnt[(size_t)j * S + k];
    for (int k = 0; k < S; ++k) total += (non_missing - cnt[(size_t)j * S + k]) * cnt[(size_t)j * S + k];
    order[j] = {total, j};
  }
  std::sort(order.begin(), order.end());
  std::vector<long long> w(R);
  for (int r = 0; r < R; ++r) w[r] = (long long)std::llround(weights[r] * L);
  int cat = 0;
  for (int i = 0; i < L; ++i) {
    if (cat < R && i >= w[cat]) ++cat;
    s2r[order[i].second] = cat < R ? cat : R - 1;
  }
  return CB_OK;
}

extern "C" int cb_ble(int device, int S, int T, int R, const double *logP, const int8_t *cx, const int8_t *cy, int n,
                      int L, const int8_t *all_seqs, int n_seqs, const double *rates, const double *weights,
                      int max_iters, int *lengths_index, int *rate_index, int *iterations, double *kernel_ms) {
  if (!logP || !cx || !cy || !all_seqs || !rates || !weights || !lengths_index || !rate_index)
    return fail(CB_EINVAL, "cb_ble: NULL argument");
  int rc = ble_check(device, S, T, R, n, L, cx, cy);
  if (rc != CB_OK) return rc;
  if (n_seqs < 1 || max_iters < 0) return fail(CB_EINVAL, "cb_ble: bad sizes");
  std::vector<int> s2r(L, 0);
  if ((rc = ble_initial_bins(all_seqs, n_seqs, L, S, R, weights, s2r.data())) != CB_OK) return rc;
  std::vector<double> priors(R);
  for (int r = 0; r < R; ++r) priors[r] = 2 * std::log(rates[r]) - 3 * rates[r];   // :199-203
  HIP_TRY(hipSetDevice(device));
  CbDevBufs d;
  const double *dP = d.up(logP, (size_t)T * R * S * S, rc), *dpr = d.up(priors.data(), R, rc);
  const std::vector<int8_t> xT = ble_transposed(cx, n, L), yT = ble_transposed(cy, n, L);
  const int8_t *dx = d.up(cx, (size_t)n * L, rc), *dy = d.up(cy, (size_t)n * L, rc);
  const int8_t *dxT = d.up(xT.data(), (size_t)n * L, rc), *dyT = d.up(yT.data(), (size_t)n * L, rc);
  int *ds = d.up(s2r.data(), L, rc);
  int *dl0 = d.up<int>(nullptr, n, rc), *dl1 = d.up<int>(nullptr, n, rc), *dflag = d.up<int>(nullptr, 1, rc);
  if (rc != CB_OK) return rc;
  const dim3 gb((n + 3) / 4), gs((L + 3) / 4), blk(256);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (kernel_ms) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipStreamSynchronize(0));  // uploads done: the timed region starts with resident inputs
    HIP_TRY(hipEventRecord(ev0, 0));
  }
  hipLaunchKernelGGL(ble_branch_lengths_kernel, gb, blk, 0, 0, S, T, R, n, L, dP, dx, dy, (const int *)ds,
                     (const int *)nullptr, dl0, (int *)nullptr);
  bool match = false;
  int iters = 0;
  while (!match && max_iters) {
    ++iters;
    HIP_TRY(hipMemsetAsync(dflag, 0, sizeof(int), 0));
    ble_launch_site_rates(S, T, R, n, L, dP, dxT, dyT, (const int *)dl0, dpr, ds);
    hipLaunchKernelGGL(ble_branch_lengths_kernel, gb, blk, 0, 0, S, T, R, n, L, dP, dx, dy, (const int *)ds,
                       (const int *)dl0, dl1, dflag);
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, dflag, sizeof flag, hipMemcpyDeviceToHost));
    match = flag == 0;
    std::swap(dl0, dl1);
    --max_iters;
  }
  if (kernel_ms) {
    float ms = 0.f;
    HIP_TRY(hipEventRecord(ev1, 0));
    HIP_TRY(hipEventSynchronize(ev1));
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    *kernel_ms = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
  }
  if (iterations) *iterations = iters;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(lengths_index, dl0, n * sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(rate_index, ds, L * sizeof(int), hipMemcpyDeviceToHost));
  return CB_OK;
}

// Many families in one call (the reference maps families over a process pool, utils.py:59-67): the
// log-transition bank is uploaded ONCE, all sequences in one transfer, and the coordinate ascents run
// in lockstep -- per round two launches per still-moving family and ONE read-back of all the
// convergence flags (cb_ble: a synchronising read-back per family per iteration, and 8 MB of bank per
// family).  A converged family is a fixed point of the ascent, so results are those of cb_ble.
extern "C" int cb_ble_batch(int device, int S, int T, int R, const double *logP, int n_fam, const int *n,
                            const int *L, const int8_t *cx, const int8_t *cy, const int8_t *all_seqs,
                            const int *n_seqs, const double *rates, const double *weights, int max_iters,
                            int *lengths_index, int *rate_index, int *iterations, double *kernel_ms) {
  if (!logP || !n || !L || !cx || !cy || !all_seqs || !n_seqs || !rates || !weights || !lengths_index || !rate_index)
    return fail(CB_EINVAL, "cb_ble_batch: NULL argument");
  if (n_fam < 1 || max_iters < 0) return fail(CB_EINVAL, "cb_ble_batch: bad sizes");
  std::vector<size_t> off_c(n_fam + 1, 0), off_n(n_fam + 1, 0), off_L(n_fam + 1, 0), off_s(n_fam + 1, 0);
  for (int f = 0; f < n_fam; ++f) {
    if (n[f] < 1 || L[f] < 1 || n_seqs[f] < 1) return fail(CB_EINVAL, "cb_ble_batch: family %d has bad sizes", f);
    off_c[f + 1] = off_c[f] + (size_t)n[f] * L[f];
    off_n[f + 1] = off_n[f] + n[f];
    off_L[f + 1] = off_L[f] + L[f];
    off_s[f + 1] = off_s[f] + (size_t)n_seqs[f] * L[f];
  }
  int rc = CB_OK;
  std::vector<int> s2r(off_L[n_fam], 0);
  std::vector<int8_t> xT(off_c[n_fam]), yT(off_c[n_fam]);
  for (int f = 0; f < n_fam; ++f) {
    if ((rc = ble_check(device, S, T, R, n[f], L[f], cx + off_c[f], cy + off_c[f])) != CB_OK) return rc;
    if ((rc = ble_initial_bins(all_seqs + off_s[f], n_seqs[f], L[f], S, R, weights, s2r.data() + off_L[f])) != CB_OK) return rc;
    for (int i = 0; i < n[f]; ++i)
      for (int k = 0; k < L[f]; ++k) {
        xT[off_c[f] + (size_t)k * n[f] + i] = cx[off_c[f] + (size_t)i * L[f] + k];
        yT[off_c[f] + (size_t)k * n[f] + i] = cy[off_c[f] + (size_t)i * L[f] + k];
      }
  }
  std::vector<double> priors(R);
  for (int r = 0; r < R; ++r) priors[r] = 2 * std::log(rates[r]) - 3 * rates[r];   // :199-203
  HIP_TRY(hipSetDevice(device));
  CbDevBufs d;
  const double *dP = d.up(logP, (size_t)T * R * S * S, rc), *dpr = d.up(priors.data(), R, rc);
  const int8_t *dx = d.up(cx, off_c[n_fam], rc), *dy = d.up(cy, off_c[n_fam], rc);
  const int8_t *dxT = d.up(xT.data(), off_c[n_fam], rc), *dyT = d.up(yT.data(), off_c[n_fam], rc);
  int *ds = d.up(s2r.data(), off_L[n_fam], rc);
  int *dl0 = d.up<int>(nullptr, off_n[n_fam], rc), *dl1 = d.up<int>(nullptr, off_n[n_fam], rc);
  int *dflag = d.up<int>(nullptr, n_fam, rc);
  if (rc != CB_OK) return rc;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (kernel_ms) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipStreamSynchronize(0));  // uploads done: the timed region starts with resident inputs
    HIP_TRY(hipEventRecord(ev0, 0));
  }
  const dim3 blk(256);
  // cur[f]: which of the two length buffers holds family f's current lengths
  std::vector<int *> cur(n_fam), nxt(n_fam);
  for (int f = 0; f < n_fam; ++f) {
    cur[f] = dl0 + off_n[f];
    nxt[f] = dl1 + off_n[f];
    hipLaunchKernelGGL(ble_branch_lengths_kernel, dim3((n[f] + 3) / 4), blk, 0, 0, S, T, R, n[f], L[f], dP, dx + off_c[f],
                       dy + off_c[f], (const int *)(ds + off_L[f]), (const int *)nullptr, cur[f], (int *)nullptr);
  }
  std::vector<int> active(n_fam), iters(n_fam, 0), flags(n_fam);
  for (int f = 0; f < n_fam; ++f) active[f] = f;
  for (int round = 0; round < max_iters && !active.empty(); ++round) {
    HIP_TRY(hipMemsetAsync(dflag, 0, n_fam * sizeof(int), 0));
    for (int f : active) {
      ble_launch_site_rates(S, T, R, n[f], L[f], dP, dxT + off_c[f], dyT + off_c[f], (const int *)cur[f], dpr, ds + off_L[f]);
      hipLaunchKernelGGL(ble_branch_lengths_kernel, dim3((n[f] + 3) / 4), blk, 0, 0, S, T, R, n[f], L[f], dP, dx + off_c[f],
                         dy + off_c[f], (const int *)(ds + off_L[f]), (const int *)cur[f], nxt[f], dflag + f);
    }
    HIP_TRY(hipMemcpy(flags.data(), dflag, n_fam * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> still;
    for (int f : active) {
      ++iters[f];
      std::swap(cur[f], nxt[f]);
      if (flags[f] != 0) still.push_back(f);
    }
    active.swap(still);
  }
  if (kernel_ms) {
    float ms = 0.f;
    HIP_TRY(hipEventRecord(ev1, 0));
    HIP_TRY(hipEventSynchronize(ev1));
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    *kernel_ms = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
  }
  HIP_TRY(hipGetLastError());
  for (int f = 0; f < n_fam; ++f) {
    HIP_TRY(hipMemcpyAsync(lengths_index + off_n[f], cur[f], n[f] * sizeof(int), hipMemcpyDeviceToHost, 0));
    if (iterations) iterations[f] = iters[f];
  }
  HIP_TRY(hipMemcpyAsync(rate_index, ds, off_L[n_fam] * sizeof(int), hipMemcpyDeviceToHost, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return CB_OK;
}

extern "C" int cb_site_rate_gather(int device, int S, int R, int n, int L, const double *tens, const int8_t *cx,
                                   const int8_t *cy, const double *log_prior, int *best) {
  if (!tens || !cx || !cy || !log_prior || !best) return fail(CB_EINVAL, "cb_site_rate_gather: NULL argument");
  int rc = ble_check(device, S, 1, R, n, L, cx, cy);
  if (rc != CB_OK) return rc;
  for (size_t i = 0; i < (size_t)n * L; ++i)
    if (cx[i] < 0 || cy[i] < 0) return fail(CB_EINVAL, "cb_site_rate_gather: negative state code (map gaps to a state)");
  HIP_TRY(hipSetDevice(device));
  CbDevBufs d;
  const double *dt = d.up(tens, (size_t)R * n * S * S, rc), *dpr = d.up(log_prior, R, rc);
  const int8_t *dx = d.up(cx, (size_t)n * L, rc), *dy = d.up(cy, (size_t)n * L, rc);
  int *dout = d.up<int>(nullptr, L, rc);
  if (rc != CB_OK) return rc;
  hipLaunchKernelGGL(site_rate_gather_kernel, dim3((L + 3) / 4), dim3(256), 0, 0, S, R, n, L, dt, dx, dy, dpr, dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(best, dout, L * sizeof(int), hipMemcpyDeviceToHost));
  return CB_OK;
}
