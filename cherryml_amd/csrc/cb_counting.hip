// libcherrybank: transition counting and SiteRM count assembly.
#include "cb_internal.hip.h"
#include "common.hip.h"
#include "counting.hip.h"

// ------------------------------------------------------------------------ counting
// Replica scratch of the resident (device-pointer) form, kept per device for the life of the
// process (the only process-wide state of the library; never holds results between calls).
#include <mutex>
// (one lock for the whole table: the counting entry points are not meant to run concurrently on one device -- their kernels
// share this scratch on the default stream -- but two host threads growing it at once must not corrupt the table)
static std::mutex g_scratch_mutex;
static int count_scratch(int device, size_t elems, unsigned long long **out, bool *moved = nullptr) {
  static unsigned long long *buf[64] = {};
  static size_t cap[64] = {};
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (device < 0 || device >= 64) return fail(CB_EINVAL, "counting: device %d out of range", device);
  if (moved) *moved = cap[device] < elems;   // a re-allocation loses the contents -- and may return the SAME address
  if (cap[device] < elems) {
    if (buf[device]) (void)hipFree(buf[device]);
    buf[device] = nullptr;
    cap[device] = 0;
    hipError_t e = hipMalloc((void **)&buf[device], elems * sizeof(unsigned long long));
    if (e != hipSuccess) return fail(CB_ENOMEM, "counting: replica scratch allocation failed");
    cap[device] = elems;
  }
  *out = buf[device];
  return CB_OK;
}

// device-pointer launch of the single-site counter: adds into counts[B*S*S].
// max_sites = largest pair.n (0 = unknown -> replica path).
static int launch_count_transitions(int device, int S, int B, const double *grid, const int8_t *seqs,
                                    const double *rates, const cb_count_pair *pairs, int64_t n_pairs,
                                    int symmetric, int max_sites, unsigned long long *counts) {
  const size_t nb = (size_t)B * S * S;
  const int words = (int)((nb + 1) / 2);
  const size_t lds = (size_t)((words + 1) & ~1) * sizeof(unsigned) + (size_t)B * sizeof(double);
  int chunk = max_sites > 0 ? 65535 / (2 * max_sites) : 0;
  if (chunk >= 8 && lds <= 150 * 1024) {
    // keep at least ~2 workgroups per CU worth of slabs when there is enough work
    const int64_t want = (n_pairs + 511) / 512;
    if (want < chunk) chunk = (int)(want > 8 ? want : 8);
    const int64_t nwg = (n_pairs + chunk - 1) / chunk;
    unsigned long long *scratch = nullptr;
    int rc = count_scratch(device, ((size_t)nwg * words + 1) / 2 + 1, &scratch);
    if (rc != CB_OK) return rc;
    unsigned *slabs = reinterpret_cast<unsigned *>(scratch);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(count_transitions_lds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(count_transitions_lds_kernel, dim3((unsigned)nwg), dim3(CNT_LDS_THREADS), lds, 0, S, B,
                       grid, seqs, rates, pairs, (long long)n_pairs, symmetric, chunk, slabs, words);
    hipLaunchKernelGGL(count_reduce_slabs, dim3((unsigned)((words + 63) / 64)), dim3(256), 0, 0, slabs, (int)nwg,
                       words, nb, counts);
  } else {
    unsigned long long *rep = nullptr;
    int rc = count_scratch(device, nb * CNT_REPLICAS, &rep);
    if (rc != CB_OK) return rc;
    HIP_TRY(hipMemsetAsync(rep, 0, nb * CNT_REPLICAS * sizeof(unsigned long long), 0));
    const unsigned blocks = (unsigned)((n_pairs + 3) / 4);
    hipLaunchKernelGGL(count_transitions_kernel, dim3(blocks), dim3(256), 0, 0, S, B, grid, seqs, rates, pairs,
                       (long long)n_pairs, symmetric, rep, CNT_REPLICAS);
    hipLaunchKernelGGL(count_reduce_replicas, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, rep,
                       CNT_REPLICAS, nb, counts);
  }
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

// device-pointer launch of the co-transition counter: adds into counts[B * S^2 * S^2] (see counting.hip.h).
// total_events = sum of pair.n when the caller knows it (host form); the resident form reads it back from the device
// after the first two kernels (one synchronisation).
static int launch_count_co_transitions(int device, int S, int B, const double *grid, const int8_t *seqs,
                                       const int32_t *contacts, const cb_count_pair *pairs, int64_t n_pairs,
                                       int symmetric, int64_t total_events, int max_n,
                                       unsigned long long *counts) {
  const int S2 = S * S;
  const int R = std::min(S2, CO_LDS_WORDS / S2);
#ifdef CB_CO_PLAIN   // (build-time experiment: time the scattered-global-atomic form this path replaced)
  const bool plain = true;
#else
  const bool plain = false;
#endif
  // (co_expand loads the contacts as int2: a list that is not 8-byte aligned -- an offset view of a device buffer -- takes the
  // plain form too, which reads them as 4-byte words; same counts)
  const bool misaligned = (reinterpret_cast<uintptr_t>(contacts) & 7) != 0;
  if (plain || misaligned || R < 1 || (size_t)2 * B * 8 > 64 * 1024) {   // S > 200 or an enormous grid: the plain atomic form
    const unsigned blocks = (unsigned)((n_pairs + 3) / 4);
    hipLaunchKernelGGL(count_co_transitions_kernel, dim3(blocks), dim3(256), 0, 0, S, B, grid, seqs, contacts, pairs,
                       (long long)n_pairs, symmetric, counts);
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  const int nrb = (S2 + R - 1) / R;
#ifndef CO_TARGET
#define CO_TARGET 1024   // measured on the 10,000-family bench input: 512 -> 0.684, 1024 -> 0.659, 2048 -> 0.760, 4096 -> 0.990 ms
#endif
  const int target = CO_TARGET;   // work items of the whole pass (about; see co_plan_kernel)
  const int max_work = target + (B + 8) * nrb;
  const unsigned pair_blocks = (unsigned)((n_pairs + CO_THREADS - 1) / CO_THREADS);
  // scratch layout (8-byte words): bucket_ev[B] cursor[B] | bucket_off[B+1] n_work[1] work[4 * max_work] qbuf events
  const size_t head = (size_t)2 * B, fixed = head + (B + 1) + 1 + (size_t)4 * max_work + ((size_t)n_pairs + 1) / 2;
  unsigned long long *scr = nullptr;
  // host form: the event total is known, so the scratch gets its final size NOW (the two-step growth below -- and its
  // repeated launches -- is for the resident form only, which learns the total from the device)
  int rc = count_scratch(device, total_events >= 0 ? fixed + ((size_t)total_events + 1) / 2 + 1 : fixed, &scr);
  if (rc != CB_OK) return rc;
  HIP_TRY(hipMemsetAsync(scr, 0, head * sizeof(unsigned long long), 0));
  auto carve = [&](unsigned long long *base) {
    struct { unsigned long long *bucket_ev, *cursor, *bucket_off; int *n_work; CoWork *work; int *qbuf; unsigned *events; } w;
    w.bucket_ev = base;
    w.cursor = base + B;
    w.bucket_off = base + head;
    w.n_work = reinterpret_cast<int *>(base + head + B + 1);
    w.work = reinterpret_cast<CoWork *>(base + head + B + 2);
    w.qbuf = reinterpret_cast<int *>(base + head + B + 2 + (size_t)4 * max_work);
    w.events = reinterpret_cast<unsigned *>(base + fixed);
    return w;
  };
  auto w = carve(scr);
  hipLaunchKernelGGL(co_bucket_kernel, dim3(pair_blocks), dim3(CO_THREADS), (size_t)B * 8, 0, B, grid, pairs,
                     (long long)n_pairs, w.qbuf, w.bucket_ev);
  hipLaunchKernelGGL(co_plan_kernel, dim3(1), dim3(256), (size_t)2 * B * 8, 0, B, nrb, target, max_work, w.bucket_ev,
                     w.bucket_off, w.work, w.n_work);
  HIP_TRY(hipGetLastError());
  if (total_events < 0) {
    // resident form: the pairs live on the device, so the event total is read back after the first two kernels (one
    // 8-byte copy, ~15 us).  The caller's bound on pair.n (`max_n`, flags bits 8..23) is only a plausibility check: sizing
    // the event array from it would turn a wrong bound into out-of-bounds stores.
    unsigned long long tot = 0;
    HIP_TRY(hipMemcpy(&tot, w.bucket_off + B, sizeof tot, hipMemcpyDeviceToHost));
    if (max_n > 0 && tot > (unsigned long long)n_pairs * (unsigned long long)max_n)
      return fail(CB_EINVAL, "cb_count_co_transitions: %llu events exceed n_pairs x the stated largest pair.n (%d)", tot, max_n);
    total_events = (int64_t)tot;
  }
  if (total_events == 0) return CB_OK;
  // the event array behind the fixed part: growing the scratch would move (and lose) the part already filled, so the
  // scratch is grown FIRST when it is too small and the two kernels above are simply run again
  const size_t need = fixed + ((size_t)total_events + 1) / 2 + 1;
  unsigned long long *scr2 = nullptr;
  bool moved = false;   // (comparing the pointers is not enough: freeing and re-allocating a block can hand the address back)
  rc = count_scratch(device, need, &scr2, &moved);
  if (rc != CB_OK) return rc;
  if (moved) {
    scr = scr2;
    w = carve(scr);
    HIP_TRY(hipMemsetAsync(scr, 0, head * sizeof(unsigned long long), 0));
    hipLaunchKernelGGL(co_bucket_kernel, dim3(pair_blocks), dim3(CO_THREADS), (size_t)B * 8, 0, B, grid, pairs,
                       (long long)n_pairs, w.qbuf, w.bucket_ev);
    hipLaunchKernelGGL(co_plan_kernel, dim3(1), dim3(256), (size_t)2 * B * 8, 0, B, nrb, target, max_work, w.bucket_ev,
                       w.bucket_off, w.work, w.n_work);
  }
  hipLaunchKernelGGL(co_expand_kernel, dim3((unsigned)((n_pairs + CO_XP - 1) / CO_XP)), dim3(CO_XP), (size_t)B * 8, 0, B,
                     seqs, contacts, pairs, (long long)n_pairs, w.qbuf, w.bucket_off, w.cursor, w.events);
  const size_t lds = (size_t)R * S2 * sizeof(unsigned);
  if (symmetric) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(co_count_lds_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(co_count_lds_kernel<true>, dim3(max_work), dim3(CO_THREADS), lds, 0, S, R, w.events, w.work,
                       w.n_work, counts);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(co_count_lds_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(co_count_lds_kernel<false>, dim3(max_work), dim3(CO_THREADS), lds, 0, S, R, w.events, w.work,
                       w.n_work, counts);
  }
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

static int count_common(int device, int S, int B, const double *grid, const int8_t *seqs,
                        int64_t seqs_bytes, const void *aux, size_t aux_bytes,
                        const cb_count_pair *pairs, int64_t n_pairs, int symmetric, int flags,
                        unsigned long long *counts, bool co) {
  if (S < 1 || B < 1 || !grid || !counts) return fail(CB_EINVAL, "counting: bad S/B/grid/counts");
  if (flags & CB_PTR_DEVICE) {  // resident form: enqueue only, add into counts
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs || !seqs))) return fail(CB_EINVAL, "counting: bad pairs");
    HIP_TRY(hipSetDevice(device));
    if (n_pairs > 0) {
      const unsigned blocks = (unsigned)((n_pairs + 3) / 4);
      (void)blocks;
      if (co) {
        // resident form: the caller states the largest pair.n in flags bits 8..23 (0 = unknown: one read-back)
        int rc = launch_count_co_transitions(device, S, B, grid, seqs, (const int32_t *)aux, pairs, n_pairs, symmetric,
                                             -1, (flags >> 8) & 0xFFFF, counts);
        if (rc != CB_OK) return rc;
      } else {
        // resident form: the caller states the largest pair.n in flags bits 8..23 (0 = unknown)
        int rc = launch_count_transitions(device, S, B, grid, seqs, (const double *)aux, pairs, n_pairs,
                                          symmetric, (flags >> 8) & 0xFFFF, counts);
        if (rc != CB_OK) return rc;
      }
      HIP_TRY(hipGetLastError());
    }
    return CB_OK;
  }
  if (n_pairs < 0 || (n_pairs > 0 && (!pairs || !seqs))) return fail(CB_EINVAL, "counting: bad pairs");
  for (int b = 1; b < B; ++b)
    if (!(grid[b] > grid[b - 1])) return fail(CB_EINVAL, "counting: quantization points must be sorted");
  int ndev = cb_device_count();
  if (ndev <= 0) return fail(CB_EHIP, "counting: no HIP device visible");
  if (device < 0 || device >= ndev) return fail(CB_EINVAL, "counting: device %d out of range", device);
  // validate offsets on the host: the kernels trust them
  const size_t nbins = co ? (size_t)B * S * S * S * S : (size_t)B * S * S;
  for (int64_t p = 0; p < n_pairs; ++p) {
    const cb_count_pair &pr = pairs[p];
    const int64_t span = co ? 1 : pr.n;  // co: sites are indexed through the contact list
    if (pr.n < 0 || pr.seq_a < 0 || pr.seq_b < 0 || pr.aux < 0 || pr.seq_a + span > seqs_bytes ||
        pr.seq_b + span > seqs_bytes)
      return fail(CB_EINVAL, "counting: pair %lld has offsets outside the sequence buffer", (long long)p);
    const size_t need = co ? ((size_t)pr.aux + pr.n) * 2 * sizeof(int32_t) : ((size_t)pr.aux + pr.n) * sizeof(double);
    if (need > aux_bytes) return fail(CB_EINVAL, "counting: pair %lld reads past its rates/contacts", (long long)p);
  }
  HIP_TRY(hipSetDevice(device));
  double *d_grid = nullptr;
  int8_t *d_seqs = nullptr;
  void *d_aux = nullptr;
  cb_count_pair *d_pairs = nullptr;
  unsigned long long *d_counts = nullptr;
  int rc = CB_OK;
  auto freeall = [&]() {
    (void)hipFree(d_grid); (void)hipFree(d_seqs); (void)hipFree(d_aux); (void)hipFree(d_pairs); (void)hipFree(d_counts);
  };
#define TRYC(expr)                                                                       \
  if (rc == CB_OK) {                                                                     \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) rc = fail(e_ == hipErrorOutOfMemory ? CB_ENOMEM : CB_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  }
  TRYC(hipMalloc((void **)&d_grid, B * sizeof(double)));
  TRYC(hipMalloc((void **)&d_seqs, seqs_bytes > 0 ? seqs_bytes : 1));
  TRYC(hipMalloc(&d_aux, aux_bytes > 0 ? aux_bytes : 1));
  TRYC(hipMalloc((void **)&d_pairs, (n_pairs > 0 ? n_pairs : 1) * sizeof(cb_count_pair)));
  TRYC(hipMalloc((void **)&d_counts, nbins * sizeof(unsigned long long)));
  TRYC(hipMemcpy(d_grid, grid, B * sizeof(double), hipMemcpyHostToDevice));
  if (seqs_bytes > 0) TRYC(hipMemcpy(d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice));
  if (aux_bytes > 0) TRYC(hipMemcpy(d_aux, aux, aux_bytes, hipMemcpyHostToDevice));
  if (n_pairs > 0) TRYC(hipMemcpy(d_pairs, pairs, n_pairs * sizeof(cb_count_pair), hipMemcpyHostToDevice));
  TRYC(hipMemset(d_counts, 0, nbins * sizeof(unsigned long long)));
  if (rc == CB_OK && n_pairs > 0) {
    if (co) {
      int64_t total_events = 0;
      for (int64_t p = 0; p < n_pairs; ++p) total_events += pairs[p].n > 0 ? pairs[p].n : 0;
      rc = launch_count_co_transitions(device, S, B, d_grid, d_seqs, (const int32_t *)d_aux, d_pairs, n_pairs, symmetric,
                                       total_events, 0, d_counts);
    } else {
      int max_sites = 0;
      for (int64_t p = 0; p < n_pairs; ++p) max_sites = pairs[p].n > max_sites ? pairs[p].n : max_sites;
      if (rc == CB_OK)
        rc = launch_count_transitions(device, S, B, d_grid, d_seqs, (const double *)d_aux, d_pairs, n_pairs,
                                      symmetric, max_sites < 32768 ? max_sites : 0, d_counts);
    }
    TRYC(hipGetLastError());
    TRYC(hipDeviceSynchronize());
  }
  TRYC(hipMemcpy(counts, d_counts, nbins * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#undef TRYC
  freeall();
  return rc;
}

extern "C" int cb_count_transitions(int device, int S, int B, const double *grid, const int8_t *seqs,
                                    int64_t seqs_bytes, const double *rates, int64_t n_rates,
                                    const cb_count_pair *pairs, int64_t n_pairs, int symmetric,
                                    int flags, unsigned long long *counts) {
  if (S > 127) return fail(CB_EINVAL, "cb_count_transitions: at most 127 states (int8 codes)");
  return count_common(device, S, B, grid, seqs, seqs_bytes, rates, (size_t)(n_rates > 0 ? n_rates : 0) * sizeof(double),
                      pairs, n_pairs, symmetric, flags, counts, false);
}

extern "C" int cb_count_co_transitions(int device, int S, int B, const double *grid, const int8_t *seqs,
                                       int64_t seqs_bytes, const int32_t *contacts, int64_t n_contacts,
                                       const cb_count_pair *pairs, int64_t n_pairs, int symmetric,
                                       int flags, unsigned long long *counts) {
  if (S > 127) return fail(CB_EINVAL, "cb_count_co_transitions: at most 127 states (int8 codes)");
  // contact indices must address sites inside the sequences: checked per pair on the host
  for (int64_t p = 0; !(flags & CB_PTR_DEVICE) && p < n_pairs && pairs && contacts; ++p) {
    const cb_count_pair &pr = pairs[p];
    if (pr.aux < 0 || pr.n < 0 || pr.aux + pr.n > n_contacts)
      return fail(CB_EINVAL, "cb_count_co_transitions: pair %lld contact range outside the list", (long long)p);
    for (int c = 0; c < pr.n; ++c) {
      const int32_t i = contacts[2 * (pr.aux + c)], j = contacts[2 * (pr.aux + c) + 1];
      if (i < 0 || j < 0 || pr.seq_a + i >= seqs_bytes || pr.seq_a + j >= seqs_bytes ||
          pr.seq_b + i >= seqs_bytes || pr.seq_b + j >= seqs_bytes)
        return fail(CB_EINVAL, "cb_count_co_transitions: contact site outside the sequence buffer");
    }
  }
  return count_common(device, S, B, grid, seqs, seqs_bytes, contacts,
                      (size_t)(n_contacts > 0 ? n_contacts : 0) * 2 * sizeof(int32_t), pairs, n_pairs, symmetric,
                      flags, counts, true);
}

// MANY families in one call: the site axis is the families' sites concatenated (family f owns rows
// [sum_{g<f} n_sites[g], +n_sites[f]) of site_rates and counts), pairs are the families' transitions
// concatenated (n_pairs[f] each, offsets into the one seqs buffer).  Sites are independent, so the
// result is cb_siterm_assemble's family by family -- and the tensor feeds ONE cb_create / cb_train_siterm
// over all families' sites.
extern "C" int cb_siterm_assemble_batch(int device, int S, int B, int n_fam, const int *n_sites, const double *grid,
                                        const int8_t *seqs, int64_t seqs_bytes, const cb_count_pair *pairs,
                                        const int64_t *n_pairs, const double *site_rates, const double *prior,
                                        double lambda, int include_reverse, int flags, double *counts,
                                        double *kernel_ms) {
  if (!grid || !seqs || !pairs || !site_rates || !prior || !counts || !n_sites || !n_pairs)
    return fail(CB_EINVAL, "cb_siterm_assemble: NULL argument");
  if (S < 2 || S > 64 || B < 1 || n_fam < 1 || seqs_bytes < 0)
    return fail(CB_EINVAL, "cb_siterm_assemble: bad sizes (S=%d, B=%d, families=%d)", S, B, n_fam);
  if (!(lambda >= 0.0 && lambda <= 1.0)) return fail(CB_EINVAL, "cb_siterm_assemble: lambda must be in [0, 1]");
  for (int b = 1; b < B; ++b)
    if (!(grid[b] > grid[b - 1])) return fail(CB_EINVAL, "cb_siterm_assemble: grid must be strictly increasing");
  int64_t tot_sites = 0, tot_pairs = 0;
  for (int f = 0; f < n_fam; ++f) {
    if (n_sites[f] < 1 || n_pairs[f] < 0)
      return fail(CB_EINVAL, "cb_siterm_assemble: family %d has n_sites = %d, n_pairs = %lld", f, n_sites[f], (long long)n_pairs[f]);
    tot_sites += n_sites[f];
    tot_pairs += n_pairs[f];
  }
  if (tot_sites > INT32_MAX) return fail(CB_EINVAL, "cb_siterm_assemble: %lld sites in one call", (long long)tot_sites);
  // the kernel's view of a transition: its family's first site (aux) and site count (n)
  std::vector<cb_count_pair> own(pairs, pairs + tot_pairs);
  {
    int64_t p = 0, site0 = 0;
    for (int f = 0; f < n_fam; ++f) {
      for (int64_t k = 0; k < n_pairs[f]; ++k, ++p) {
        if (own[p].seq_a < 0 || own[p].seq_b < 0 || own[p].seq_a + n_sites[f] > seqs_bytes || own[p].seq_b + n_sites[f] > seqs_bytes)
          return fail(CB_EINVAL, "cb_siterm_assemble: pair %lld points outside seqs", (long long)p);
        own[p].aux = (int32_t)site0;
        own[p].n = n_sites[f];
      }
      site0 += n_sites[f];
    }
  }
  const int64_t n_pairs_all = tot_pairs;
  const int ndev = cb_device_count();
  if (ndev <= 0) return fail(CB_EHIP, "cb_siterm_assemble: no HIP device visible");
  if (device < 0 || device >= ndev) return fail(CB_EINVAL, "cb_siterm_assemble: device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  const size_t SS = (size_t)S * S, nmat = (size_t)tot_sites * B, ncounts = nmat * SS;
  void *d_grid = nullptr, *d_seqs = nullptr, *d_pairs = nullptr, *d_rates = nullptr, *d_prior = nullptr,
       *d_live = nullptr, *d_counts_own = nullptr;
  int rc = CB_OK;
#define TRYA(expr) \
  if (rc == CB_OK && (expr) != hipSuccess) rc = fail(CB_EHIP, "cb_siterm_assemble: %s failed", #expr)
  TRYA(hipMalloc(&d_grid, B * sizeof(double)));
  TRYA(hipMalloc(&d_seqs, seqs_bytes > 0 ? seqs_bytes : 1));
  TRYA(hipMalloc(&d_pairs, (n_pairs_all > 0 ? n_pairs_all : 1) * sizeof(cb_count_pair)));
  TRYA(hipMalloc(&d_rates, tot_sites * sizeof(double)));
  TRYA(hipMalloc(&d_prior, (size_t)B * SS * sizeof(double)));
  TRYA(hipMalloc(&d_live, nmat * sizeof(int)));
  double *d_counts = counts;
  if (!(flags & CB_PTR_DEVICE)) {
    TRYA(hipMalloc(&d_counts_own, ncounts * sizeof(double)));
    d_counts = static_cast<double *>(d_counts_own);
  }
  TRYA(hipMemcpyAsync(d_grid, grid, B * sizeof(double), hipMemcpyHostToDevice, 0));
  TRYA(hipMemcpyAsync(d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice, 0));
  TRYA(hipMemcpyAsync(d_pairs, own.data(), n_pairs_all * sizeof(cb_count_pair), hipMemcpyHostToDevice, 0));
  TRYA(hipMemcpyAsync(d_rates, site_rates, tot_sites * sizeof(double), hipMemcpyHostToDevice, 0));
  TRYA(hipMemcpyAsync(d_prior, prior, (size_t)B * SS * sizeof(double), hipMemcpyHostToDevice, 0));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (kernel_ms && rc == CB_OK) {
    TRYA(hipEventCreate(&ev0));
    TRYA(hipEventCreate(&ev1));
    TRYA(hipStreamSynchronize(0));
    TRYA(hipEventRecord(ev0, 0));
  }
  TRYA(hipMemsetAsync(d_live, 0, nmat * sizeof(int), 0));
  TRYA(hipMemsetAsync(d_counts, 0, ncounts * sizeof(double), 0));
  if (rc == CB_OK) {
    if (n_pairs_all > 0)
      hipLaunchKernelGGL(siterm_raw_counts_kernel, dim3((unsigned)((n_pairs_all + 3) / 4)), dim3(256), 0, 0, S, B,
                         (const double *)d_grid, (const int8_t *)d_seqs, (const cb_count_pair *)d_pairs,
                         (long long)n_pairs_all, d_counts, (int *)d_live);
    hipLaunchKernelGGL(siterm_mix_kernel, dim3((unsigned)nmat), dim3(64), 0, 0, S, B, (const double *)d_grid,
                       (const double *)d_rates, (const double *)d_prior, lambda, include_reverse,
                       (const int *)d_live, d_counts);
    TRYA(hipGetLastError());
  }
  if (kernel_ms && rc == CB_OK) {
    float ms = 0.f;
    TRYA(hipEventRecord(ev1, 0));
    TRYA(hipEventSynchronize(ev1));
    TRYA(hipEventElapsedTime(&ms, ev0, ev1));
    *kernel_ms = ms;
  }
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  if (!(flags & CB_PTR_DEVICE)) TRYA(hipMemcpyAsync(counts, d_counts, ncounts * sizeof(double), hipMemcpyDeviceToHost, 0));
  TRYA(hipStreamSynchronize(0));
#undef TRYA
  for (void *q : {d_grid, d_seqs, d_pairs, d_rates, d_prior, d_live, d_counts_own})
    if (q) (void)hipFree(q);
  return rc;
}

extern "C" int cb_siterm_assemble(int device, int S, int B, int n_sites, const double *grid, const int8_t *seqs,
                                  int64_t seqs_bytes, const cb_count_pair *pairs, int64_t n_pairs,
                                  const double *site_rates, const double *prior, double lambda,
                                  int include_reverse, int flags, double *counts, double *kernel_ms) {
  if (n_sites < 1 || n_pairs < 0) return fail(CB_EINVAL, "cb_siterm_assemble: bad sizes (S=%d, B=%d, n_sites=%d)", S, B, n_sites);
  return cb_siterm_assemble_batch(device, S, B, 1, &n_sites, grid, seqs, seqs_bytes, pairs, &n_pairs, site_rates, prior,
                                  lambda, include_reverse, flags, counts, kernel_ms);
}

// ------------------------------------------------------------------------ JTT-IPW statistics
extern "C" int cb_jtt_ipw_stats(int device, int S, int B, const void *counts, int counts_f64, const double *grid, double unit,
                                int symmetrize, int flags, double *F, double *R) {
  if (S <= 0 || B <= 0 || !counts || !grid || !F || !R) return fail(CB_EINVAL, "cb_jtt_ipw_stats: bad argument");
  HIP_TRY(hipSetDevice(device));
  const size_t SS = (size_t)S * S, nC = (size_t)B * SS;
  const bool devp = (flags & CB_PTR_DEVICE) != 0;
  const void *dC = counts;
  const double *dgrid = grid;
  double *dF = F, *dR = R;
  void *tmp[3] = {nullptr, nullptr, nullptr};
  auto cleanup = [&]() {
    for (void *p : tmp)
      if (p) (void)hipFree(p);
  };
  if (!devp) {
    if (hipMalloc(&tmp[0], nC * 8) != hipSuccess || hipMalloc(&tmp[1], (size_t)B * 8) != hipSuccess ||
        hipMalloc(&tmp[2], 2 * SS * 8) != hipSuccess) {
      cleanup();
      return fail(CB_ENOMEM, "cb_jtt_ipw_stats: device allocation failed");
    }
    if (hipMemcpy(tmp[0], counts, nC * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(tmp[1], grid, (size_t)B * 8, hipMemcpyHostToDevice) != hipSuccess) {
      cleanup();
      return fail(CB_EHIP, "cb_jtt_ipw_stats: upload failed");
    }
    dC = tmp[0];
    dgrid = static_cast<const double *>(tmp[1]);
    dF = static_cast<double *>(tmp[2]);
    dR = dF + SS;
  }
  const int nt = (S + 15) / 16, npairs = nt * (nt + 1) / 2, nchunks = (B + JT_BC - 1) / JT_BC;
  unsigned long long *scratch = nullptr;
  int rc = count_scratch(device, (size_t)nchunks * 2 * SS, &scratch);
  if (rc != CB_OK) {
    cleanup();
    return rc;
  }
  double *part = reinterpret_cast<double *>(scratch);
  // (the padded rows / columns of edge tiles are never written; jtt_stats_reduce only reads entries inside S x S)
  const dim3 g((unsigned)npairs, (unsigned)nchunks);
  if (counts_f64)
    hipLaunchKernelGGL(jtt_stats_partial<double>, g, dim3(256), 0, 0, S, B, static_cast<const double *>(dC), dgrid, unit,
                       symmetrize, part);
  else
    hipLaunchKernelGGL(jtt_stats_partial<unsigned long long>, g, dim3(256), 0, 0, S, B,
                       static_cast<const unsigned long long *>(dC), dgrid, unit, symmetrize, part);
  hipLaunchKernelGGL(jtt_stats_reduce, dim3((unsigned)((2 * SS + 255) / 256)), dim3(256), 0, 0, 2 * SS, nchunks, part, dF, dR);
  if (hipGetLastError() != hipSuccess) {
    cleanup();
    return fail(CB_EHIP, "cb_jtt_ipw_stats: launch failed");
  }
  if (!devp) {
    const hipError_t e1 = hipMemcpy(F, dF, SS * 8, hipMemcpyDeviceToHost), e2 = hipMemcpy(R, dR, SS * 8, hipMemcpyDeviceToHost);
    cleanup();
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(CB_EHIP, "cb_jtt_ipw_stats: download failed");
  }
  return CB_OK;
}
