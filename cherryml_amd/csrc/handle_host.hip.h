// The handle (`struct cb_bank`) and its host-side helpers: event sets of the phase timer, device / workspace /
// pinned-staging allocation.  Included by cherrybank.hip (one translation unit; see eigh_large_host.hip.h).
#pragma once
#include <hip/hip_ext.h>
#include <future>
// ------------------------------------------------------------------ handle
// plan of a device-controlled warm eigensolve (eigh_planned_host.hip.h)
struct EighSlot {
  int cap = 12;         // polynomial order the slot's launches can evaluate: 2 (one product), 4 (two) or 12 (powers, polynomial, two more)
  int nsq = 0;          // squaring launches
  int band_after = 0;   // a banded Jacobi pass behind the sweep (runs when the sweep was a masked one)
  int so = 0;           // the second-order launch (lge_so; runs when the sweep is an all-pairs one above 1e-8)
  // what the previous solve did at this position (timing only, never results): launches expected to RUN request their
  // decision-independent operands before they look at the control block (EgArgs::early)
  int expect_run = 1, expect_order = 12, expect_sq = 2;
};
struct EighPlan {
  int lead_band = 1;    // a banded Jacobi pass before the first sweep
  int nslots = 0;
  EighSlot slot[EC_MAXREC];
};
struct cb_bank {
  int dev = 0, S = 0, L = 0, B = 0;
  int B_cap = 0;        // B at creation (cb_internal_set_times may lower B)
  // the optimisation a later CB_TRAIN_RESUME call continues (large fused trainer): epochs done, Adam's beta powers,
  // and what the call looked like (mask, moments) -- a resumed call must look the same
  hipStream_t xstream[3] = {};     // CB_BANK_STREAMS: extra queues, each with its share of the buckets
  hipEvent_t ev_fork = nullptr, ev_join[3] = {};
  int tr_epochs = 0;
  uint64_t tr_sig = 0;
  int last_form = 0;    // which trainer kernels the last training call launched (cb_last_kernel_form)
  double tr_pow_b1 = 1.0, tr_pow_b2 = 1.0;
  int dtype = CB_F64;   // element type of the bank products (large path): CB_F64 or CB_F32
  int LD = 0;           // large path: padded leading dimension
  bool large = false;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<void *> allocs;
  std::vector<double> n_host;  // [L]
  // resident bank
  double *t = nullptr;       // [L,B]
  double *Ct = nullptr;      // small: [L,B,S,S]; large: [B,LD,LD]
  double *n_dev = nullptr;   // [L]
  double *inv_n = nullptr;   // [L]  1/n
  double *ones = nullptr;    // [L]  1.0
  double *Cq = nullptr;      // S <= 24: counts in quad order [L][nq][TS*TS][64]
  int nq = 0;
  double *dirsum = nullptr;  // [L,S] colsum - rowsum of sum_b C
  double *dirsum_g = nullptr;  // the same summed over the ranks (cb_allreduce_setup; the trainers' direct pi term)
  // live buckets (C_b != 0), stored first per site; Bl = max over sites = stride of Ct / t_live
  int Bl = 0;
  double *t_live = nullptr;  // [L,Bl]
  int *nlive = nullptr;      // [L] device
  std::vector<int> nlive_host;
  // staging for host-pointer calls
  double *Q = nullptr, *pi = nullptr, *loss = nullptr, *dQ = nullptr;
  int *status = nullptr;
  // large-path workspaces
  double *Gc2 = nullptr, *gx = nullptr;  // second column buffer and 12 LD^2 + LD scratch of the first-order / hybrid sweeps
  int last_light = 0;
  // in-library all-reduce (cb_allreduce_setup)
  void *comm = nullptr;
  int (*allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  double *inv_n_global = nullptr;  // [L] 1 / n_total
  std::vector<double> n_global;
  bool expm_only = false;   // created with CB_EXPM_ONLY: no counts, no loss / training entry points
  bool per_bucket_products = false;   // created with CB_PER_BUCKET_PRODUCTS: never sum the buckets before the last product
  bool sym_counts = false;  // every live bucket has C_b == C_b^T (cherry counts are, by construction)
  int spec_sweeps = 0;  // Jacobi sweeps to enqueue before the first host check (learned from the previous solve)
  double *A = nullptr, *dsq = nullptr, *Gc = nullptr, *Vc = nullptr, *U = nullptr, *lam = nullptr,
         *sigma = nullptr, *F = nullptr, *E = nullptr, *H = nullptr, *Gt = nullptr, *T = nullptr,
         *Mt_part = nullptr, *Mt = nullptr, *X = nullptr, *loss_part = nullptr;
  // CB_F32 (large path): counts, Gt / W, T and the per-epoch operand copies in float32
  float *Ct32 = nullptr, *Gt32 = nullptr, *T32 = nullptr, *Uf = nullptr, *Utf = nullptr, *Af = nullptr, *Ff = nullptr;
  unsigned long long *off_bits = nullptr;
  unsigned long long *poll = nullptr;      // 8 words of coherent pinned host memory the first-order sweep publishes to
  unsigned long long poll_seq = 0;
  // planned (device-controlled) warm solves, eigh_planned_host.hip.h: control block, statistics partials, pinned records
  unsigned long long *ectl = nullptr, *epin = nullptr;
  double *epart = nullptr;
  bool begin_folded = false;          // this epoch's lt_build carried the planned solve's prologue (train_host.hip.h)
  double *sigma_home = nullptr;       // h->sigma outside the trainer's epochs
  double *sigma2 = nullptr;           // two words used in turn by the trainer's epochs (lt_build folds max |A_ii| into them)
  unsigned long long eseq = 0;
  bool planned_unavailable = false;   // eigh_planned_setup failed once on this handle: not tried again (host-driven solver)
  int planned_solves = 0, planned_stalls = 0;   // counters (cb_eigh_counters)
  long long record_spins = 0;         // ... and how often the host found a solve's record not yet published when it looked
  std::vector<double> epoch_seconds;  // cb_train_epoch_times: seconds from the entry of the last training call to the end of
                                      // each of its epochs on the device
  EighPlan eplan;            // the next solve's plan: part of the optimisation's state (a resumed call continues with it,
                             // so W + K epochs in two calls equal one call bit for bit)
  int k3_chunk = 0, k3_nchunks = 0;
  double *Yk = nullptr, *Lk = nullptr;   // the bucket sum without the third product (large_bank.hip.h, ky_reduce_loss):
                                         // [1 + CB_PHI_TERMS][LD][LD] each
  double phi_delta = 0.0;                // eigenvalue distance below which kphi_combine takes the series: 0.2 / max t
  unsigned int *bank_queue = nullptr;   // fused bank kernel (k123_bank): ticket queues + tile counters, 8 + 2 B words
  unsigned char *bank_args = nullptr;   // ... and its argument block (written by lg_tables in front of every launch)
  int bank_slots = 0;                   // its grid: resident workgroups of the device (4 per CU)
  int bank_claims = 0;                  // reserved first tickets per queue
  bool bank_fused = false;              // the last evaluation ran K1 -> K2 -> K3 as one launch
  int bank_kg = 1;                      // ... with four-wave (1) or eight-wave (2) tiles
  bool bank_accum = false;              // ... and summed the buckets before the last product (no K3)
  // the bank in a time basis (tbasis.hip.h): host copy of the interpolative decomposition (tb.B == 0: none), two device sets
  // (a rebuild fills the idle one while the kernels of the epoch in flight still read the other)
  CbTimeBasisHost tb;
  int tb_set = 0;
  double *tb_Ls[2] = {}, *tb_Lg[2] = {}, *tb_tf[2] = {}, *tb_tg[2] = {};
  // the NEXT basis, built by a helper thread while the optimisation runs on the current one (train_host.hip.h, tb_maintain):
  // started when sigma comes within 1.6 of the range's end, swapped in at a fixed epoch -- results do not depend on how long
  // the thread took
  std::future<CbTimeBasisHost> tb_next;
  bool tb_next_pending = false;
  int tb_next_epoch = 0;
  size_t tb_ew_lds[4] = {0, 0, 0, 0};    // dynamic-LDS limit set on this handle's device per tb_ew instantiation (cb_tb_prepare_ew)
  int tb_builds = 0, tb_stale_epochs = 0;   // cb_time_basis_info: bases built; evaluations repeated because theirs was out of range
  bool tb_failed = false;               // the grid needs more skeleton buckets than the maxima: the per-bucket forms, for good
  bool tb_block = false;                // this evaluation repeats one whose basis was out of range: per-bucket products
  bool bank_tb = false;                 // the last evaluation ran in the time basis
  int last_sweeps = 0;
  double *gn_scratch = nullptr, *gn_partial = nullptr;  // general path, allocated on first use
  int gn_nw = 0;
  // general path, S > 32 (general_large.hip.h), allocated on first use / grown with the number of squarings
  struct {
    double *Qn = nullptr, *QT = nullptr, *colsum = nullptr, *alpha = nullptr, *R = nullptr, *RT = nullptr, *E = nullptr,
           *ET = nullptr, *G = nullptr, *GT = nullptr, *Xbar = nullptr, *lpart = nullptr;
    int *nsq = nullptr;
    int cap_slots = 0;   // squaring slots E / ET can hold
  } gl;
  std::vector<double> t_host, t_live_host;   // branch lengths on the host (all buckets / live buckets first), L == 1
  bool have_prev = false;  // h->U / h->Vc hold the eigenvectors of the previous solve
  // trainer workspaces, kept between calls (hipMalloc / hipFree cost milliseconds each)
  double *ws_ptr[16] = {};
  size_t ws_cap[16] = {};
  // pinned staging for the trainers' parameter / result transfers: hipMemcpyAsync straight from
  // fresh pageable user arrays re-pins pages and was measured at ~20 ms per call
  char *pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;
  // profiling
  bool profile = false;
  int profile_every = 1;       // cb_profile(h, n > 1): the C-driven large trainer records its phase events in every n-th epoch only
  bool profile_now = false;    // ... and this epoch is one of them (everywhere else: = profile)
  hipEvent_t ev[CB_T_COUNT + 1] = {};
  bool ev_rec[CB_T_COUNT + 1] = {};
  double t_sum[CB_T_COUNT] = {};
  int t_calls = 0;
  bool t_pending = false;  // last profiled call not yet folded into t_sum
  // second event set: the C-driven trainer alternates between the two, so that folding an epoch's
  // phase times never waits for the epoch just enqueued (that wait starved the queue: ~30 us of
  // launch gaps at the start of every profiled epoch)
  hipEvent_t ev2[CB_T_COUNT + 1] = {};
  bool ev_rec2[CB_T_COUNT + 1] = {};
  bool t_pending2 = false;
};

static void swap_event_sets(cb_bank *h) {
  for (int i = 0; i <= CB_T_COUNT; ++i) {
    std::swap(h->ev[i], h->ev2[i]);
    std::swap(h->ev_rec[i], h->ev_rec2[i]);
  }
  std::swap(h->t_pending, h->t_pending2);
}

static void fold_pending(cb_bank *h);
// event i marks the END of phase i-1 .. see mark()
enum { EV_START = 0, EV_EIGH, EV_K1, EV_K2, EV_K3, EV_K4, EV_SMALL, EV_END, EV_AR };   // (EV_AR = CB_T_COUNT: the last slot)
static void mark(cb_bank *h, int which) {
  if (!h->profile || !h->profile_now) return;
  if (!h->ev[which]) (void)hipEventCreate(&h->ev[which]);
  (void)hipEventRecord(h->ev[which], h->stream);
  h->ev_rec[which] = true;
}

// The same mark WITHOUT a packet of its own between two kernels (hipEventRecord costs ~4-6 us of idle GPU there, 23 us per
// epoch for the four phase marks): the event rides on the phase's LAST kernel as its stop event (hipExtLaunchKernelGGL;
// profiles/tools/ext_event_probe: +0.0 us per launch, a START event costs 5 us).  stop_event() returns the event to pass --
// null when the call is not profiled, and the launch is then a plain one.
static hipEvent_t stop_event(cb_bank *h, int which) {
  if (!h->profile || !h->profile_now) return nullptr;
  if (!h->ev[which] && hipEventCreate(&h->ev[which]) != hipSuccess) return nullptr;
  h->ev_rec[which] = true;
  return h->ev[which];
}
static void clear_marks(cb_bank *h) {   // (an epoch that records no events must leave the pending set's flags alone)
  if (h->profile && !h->profile_now) return;
  for (bool &b : h->ev_rec) b = false;
}
#define LAUNCH_STOP(ev, kernel, grid, block, shmem, stream, ...)                                              \
  do {                                                                                                        \
    hipEvent_t ev_ = (ev);                                                                                    \
    if (ev_) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, ev_, 0, __VA_ARGS__);         \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                 \
  } while (0)

template <typename T>
static int dev_alloc(cb_bank *h, T **p, size_t count) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T) + 64);
  if (e != hipSuccess)
    return fail(CB_ENOMEM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T),
                hipGetErrorString(e));
  h->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return CB_OK;
}
#define ALLOC(ptr, count)                         \
  do {                                            \
    int rc_ = dev_alloc(h, &(ptr), (count));      \
    if (rc_ != CB_OK) return rc_;                 \
  } while (0)

// workspace slot `slot` with room for `n` doubles (grown on demand -- generously, because a
// hipFree + hipMalloc pair stalls the next call by ~13 ms -- and freed with the handle)
static bool ws_get(cb_bank *h, int slot, size_t n, double **out) {
  if (n == 0) n = 1;
  if (h->ws_cap[slot] < n) {
    size_t want = 4096;
    while (want < n) want *= 2;
    if (want * sizeof(double) <= (size_t)256 << 20) n = want;
    if (h->ws_ptr[slot]) {
      (void)hipStreamSynchronize(h->stream);
      (void)hipFree(h->ws_ptr[slot]);
      h->ws_ptr[slot] = nullptr;
      h->ws_cap[slot] = 0;
    }
    void *q = nullptr;
    if (hipMalloc(&q, n * sizeof(double) + 64) != hipSuccess) return false;
    h->ws_ptr[slot] = static_cast<double *>(q);
    h->ws_cap[slot] = n;
  }
  *out = h->ws_ptr[slot];
  return true;
}

static bool pin_reserve(cb_bank *h, size_t bytes) {
  h->pin_off = 0;
  if (h->pin_cap >= bytes) return true;
  if (h->pin) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_cap = 0;
  }
  size_t want = (size_t)1 << 20;
  while (want < bytes) want *= 2;
  void *q = nullptr;
  if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return false;
  h->pin = static_cast<char *>(q);
  h->pin_cap = want;
  return true;
}
// host -> device through the staging buffer (asynchronous; the slice stays reserved until the
// next pin_reserve)
static hipError_t h2d_staged(cb_bank *h, void *dst, const void *src, size_t bytes) {
  char *slice = h->pin + h->pin_off;
  h->pin_off += (bytes + 63) & ~(size_t)63;
  memcpy(slice, src, bytes);
  return hipMemcpyAsync(dst, slice, bytes, hipMemcpyHostToDevice, h->stream);
}
// device -> staging slice (asynchronous); *slice_out is valid after the stream is synchronised
static hipError_t d2h_staged(cb_bank *h, const void *src, size_t bytes, char **slice_out) {
  char *slice = h->pin + h->pin_off;
  h->pin_off += (bytes + 63) & ~(size_t)63;
  *slice_out = slice;
  return hipMemcpyAsync(slice, src, bytes, hipMemcpyDeviceToHost, h->stream);
}
