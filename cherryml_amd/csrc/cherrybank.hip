// libcherrybank: C ABI (include/cherrybank.h) over the gfx950 kernels.
#include "cb_internal.hip.h"

#include "common.hip.h"
#include "jacobi_block.hip.h"
#include "jacobi_wave.hip.h"
#include "large_bank.hip.h"
#include "small_bank.hip.h"
#include "train_small.hip.h"
#include "train_large.hip.h"
#include "general_small.hip.h"
#include "general_large.hip.h"
#include "eigh_planned.hip.h"

#define CB_ABI_VERSION 2

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

int cb_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#include "handle_host.hip.h"

extern "C" int cb_version(void) { return CB_ABI_VERSION; }
extern "C" const char *cb_last_error(void) { return g_err.c_str(); }
extern "C" int cb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

#include "create_host.hip.h"

extern "C" int cb_set_stream(cb_handle h, void *hip_stream, int own) {
  if (!h) return fail(CB_EINVAL, "cb_set_stream: NULL handle");
  h->stream = own ? h->own_stream : static_cast<hipStream_t>(hip_stream);
  return CB_OK;
}

extern "C" int cb_allreduce_setup(cb_handle h, void *rccl_comm, void *nccl_allreduce_fn, const double *n_total) {
  if (!h) return fail(CB_EINVAL, "cb_allreduce_setup: NULL handle");
  if (!rccl_comm) {
    h->comm = nullptr;
    h->allreduce = nullptr;
    return CB_OK;
  }
  if (!nccl_allreduce_fn || !n_total) return fail(CB_EINVAL, "cb_allreduce_setup: NULL argument");
  for (int l = 0; l < h->L; ++l)
    if (!(n_total[l] > 0.0) || !std::isfinite(n_total[l])) return fail(CB_EINVAL, "cb_allreduce_setup: n_total[%d] = %g", l, n_total[l]);
  HIP_TRY(hipSetDevice(h->dev));
  if (!h->inv_n_global) {
    int rc = dev_alloc(h, &h->inv_n_global, h->L);
    if (rc != CB_OK) return rc;
  }
  std::vector<double> inv(h->L);
  for (int l = 0; l < h->L; ++l) inv[l] = 1.0 / n_total[l];
  HIP_TRY(hipMemcpy(h->inv_n_global, inv.data(), h->L * sizeof(double), hipMemcpyHostToDevice));
  h->n_global.assign(n_total, n_total + h->L);
  h->comm = rccl_comm;
  h->allreduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(nccl_allreduce_fn);
  // the direct (log pi) term of the trainers' parameter gradient needs the job-wide count margins
  const size_t nd = (size_t)h->L * h->S;
  if (!h->dirsum_g) {
    int rc = dev_alloc(h, &h->dirsum_g, nd);
    if (rc != CB_OK) return rc;
  }
  HIP_TRY(hipMemcpyAsync(h->dirsum_g, h->dirsum, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (h->allreduce(h->dirsum_g, h->dirsum_g, nd, 8, 0, h->comm, h->stream) != 0)
    return fail(CB_EHIP, "cb_allreduce_setup: ncclAllReduce failed");
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CB_OK;
}

// sum (loss[L], dQ[L,S,S]) over the ranks, in place, on the handle's stream (ncclDouble = 8, ncclSum = 0)
static int allreduce_results(cb_bank *h, double *lossd, double *dQd) {
  if (!h->comm) return CB_OK;
  int rc = h->allreduce(lossd, lossd, (size_t)h->L, 8, 0, h->comm, h->stream);
  if (rc == 0 && dQd) rc = h->allreduce(dQd, dQd, (size_t)h->L * h->S * h->S, 8, 0, h->comm, h->stream);
  if (rc != 0) return fail(CB_EHIP, "ncclAllReduce failed with code %d", rc);
  return CB_OK;
}

extern "C" int cb_live_buckets(cb_handle h, int *nlive) {
  if (!h || !nlive) return fail(CB_EINVAL, "cb_live_buckets: NULL argument");
  memcpy(nlive, h->nlive_host.data(), h->L * sizeof(int));
  return CB_OK;
}

extern "C" int cb_total_counts(cb_handle h, double *n) {
  if (!h || !n) return fail(CB_EINVAL, "cb_total_counts: NULL argument");
  memcpy(n, h->n_host.data(), h->L * sizeof(double));
  return CB_OK;
}

// tiles per side of the 4x4-tile path, as the kernel dispatch instantiates it (S <= 20)
static int quad_ts(int S) { return S <= 4 ? 1 : S <= 8 ? 2 : S <= 16 ? 4 : S <= 20 ? 5 : 6; }

// ------------------------------------------------------------- small dispatch
template <int MODE, int NW>
static int launch_small_nw(cb_bank *h, const SmallArgs &a) {
  const size_t lds = SmallLds<NW>::TOTAL * sizeof(double);
  const int S = h->S;
#define LAUNCH(NT, KS)                                                                          \
  do {                                                                                          \
    auto kern = small_bank_kernel<NT, KS, NW, MODE>;                                            \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    hipLaunchKernelGGL(kern, dim3(h->L * (MODE == SMALL_EXPM && a.nchunk > 1 ? a.nchunk : 1)), dim3(NW * 64), lds, h->stream, a); \
  } while (0)
  if (S <= 4) LAUNCH(1, 1);
  else if (S <= 8) LAUNCH(1, 2);
  else if (S <= 16) LAUNCH(1, 4);
  else if (S <= 20) LAUNCH(2, 5);
  else if constexpr (NW == 4) {   // more than 20 states: the four-wave form only (see the kernels' launch bounds)
    if (S <= 24) LAUNCH(2, 6);
    else LAUNCH(2, 8);
  } else {
    return fail(CB_EINVAL, "internal: %d states dispatched to the eight-wave small kernels", S);
  }
#undef LAUNCH
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

template <int MODE>
static int launch_small(cb_bank *h, const SmallArgs &a) {
  // few sites: more waves per site; many sites: 4 waves (2 workgroups per CU); more than 20 states: 4 waves always
  // (one per SIMD with the whole register file, small_bank_kernel's launch bounds)
  if (h->L < 512 && h->S <= 20) return launch_small_nw<MODE, 8>(h, a);
  return launch_small_nw<MODE, 4>(h, a);
}

// --------------------------------------------------------------- large path
#include "eigh_large_host.hip.h"   // launch_sg, large_eigh
#include "eigh_planned_host.hip.h" // EighPlan, enqueue_planned_solve, eigh_planned_record

#ifndef CB_BANK_FUSED_MIN_B
#define CB_BANK_FUSED_MIN_B 64   // live buckets from which K1 -> K2 -> K3 run as ONE persistent launch
#endif
#ifndef CB_BANK_SUM_FIRST_MIN_B
#define CB_BANK_SUM_FIRST_MIN_B 24   // live buckets from which the buckets are summed before the last product (symmetric counts)
#endif
#ifndef CB_BANK_KG2_MAX_B
#define CB_BANK_KG2_MAX_B 20     // live buckets below which the tiles run on eight waves (two K-groups)
#endif
#ifndef CB_TB_MIN_B
#define CB_TB_MIN_B 28           // live buckets from which the bank runs in a time basis (tbasis.hip.h; float64 / mixed, symmetric counts):
                                 // 0.522 against 0.534 ms per epoch at 32 buckets, 0.524 against 0.563 at 40 (profiles/tools/
                                 // r5_tb_short_banks.py); the reference's real bank (43 live buckets) gains 2 % and, with a divided difference
                                 // per virtual bucket instead of the bucket sums' differences, follows the reference to 1e-14 instead of
                                 // 5e-11 over 50 epochs
#endif
#ifndef CB_TB_GROWTH
#define CB_TB_GROWTH 3.0         // a basis is built for spectra up to GROWTH x the Gershgorin bound 2 max|Q_ii| of its first matrix
#endif
// The time basis serves 2 sigma = 2 max|Q_ii| (>= the spectral radius) with a quarter of headroom left -- one optimiser step
// moves sigma by a few per cent --, and is not kept when the spectrum has shrunk to 1/64 of its range (ranks larger than needed).
static bool tb_in_range(const cb_bank *h, int B, double two_sigma, double headroom = 1.25) {
  return h->tb.B == B && two_sigma * headroom <= h->tb.rho_max && two_sigma * 64.0 >= h->tb.rho_max;
}
// (Re)build it on the host for the live buckets' grid and upload it to the idle device set (the kernels of an epoch that is
// still queued read the other one).  A grid that needs more skeleton buckets than the maxima switches the form off for good.
static double tb_growth() {
  if (const char *g = cb_test_hook("CB_TB_TEST_GROWTH")) return std::max(1.0, atof(g));   // (tests: a basis that is outgrown at once)
  return CB_TB_GROWTH;
}
static void tb_drop_next(cb_bank *h) {   // (a helper thread still building: wait for it, forget its result)
  if (h->tb_next_pending) (void)h->tb_next.get();
  h->tb_next_pending = false;
}
// upload a built basis to the idle device set and make it the current one
static int tb_install(cb_bank *h, int B, CbTimeBasisHost &&nb, double ms) {
  // (a CB_MIXED handle keeps Psi_r / P_b as doubles in its float32 T buffer: 2 (ns + nd) of its B planes)
  if (nb.B != B || !cb_tb_supported(B, h->LD, nb.ns, nb.ng) || (h->dtype == CB_MIXED && 2 * (nb.ns + nb.nd) > B) ||
      cb_tb_prepare_ew(B, nb.ns, nb.ng, h->tb_ew_lds) != 0) {   // (the elementwise kernel's LDS limit, on THIS handle's device)
    h->tb_failed = true;
    h->tb = CbTimeBasisHost{};
    return CB_OK;
  }
  const int set = h->tb_set ^ 1;
  HIP_TRY(hipMemcpy(h->tb_Ls[set], nb.Ls.data(), nb.Ls.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->tb_Lg[set], nb.Lg.data(), nb.Lg.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->tb_tf[set], nb.tf.data(), nb.tf.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->tb_tg[set], nb.tg.data(), nb.tg.size() * sizeof(double), hipMemcpyHostToDevice));
  h->tb = std::move(nb);
  h->tb_set = set;
  ++h->tb_builds;
  if (getenv("CB_DEBUG"))
    fprintf(stderr, "[cherrybank] time basis %d: rho_max %.3f, %d skeleton + %d direct forward, %d gradient buckets of %d; residuals %.1e / %.1e; %.1f ms%s\n",
            h->tb_builds, h->tb.rho_max, h->tb.ns, h->tb.nd, h->tb.ng, B, h->tb.res_s, h->tb.res_g, ms, ms < 0.0 ? " (built beside the epochs)" : "");
  return CB_OK;
}
// (Re)build it NOW on the host for the live buckets' grid.  A grid that needs more skeleton buckets than the maxima switches
// the form off for good.
static int tb_rebuild(cb_bank *h, int B, double two_sigma) {
  if (!(two_sigma > 0.0) || !std::isfinite(two_sigma)) return fail(CB_ENUMERIC, "time basis: max |Q_ii| = %g", 0.5 * two_sigma);
  tb_drop_next(h);
  const auto t0 = std::chrono::steady_clock::now();
  CbTimeBasisHost nb;
  if (!cb_tb_build(B, h->t_live_host.data(), two_sigma * tb_growth(), nb)) nb = CbTimeBasisHost{};
  // tb_install writes the device set that is NOT current; after an install earlier in the same call that is the set the kernels
  // of a queued epoch still read, and the uploads are not ordered behind h->stream (non-blocking): wait (30 ms were just spent)
  HIP_TRY(hipStreamSynchronize(h->stream));
  return tb_install(h, B, std::move(nb), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}
// After every planned epoch of the trainer (`epoch` counts the optimisation's epochs, sigma is the finished solve's): keep a
// basis in range WITHOUT stopping for the ~30 ms its construction takes.  When 2 sigma comes within a factor 1.6 of the range's
// end (or has fallen 64 times below it) a helper thread starts on the next basis, for the sigma of that epoch; it is swapped in
// CB_TB_LEAD epochs later -- or at once when 2 sigma reaches the headroom of 1.25 first (the thread is then waited for) --, i.e. at
// an epoch that depends on the sigma sequence only: the optimisation's bits do not depend on the thread's timing.
#ifndef CB_TB_LEAD
#define CB_TB_LEAD 64
#endif
static int tb_maintain(cb_bank *h, int B, int epoch, double two_sigma, bool stale) {
  if (cb_test_hook("CB_TB_TEST_GROWTH")) {   // (tests: no helper thread, no headroom -- the device finds the basis out of range)
    if (stale) return tb_rebuild(h, B, two_sigma);
    return CB_OK;
  }
  if (h->tb_failed) return CB_OK;
  if (stale || !tb_in_range(h, B, two_sigma)) {
    bool took = false;
    if (h->tb_next_pending) {   // the next one is under way (or done): take it now
      CbTimeBasisHost nb = h->tb_next.get();
      h->tb_next_pending = false;
      int rc = tb_install(h, B, std::move(nb), -1.0);
      if (rc != CB_OK || h->tb_failed) return rc;
      took = true;
    }
    // (`stale` was the device's verdict on the basis that has just been replaced: only the range of the new one counts)
    if ((stale && !took) || !tb_in_range(h, B, two_sigma)) return tb_rebuild(h, B, two_sigma);
    return CB_OK;
  }
  // (test hook CB_TB_TEST_WARN="<headroom> <lead>": e.g. "100 3" starts a helper thread on the next basis right after every swap
  // and swaps three epochs later -- the path a real optimisation takes a few times in thousands of epochs)
  double warn = 1.6;
  int lead = CB_TB_LEAD;
  if (const char *w = cb_test_hook("CB_TB_TEST_WARN")) sscanf(w, "%lf %d", &warn, &lead);
  if (h->tb_next_pending && epoch >= h->tb_next_epoch) {
    CbTimeBasisHost nb = h->tb_next.get();
    h->tb_next_pending = false;
    int rc = tb_install(h, B, std::move(nb), -1.0);
    if (rc != CB_OK || h->tb_failed) return rc;
  }
  if (!h->tb_next_pending && !tb_in_range(h, B, two_sigma, warn)) {
    std::vector<double> t(h->t_live_host.begin(), h->t_live_host.begin() + B);
    const double rho = two_sigma * tb_growth();
    h->tb_next = std::async(std::launch::async, [t, rho, B]() {
      CbTimeBasisHost nb;
      if (!cb_tb_build(B, t.data(), rho, nb)) nb = CbTimeBasisHost{};
      return nb;
    });
    h->tb_next_pending = true;
    h->tb_next_epoch = epoch + lead;
  }
  return CB_OK;
}

// h->A (padded, symmetric) and h->dsq are filled.  Output: dQ (S x S, dQ = D^1/2 dA D^-1/2) when
// `dA_padded` is false, else dL/dA itself as a padded LD x LD matrix.
// `plan`: a warm solve enqueued as a device-controlled plan (eigh_planned_host.hip.h) instead of the host-driven loop; the
// caller reads the record of solve `h->eseq` afterwards and repeats the evaluation without a plan if the solve stalled.
static int large_eval(cb_bank *h, bool normalize, double *lossd, double *out, bool dA_padded, double *Pd,
                      bool reuse_eigh = false, const EighPlan *plan = nullptr, int plan_first_slot = 0) {
  const int S = h->S, LD = h->LD;
  const int B = Pd ? h->B : h->Bl;                 // the loss visits live buckets only
  const double *tb = Pd ? h->t : h->t_live;
  const size_t LL = (size_t)LD * LD;
  double *dQd = out;
  int rc = CB_OK;
  const bool planned_now = plan && h->have_prev;
  static const int n_parts = getenv("CB_BANK_STREAMS") ? std::min(4, std::max(1, atoi(getenv("CB_BANK_STREAMS")))) : 1;
  // The bank in a TIME BASIS (tbasis.hip.h): float64 banks with symmetric counts from CB_TB_MIN_B live buckets on; CB_BANK_TB=0 / 1
  // forces it off / on (test hook), CB_PER_BUCKET_PRODUCTS at cb_create keeps every bucket's own products.  Behind a planned
  // solve the basis is the one the trainer kept in range with the previous epoch's sigma (lge_norms guards this epoch's: EC_SKIP);
  // otherwise the host reads sigma below (the host-driven solver has waited for the device several times by then).
  // A rank of a sharded job runs ITS buckets in its own basis (every N-th bucket of an ascending grid is one; the partial sums it
  // all-reduces are linear in its G_b whatever the form); sigma, hence every rebuild and repeat decision, is the same on all ranks.
  // (a test hook that pins one of the per-bucket forms -- CB_BANK_FUSED / _UNFUSED / _K3 / _KG / _TEST_NO_CLAIM -- means those forms)
  const char *tb_hook = cb_test_hook("CB_BANK_TB");
  const bool form_hooks = cb_test_hook("CB_BANK_FUSED") || cb_test_hook("CB_BANK_UNFUSED") || cb_test_hook("CB_BANK_K3") ||
                          cb_test_hook("CB_BANK_KG") || cb_test_hook("CB_BANK_TEST_NO_CLAIM");
  bool use_tb = h->sym_counts && !h->per_bucket_products && !h->tb_block && !h->tb_failed && !Pd && dQd != nullptr && h->tb_Ls[0] &&
                n_parts == 1 && (h->dtype == CB_F64 || h->dtype == CB_MIXED) &&
                (tb_hook ? atoi(tb_hook) != 0 : (!form_hooks && B >= CB_TB_MIN_B));
  if (use_tb && planned_now && h->tb.B != B) use_tb = false;
  {   // (tb_ew leaves one loss partial per half block of the upper block triangle in h->loss_part, B x tiles doubles)
    const size_t nb16 = (size_t)LD / 16, cap = (size_t)B * ((LD + LG_TM - 1) / LG_TM) * ((LD + LG_TN - 1) / LG_TN);
    if (nb16 * (nb16 + 1) > cap) use_tb = false;
  }
  // (the three bank kernels return at once when the planned solve in front of them stalled: EC_STALL)
  const unsigned long long *skipw = planned_now ? h->ectl + (use_tb ? EC_SKIP : EC_STALL) : nullptr;
  if (planned_now) {
    TbTableArgs tbt;   // (the bank's spectral tables ride on the solve's last launch)
    if (use_tb) tbt = TbTableArgs{h->tb.ns, h->tb.nd, h->tb.ng, h->tb_tf[h->tb_set], h->tb_tg[h->tb_set], h->F, h->E, h->H};
    rc = enqueue_planned_solve(h, *plan, ++h->eseq, plan_first_slot, use_tb ? h->tb.rho_max : 0.0, tbt);
  }
  else if (!(reuse_eigh && h->have_prev)) rc = large_eigh(h, true);   // reuse: same matrix as the previous call (CB_REUSE_EIGH)
  if (rc != CB_OK) return rc;
  if (use_tb && !planned_now) {
    double sg = 0.0;
    HIP_TRY(hipMemcpyAsync(&sg, h->sigma, sizeof sg, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!tb_in_range(h, B, 2.0 * sg)) {
      rc = tb_rebuild(h, B, 2.0 * sg);
      if (rc != CB_OK) return rc;
      if (h->tb_failed) use_tb = false;
    }
  }
  SlowScope scope_bank("large_eval: tables, bank, K4 enqueue");
  if (!planned_now) mark(h, EV_EIGH);   // (a planned solve's last launch carries the mark)
  const int tm = (LD + LG_TM - 1) / LG_TM, tn = (LD + LG_TN - 1) / LG_TN, tiles = tm * tn;
  const double inv_n = normalize ? 1.0 / (h->comm ? h->n_global[0] : h->n_host[0]) : 1.0;
  const int tiles_k1 = tn * (tn + 1) / 2;  // Pt is symmetric: upper-triangular tiles only
  h->bank_tb = use_tb;
  if (use_tb) {
    // tables of the virtual buckets -> K1' (Psi_r of the forward skeleton, P_b of the long-branch buckets: ns + nd products
    // into h->T) -> tb_ew (every element of every bucket: P_b, loss, G_b, the ng sums Gh_r into h->Gt) -> K2' (Th_r = Gh_r U
    // over h->T) -> K3' (W_r = (Th_r^T U) o Phi(t_r) over h->Gt) -> the sum over r + the loss -> K4: the kernels of the per-bucket
    // form on ns + nd and ng buckets instead of B.
    const CbTimeBasisHost &bas = h->tb;
    const int set = h->tb_set, nf = bas.ns + bas.nd, ng = bas.ng;
    h->bank_fused = false;
    h->bank_accum = false;
    h->bank_kg = 1;
    // (behind a planned solve the tables were written by its last launch, lge_finish: forward products 0.081 -> 0.076 ms, the
    // solve +0.002)
    if (!planned_now &&
        cb_tb_launch_tables(LD, bas.ns, bas.nd, ng, h->tb_tf[set], h->tb_tg[set], h->lam, h->F, h->E, h->H, skipw, h->stream) != 0)
      return fail(CB_EHIP, "tb_tables: launch failed");
    // CB_MIXED: everything up to G_b in float64 (the forward products, P_b, the loss), Gh_r rounded to float32 once by tb_ew,
    // the two gradient products on the float32 MFMA.  A mixed handle has no float64 Gt / T: Psi_r / P_b live (as doubles) in
    // the float32 T buffer -- dead once tb_ew has run, before K2' writes Th_r there -- and need 2 (ns + nd) <= B of its planes.
    const bool tb_mixed = h->dtype == CB_MIXED;
    double *psi_buf = tb_mixed ? reinterpret_cast<double *>(h->T32) : h->T;
    K1Args<double> k1{S, LD, nf, h->Vc, h->A, h->tb_tf[set], h->F, h->sigma, h->Ct, psi_buf, h->loss_part, inv_n, h->dsq, nullptr, skipw};
    LAUNCH_STOP(stop_event(h, EV_K1), (k1_pt_loss_gt<double, double, false, 1, true>), dim3(tiles_k1 * nf), dim3(LG4_THREADS), 0, h->stream, k1);
    const CbTbEwArgs ew{S, LD, B, bas.ns, bas.nd, ng, B - bas.nd, h->Ct, psi_buf, h->A, tb, h->tb_Ls[set], h->tb_Lg[set], h->Gt,
                        tb_mixed ? h->Gt32 : nullptr, h->loss_part, inv_n, skipw};
    int ew_parts = 0;
    // (phase marks of a time-basis evaluation: CB_T_K1 = tables + forward products, CB_T_K2 = the elementwise kernel,
    // CB_T_K3 = the two gradient products, CB_T_K4 = the sum over the virtual buckets + K4)
    if (cb_tb_launch_ew(ew, h->stream, stop_event(h, EV_K2), &ew_parts) != 0) return fail(CB_EHIP, "tb_ew: launch failed");
    const LossArgs la{h->loss_part, ew_parts, S, h->dsq, h->dirsum, inv_n, lossd, skipw};
    const dim3 red_grid((unsigned)((LL + 255) / 256) + 1);
    // (as ONE persistent launch -- k123_bank with an empty first stage, K3' tiles of a virtual bucket filling the drain of its
    // K2' tiles -- the pair took 0.227 ms against 0.146 for the two launches: ~30 buckets are a short bank, EXPERIMENTS section 13;
    // eight-wave tiles for the last product -- 450 tiles, fewer than two per CU -- measured: 0.1298 against 0.1306 ms, no gain)
    if (tb_mixed) {
      hipLaunchKernelGGL(lg_cast_f32, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream, LL, (size_t)0, h->U, h->Vc, h->A, h->F,
                         h->Uf, h->Utf, h->Af, h->Ff);
      K2Args<float> k2{LD, h->Gt32, h->Uf, h->T32, skipw};
      K3Args<float> k3{LD, ng, h->T32, h->Uf, h->tb_tg[set], h->lam, h->E, h->H, h->Gt32, 1, skipw};
      hipLaunchKernelGGL((k2_t_eq_g_u<float, 1>), dim3(tiles * ng), dim3(LG4_THREADS), 0, h->stream, k2);
      LAUNCH_STOP(stop_event(h, EV_K3), (k3_w_phi<float, 1>), dim3(tiles_k1 * ng), dim3(LG4_THREADS), 0, h->stream, k3);
      hipLaunchKernelGGL(k3_reduce_loss<float>, red_grid, dim3(256), 0, h->stream, h->Gt32, ng, LL, h->Mt, LD, la);
    } else {
      K2Args<double> k2{LD, h->Gt, h->U, h->T, skipw};
      K3Args<double> k3{LD, ng, h->T, h->U, h->tb_tg[set], h->lam, h->E, h->H, h->Gt, 1, skipw};
      hipLaunchKernelGGL((k2_t_eq_g_u<double, 1>), dim3(tiles * ng), dim3(LG4_THREADS), 0, h->stream, k2);
      LAUNCH_STOP(stop_event(h, EV_K3), (k3_w_phi<double, 1>), dim3(tiles_k1 * ng), dim3(LG4_THREADS), 0, h->stream, k3);
      hipLaunchKernelGGL(k3_reduce_loss<double>, red_grid, dim3(256), 0, h->stream, h->Gt, ng, LL, h->Mt, LD, la);
    }
    K4Args k4a{S, LD, h->Mt, h->Vc, h->X, nullptr, nullptr, nullptr};
    k4a.skip = skipw;
    launch_sg(h, k4a, 0);
    K4Args k4b{S, LD, h->Vc, h->X, dQd, dA_padded ? nullptr : h->dsq, nullptr, nullptr};
    k4b.skip = skipw;
    launch_sg(h, k4b, 0, 0.0, 0.0, nullptr, stop_event(h, EV_K4));
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  // Symmetric counts: the buckets are summed BEFORE the last product (large_bank.hip.h, ky_reduce_loss / kphi_combine: one
  // streaming pass over T and 1 + CB_PHI_TERMS single products instead of a third product per bucket); CB_BANK_K3=1 keeps
  // the per-bucket third product (test hook: the reference point of the accuracy test).  Other banks: K3 on all tiles.
  // float64 banks only (with T_b in float32 the difference Le - Le^T amplifies its 1e-7 to 7e-3 in dL/dQ on the demo bank), and
  // from CB_BANK_SUM_FIRST_MIN_B live buckets on: the tail -- one pass over T, seven single products, the combination: ~45 us --
  // is dearer than the third product of a short bank (17 buckets: 0.160 against 0.153 ms); CB_BANK_K3=0 forces it on.
  const char *k3_hook = cb_test_hook("CB_BANK_K3");
  const bool accum = h->sym_counts && !h->per_bucket_products && !Pd && dQd != nullptr && h->Yk && n_parts == 1 && h->dtype == CB_F64 &&
                     (k3_hook ? atoi(k3_hook) == 0 : (B >= CB_BANK_SUM_FIRST_MIN_B && h->phi_delta >= 4e-3));
  // (phi_delta = 0.2 / max t: a grid that reaches branch lengths beyond 50 -- the reference's ends at 13.7 -- would push the
  // quotient (Le_ij - Le_ji) / dlam down to eigenvalue distances where its cancellation costs more than six digits)
  const int tiles_k3 = accum ? 0 : h->sym_counts ? tiles_k1 : tiles;
  // M from the bucket sums: Y_k = sum_b T_b diag(c_bk) (+ the loss), Lt_k = Y_k^T U (one launch), M = combine(Lt, lam)
  auto accumulated_tail = [&](bool narrow, hipEvent_t stop) {
    const LossArgs la{h->loss_part, B * tiles_k1, S, h->dsq, h->dirsum, inv_n, lossd, skipw};
    const dim3 red_grid((unsigned)((LL + 255) / 256) + 1);   // (+ the workgroup that sums the loss partials)
    const YArgs ya{LD, B, tb, h->E, h->Yk};
    if (narrow) hipLaunchKernelGGL(ky_reduce_loss<float>, red_grid, dim3(256), 0, h->stream, h->T32, ya, la);
    else hipLaunchKernelGGL(ky_reduce_loss<double>, red_grid, dim3(256), 0, h->stream, h->T, ya, la);
    K4Args gl{S, LD, h->Yk, h->U, h->Lk, nullptr, nullptr, nullptr};
    gl.skip = skipw;
    gl.ystride = LL;
    launch_sg(h, gl, 0, 0.0, 0.0, nullptr, nullptr, 1 + CB_PHI_TERMS);
    double delta = h->phi_delta;
    if (const char *z = cb_test_hook("CB_PHI_Z")) delta *= atof(z) / 0.1;   // (experiment: the series' range |z| <= CB_PHI_Z)
    LAUNCH_STOP(stop, kphi_combine<CB_PHI_TERMS>, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream, LD, h->Lk, h->lam, delta,
                h->Mt, skipw);
  };
  // K1 -> K2 -> K3 as ONE persistent launch (k123_bank, large_bank.hip.h) whenever the gradient is wanted; CB_BANK_UNFUSED=1
  // keeps the three launches (per-kernel profiles, and the reference point of tests/test_gpu_s400_full.py)
  // (test hooks, read per call: the tests switch them inside one process)
  const bool f32 = h->dtype == CB_F32 && !Pd, mixed = h->dtype == CB_MIXED && !Pd;
  // Which form the three bank products take is decided by the SHAPE (live buckets; profiles/tools/r5_bank_sweep.py, round 5):
  // the persistent launch k123_bank pays when the bank fills the chip several times over (its gain is the two drains it
  // hides); a short bank -- the reference's real bank has 43 live buckets, one rank's share of an 8-rank job 17 -- is a chain
  // of three tile latencies whatever is launched, and the persistent launch's 1024 resident workgroups (most of them waiting)
  // make its tiles slower: 0.40 against 0.29 ms at 43 buckets, 0.24 against 0.14 at 17.  CB_BANK_UNFUSED=1 / CB_BANK_FUSED=1
  // force a form (test hooks).
  const bool unfused_env = cb_test_hook("CB_BANK_UNFUSED") != nullptr, fused_env = cb_test_hook("CB_BANK_FUSED") != nullptr;
  const bool fused = !Pd && dQd && !unfused_env && n_parts == 1 && h->bank_queue && (fused_env || B >= CB_BANK_FUSED_MIN_B);
  // tile form of K1 .. K3 (large_bank.hip.h, lg4_gemm_tile): eight waves per 80 x 80 tile (two K-groups, two workgroups per
  // CU) or four (four workgroups per CU); CB_BANK_KG=1 / 2 forces one (test hook).  cb_expm_bank keeps
  // the four-wave form.
  // Below ~20 buckets every stage has at most two tiles per CU and the eight-wave tile's shorter latency wins (0.141 against
  // 0.152 ms at 17 buckets); above, four independent four-wave workgroups per CU overlap better than two eight-wave ones whose
  // K-groups share their barriers (0.75 against 0.65 ms at 129).  The float32 bank keeps the four-wave form: its P_b entries
  // of O(t^2) are sums with cancellation whose SIGN in float32 depends on the summation order (DESIGN / EXPERIMENTS, round 5).
  int kg = (B < CB_BANK_KG2_MAX_B && !f32) ? 2 : 1;
  if (const char *k = cb_test_hook("CB_BANK_KG")) kg = atoi(k) == 1 ? 1 : 2;
  if (Pd) kg = 1;
  h->bank_kg = kg;
  h->bank_accum = accum;
  // (queues and argument block: allocated with the handle, create_host.hip.h)
  h->bank_fused = fused;
  const dim3 tables_grid((unsigned)(((size_t)B * LD + 255) / 256));
  if (!fused)
    hipLaunchKernelGGL(lg_tables<NoBankArgs>, tables_grid, dim3(256), 0, h->stream, LD, B, tb, h->lam, h->sigma, h->F, h->E, h->H,
                       NoBankArgs{}, (NoBankArgs *)nullptr, 0);
  // float32 bank (cb_create(dtype = CB_F32)): the loss / gradient products run on the f32 MFMA from f32
  // copies of this epoch's U, U^T, A and F; cb_expm_bank (Pd) always takes the float64 kernels
  // CB_MIXED: P_b, the loss and G_b in float64 (the O(t^2) entries of P_b keep their relative accuracy),
  // G_b rounded to float32 once, the two contractions on the float32 MFMA
  if ((f32 || mixed) && !fused)
    hipLaunchKernelGGL(lg_cast_f32, dim3((unsigned)((std::max(LL, (size_t)B * LD) + 255) / 256)), dim3(256), 0, h->stream, LL,
                       (size_t)B * LD, h->U, h->Vc, h->A, h->F, h->Uf, h->Utf, h->Af, h->Ff);
  // (no event between the spectral tables and K1: a hipEventRecord costs ~6 us of idle GPU between two kernels; CB_T_K1
  // is the span from the end of the eigensolver to the end of K1 = lg_tables (4 us) [+ the float32 casts] + K1)
  if (fused) {
    // lg_tables (+ the argument block and the zeroed queues), then one launch of 4 workgroups per CU (fewer when the bank is
    // small); the loss partials are summed after it
    const int total = B * (tiles_k1 + tiles + tiles_k3), grid = std::min(h->bank_slots / kg, total);
    const int sym = h->sym_counts ? 1 : 0;
    const int test_no_claim = cb_test_hook("CB_BANK_TEST_NO_CLAIM") ? 1 : 0;   // (tests/test_gpu_s400_full.py: the help path)
    // (the phase marks ride on the launches as stop events: handle_host.hip.h, stop_event())
    if (h->profile && h->profile_now) h->ev_rec[EV_K1] = h->ev_rec[EV_K2] = false;   // CB_T_K1 = the whole launch (+ tables), see read_phase_times
    const hipEvent_t bank_stop = stop_event(h, EV_K3);
    auto launch = [&](auto args) {
      typedef decltype(args) A;
      A *dst = reinterpret_cast<A *>(h->bank_args);
      hipLaunchKernelGGL(lg_tables<A>, tables_grid, dim3(256), 0, h->stream, LD, B, tb, h->lam, h->sigma, h->F, h->E, h->H, args, dst, grid);
      return dst;
    };
    static_assert(sizeof(K123Args<float, float>) == sizeof(K123Args<double, double>) &&
                  sizeof(K123Args<double, float>) == sizeof(K123Args<double, double>), "one argument block for all");
    if (f32) {
      // (the float32 casts read F: the tables first, then the casts, then the bank -- as in the separate launches)
      K123Args<float, float> a{{S, LD, B, h->Utf, h->Af, tb, h->Ff, h->sigma, h->Ct32, h->Gt32, h->loss_part, inv_n, h->dsq, nullptr, skipw},
                               {LD, h->Gt32, h->Uf, h->T32, skipw},
                               {LD, B, h->T32, h->Uf, tb, h->lam, h->E, h->H, h->Gt32, sym, skipw},
                               {h->bank_queue, B, tiles_k1, tiles, tiles_k3, h->bank_claims, test_no_claim}};
      auto *dst = launch(a);
      hipLaunchKernelGGL(lg_cast_f32, dim3((unsigned)((std::max(LL, (size_t)B * LD) + 255) / 256)), dim3(256), 0, h->stream, LL,
                         (size_t)B * LD, h->U, h->Vc, h->A, h->F, h->Uf, h->Utf, h->Af, h->Ff);
      if (cb_launch_bank_fused(1, kg, dst, grid, h->stream, bank_stop) != 0) return fail(CB_EHIP, "k123_bank: launch failed");
    } else if (mixed) {
      K123Args<double, float> a{{S, LD, B, h->Vc, h->A, tb, h->F, h->sigma, h->Ct, h->Gt32, h->loss_part, inv_n, h->dsq, nullptr, skipw},
                                {LD, h->Gt32, h->Uf, h->T32, skipw},
                                {LD, B, h->T32, h->Uf, tb, h->lam, h->E, h->H, h->Gt32, sym, skipw},
                                {h->bank_queue, B, tiles_k1, tiles, tiles_k3, h->bank_claims, test_no_claim}};
      auto *dst = launch(a);
      hipLaunchKernelGGL(lg_cast_f32, dim3((unsigned)((std::max(LL, (size_t)B * LD) + 255) / 256)), dim3(256), 0, h->stream, LL,
                         (size_t)B * LD, h->U, h->Vc, h->A, h->F, h->Uf, h->Utf, h->Af, h->Ff);
      if (cb_launch_bank_fused(2, kg, dst, grid, h->stream, bank_stop) != 0) return fail(CB_EHIP, "k123_bank: launch failed");
    } else {
      K123Args<double, double> a{{S, LD, B, h->Vc, h->A, tb, h->F, h->sigma, h->Ct, h->Gt, h->loss_part, inv_n, h->dsq, nullptr, skipw},
                                 {LD, h->Gt, h->U, h->T, skipw},
                                 {LD, B, h->T, h->U, tb, h->lam, h->E, h->H, h->Gt, sym, skipw},
                                 {h->bank_queue, B, tiles_k1, tiles, tiles_k3, h->bank_claims, test_no_claim}};
      auto *dst = launch(a);
      if (cb_launch_bank_fused(0, kg, dst, grid, h->stream, bank_stop) != 0) return fail(CB_EHIP, "k123_bank: launch failed");
    }
    const LossArgs la{h->loss_part, B * tiles_k1, S, h->dsq, h->dirsum, inv_n, lossd, skipw};
    const dim3 red_grid((unsigned)((LL + 255) / 256) + 1);   // (+ the workgroup that sums the loss partials)
    if (accum)
      accumulated_tail(f32 || mixed, nullptr);
    else if (f32 || mixed)
      hipLaunchKernelGGL(k3_reduce_loss<float>, red_grid, dim3(256), 0, h->stream, h->Gt32, B, LL, h->Mt, h->sym_counts ? LD : 0, la);
    else
      hipLaunchKernelGGL(k3_reduce_loss<double>, red_grid, dim3(256), 0, h->stream, h->Gt, B, LL, h->Mt, h->sym_counts ? LD : 0, la);
    // (a stalled planned solve: the reduction and K4 return at once too -- nothing runs on stale Gt / T, and h->loss / h->Mt
    // are only written by the evaluation that follows a finished solve)
    K4Args k4a{S, LD, h->Mt, h->Vc, h->X, nullptr, nullptr, nullptr};
    k4a.skip = skipw;
    launch_sg(h, k4a, 0);
    K4Args k4b{S, LD, h->Vc, h->X, dQd, dA_padded ? nullptr : h->dsq, nullptr, nullptr};
    k4b.skip = skipw;
    launch_sg(h, k4b, 0, 0.0, 0.0, nullptr, stop_event(h, EV_K4));
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  // (the separate launches in the tile form `kg`: same bits as the fused launch of that form)
  // (the phase marks ride on the kernels as stop events -- a hipEventRecord between two kernels costs ~5 us of idle GPU)
#define LAUNCH_KG_ON(st_, ev_, kern1, kern2, grid_, args_)                                                     \
  do {                                                                                                         \
    if (kg == 2) LAUNCH_STOP(ev_, kern2, grid_, dim3(2 * LG4_THREADS), 0, st_, args_);                         \
    else LAUNCH_STOP(ev_, kern1, grid_, dim3(LG4_THREADS), 0, st_, args_);                                     \
  } while (0)
#define LAUNCH_KG(ev_, kern1, kern2, grid_, args_) LAUNCH_KG_ON(h->stream, ev_, kern1, kern2, grid_, args_)
  const hipEvent_t ev_k1 = stop_event(h, EV_K1);   // (null when the call is not profiled)
  if (f32) {
    K1Args<float> k1{S, LD, B, h->Utf, h->Af, tb, h->Ff, h->sigma, h->Ct32, h->Gt32, h->loss_part, inv_n, h->dsq, nullptr, skipw};
    LAUNCH_KG(ev_k1, (k1_pt_loss_gt<float, float, false, 1>), (k1_pt_loss_gt<float, float, false, 2>), dim3(tiles_k1 * B), k1);
  } else if (mixed) {
    K1Args<double, float> k1{S, LD, B, h->Vc, h->A, tb, h->F, h->sigma, h->Ct, h->Gt32, h->loss_part, inv_n, h->dsq, nullptr, skipw};
    LAUNCH_KG(ev_k1, (k1_pt_loss_gt<double, float, false, 1>), (k1_pt_loss_gt<double, float, false, 2>), dim3(tiles_k1 * B), k1);
  } else if (n_parts > 1 && !Pd && dQd && B >= 4 * n_parts && !h->comm && !h->profile) {
    // OPT-IN (CB_BANK_STREAMS=n, float64 bank, no profile markers): the buckets in n equal parts on n queues,
    // K1 -> K2 -> K3 each, so that the drain of one part's kernel overlaps the other parts' kernels; same results bit
    // for bit (disjoint buckets, the bucket sum is taken after the join).  n = 2: 1.110 -> 1.072 ms per epoch on the
    // bench bank; 3 and 4 are slower again.  Not the default: with kernels of two queues sharing the chip the
    // per-kernel durations the bench reports (HIP events on one stream, rocprof averages) stop meaning anything.
    if (!h->ev_fork) HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int p = 1; p < n_parts; ++p)
      if (!h->xstream[p - 1]) {
        HIP_TRY(hipStreamCreateWithFlags(&h->xstream[p - 1], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join[p - 1], hipEventDisableTiming));
      }
    HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    for (int p = 1; p < n_parts; ++p) HIP_TRY(hipStreamWaitEvent(h->xstream[p - 1], h->ev_fork, 0));
    auto part_of = [&](int p, int &b0, int &Bs, hipStream_t &st) {
      b0 = (int)((long)B * p / n_parts);
      Bs = (int)((long)B * (p + 1) / n_parts) - b0;
      st = p ? h->xstream[p - 1] : h->stream;
    };
    for (int p = 0; p < n_parts; ++p) {
      int b0, Bs; hipStream_t st;
      part_of(p, b0, Bs, st);
      K1Args<double> k1{S, LD, Bs, h->Vc, h->A, tb + b0, h->F + (size_t)b0 * LD, h->sigma, h->Ct + b0 * LL, h->Gt + b0 * LL,
                        h->loss_part + (size_t)b0 * tiles_k1, inv_n, h->dsq, nullptr};
      LAUNCH_KG_ON(st, nullptr, (k1_pt_loss_gt<double, double, false, 1>), (k1_pt_loss_gt<double, double, false, 2>), dim3(tiles_k1 * Bs), k1);
    }
    for (int p = 0; p < n_parts; ++p) {
      int b0, Bs; hipStream_t st;
      part_of(p, b0, Bs, st);
      K2Args<double> k2{LD, h->Gt + b0 * LL, h->U, h->T + b0 * LL};
      LAUNCH_KG_ON(st, nullptr, (k2_t_eq_g_u<double, 1>), (k2_t_eq_g_u<double, 2>), dim3(tiles * Bs), k2);
    }
    for (int p = 0; p < n_parts; ++p) {
      int b0, Bs; hipStream_t st;
      part_of(p, b0, Bs, st);
      K3Args<double> k3{LD, Bs, h->T + b0 * LL, h->U, tb + b0, h->lam, h->E + (size_t)b0 * LD, h->H + (size_t)b0 * LD,
                        h->Gt + b0 * LL, h->sym_counts ? 1 : 0};
      LAUNCH_KG_ON(st, nullptr, (k3_w_phi<double, 1>), (k3_w_phi<double, 2>), dim3(tiles_k3 * Bs), k3);
    }
    for (int p = 1; p < n_parts; ++p) {
      HIP_TRY(hipEventRecord(h->ev_join[p - 1], h->xstream[p - 1]));
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join[p - 1], 0));
    }
    hipLaunchKernelGGL(lg_finish_loss, dim3(1), dim3(256), 0, h->stream, h->loss_part, B * tiles_k1, S,
                       h->dsq, h->dirsum, inv_n, lossd);
    hipLaunchKernelGGL(k3_reduce<double>, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream,
                       h->Gt, B, LL, h->Mt, h->sym_counts ? LD : 0);
    K4Args k4a{S, LD, h->Mt, h->Vc, h->X, nullptr, nullptr, nullptr};
    launch_sg(h, k4a, 0);
    K4Args k4b{S, LD, h->Vc, h->X, dQd, dA_padded ? nullptr : h->dsq, nullptr, nullptr};
    launch_sg(h, k4b, 0);
    HIP_TRY(hipGetLastError());
    return CB_OK;
  } else {
    K1Args<double> k1{S, LD, B, h->Vc, h->A, tb, h->F, h->sigma, h->Ct, h->Gt, h->loss_part, inv_n, h->dsq, Pd, skipw};
    if (Pd) LAUNCH_STOP(ev_k1, (k1_pt_loss_gt<double, double, true>), dim3(tiles_k1 * B), dim3(LG4_THREADS), 0, h->stream, k1);
    else LAUNCH_KG(ev_k1, (k1_pt_loss_gt<double, double, false, 1>), (k1_pt_loss_gt<double, double, false, 2>), dim3(tiles_k1 * B), k1);
  }
  if (Pd) {
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  if (!dQd) {   // the loss alone
    hipLaunchKernelGGL(lg_finish_loss, dim3(1), dim3(256), 0, h->stream, h->loss_part, B * tiles_k1, S,
                       h->dsq, h->dirsum, inv_n, lossd);
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  // K1 -> K2 -> K3 back to back (the loss partials are summed beside the bucket sum, behind K3: a one-workgroup launch between
  // K1 and K2 would sit on the chain), then the bucket sum + loss, then K4 -- the sequence of the fused path as three launches
  const LossArgs la{h->loss_part, B * tiles_k1, S, h->dsq, h->dirsum, inv_n, lossd, skipw};
  const dim3 red_grid((unsigned)((LL + 255) / 256) + 1);   // (+ the workgroup that sums the loss partials)
  if (f32 || mixed) {
    K2Args<float> k2{LD, h->Gt32, h->Uf, h->T32, skipw};
    LAUNCH_KG(stop_event(h, EV_K2), (k2_t_eq_g_u<float, 1>), (k2_t_eq_g_u<float, 2>), dim3(tiles * B), k2);
    if (accum) {
      accumulated_tail(true, stop_event(h, EV_K3));   // (CB_T_K3: the bucket sums, their products and the combination)
    } else {
      K3Args<float> k3{LD, B, h->T32, h->Uf, tb, h->lam, h->E, h->H, h->Gt32, h->sym_counts ? 1 : 0, skipw};
      LAUNCH_KG(stop_event(h, EV_K3), (k3_w_phi<float, 1>), (k3_w_phi<float, 2>), dim3(tiles_k3 * B), k3);
      hipLaunchKernelGGL(k3_reduce_loss<float>, red_grid, dim3(256), 0, h->stream, h->Gt32, B, LL, h->Mt, h->sym_counts ? LD : 0, la);
    }
  } else {
    K2Args<double> k2{LD, h->Gt, h->U, h->T, skipw};
    LAUNCH_KG(stop_event(h, EV_K2), (k2_t_eq_g_u<double, 1>), (k2_t_eq_g_u<double, 2>), dim3(tiles * B), k2);
    if (accum) {
      accumulated_tail(false, stop_event(h, EV_K3));
    } else {
      K3Args<double> k3{LD, B, h->T, h->U, tb, h->lam, h->E, h->H, h->Gt, h->sym_counts ? 1 : 0, skipw};
      LAUNCH_KG(stop_event(h, EV_K3), (k3_w_phi<double, 1>), (k3_w_phi<double, 2>), dim3(tiles_k3 * B), k3);
      hipLaunchKernelGGL(k3_reduce_loss<double>, red_grid, dim3(256), 0, h->stream, h->Gt, B, LL, h->Mt, h->sym_counts ? LD : 0, la);
    }
  }
  {
    K4Args k4a{S, LD, h->Mt, h->Vc, h->X, nullptr, nullptr, nullptr};
    k4a.skip = skipw;
    launch_sg(h, k4a, 0);
    K4Args k4b{S, LD, h->Vc, h->X, dQd, dA_padded ? nullptr : h->dsq, nullptr, nullptr};
    k4b.skip = skipw;
    launch_sg(h, k4b, 0, 0.0, 0.0, nullptr, stop_event(h, EV_K4));
  }
#undef LAUNCH_KG
#undef LAUNCH_KG_ON
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

static int large_loss_grad(cb_bank *h, const double *Qd, const double *pid, bool normalize,
                           double *lossd, double *dQd, double *Pd, bool reuse_eigh = false) {
  const size_t LL = (size_t)h->LD * h->LD;
  if (!(reuse_eigh && h->have_prev))
    hipLaunchKernelGGL(lg_build_A, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream, h->S, h->LD,
                       Qd, pid, h->A, h->dsq);
  return large_eval(h, normalize, lossd, dQd, false, Pd, reuse_eigh);
}

// ---------------------------------------------------------------- entry points
static int general_run(cb_bank *h, const double *Qd, int flags, double *lossd, double *dQd,
                       double *Pd);
static int finish_call(cb_bank *h, int flags) {
  if ((flags & CB_PTR_DEVICE) && (flags & CB_NO_SYNC)) return CB_OK;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CB_OK;
}

extern "C" int cb_loss_grad(cb_handle h, const double *Q, const double *pi, int flags, double *loss,
                            double *dQ) {
  if (!h || !Q || !pi || !loss) return fail(CB_EINVAL, "cb_loss_grad: NULL argument");
  if (h->expm_only) return fail(CB_EINVAL, "cb_loss_grad: the handle was created with CB_EXPM_ONLY (no counts)");
  HIP_TRY(hipSetDevice(h->dev));
  const size_t SS = (size_t)h->S * h->S;
  const bool devp = flags & CB_PTR_DEVICE;
  const double *Qd = Q, *pid = pi;
  double *lossd = loss, *dQd = dQ;
  if (!devp) {
    HIP_TRY(hipMemcpyAsync(h->Q, Q, h->L * SS * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->pi, pi, (size_t)h->L * h->S * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
    Qd = h->Q;
    pid = h->pi;
    lossd = h->loss;
    dQd = dQ ? h->dQ : nullptr;
  }
  int rc;
  if (h->profile) fold_pending(h);
  for (bool &b : h->ev_rec) b = false;
  mark(h, EV_START);
  if (h->large) {
    rc = large_loss_grad(h, Qd, pid, flags & CB_NORMALIZE, lossd, dQd, nullptr);
  } else {
    SmallArgs a{};
    a.S = h->S;
    a.L = h->L;
    a.B = h->Bl;
    a.nlive = h->nlive;
    a.t = h->t_live;
    a.Ct = h->Ct;
    a.Cq = h->Cq;
    a.nq = h->nq;
    a.inv_n = (flags & CB_NORMALIZE) ? (h->comm ? h->inv_n_global : h->inv_n) : h->ones;
    a.dirsum = h->dirsum;
    a.Q = Qd;
    a.pi = pid;
    a.loss = lossd;
    a.dQ = dQd;
    a.status = h->status;
    rc = launch_small<SMALL_LOSSGRAD>(h, a);
    mark(h, EV_SMALL);
  }
  if (rc == CB_OK && cb_test_hook("CB_FAULT_INJECT") && atoi(cb_test_hook("CB_FAULT_INJECT")) == -1)
    rc = fail(CB_ENUMERIC, "injected fault (CB_FAULT_INJECT)");
  if (rc != CB_OK && h->comm) {   // keep this rank's place in the collective (NaN payload): the peers get NaN, not a hang
    const std::string first_error = g_err;
    (void)hipMemsetAsync(lossd, 0xFF, h->L * sizeof(double), h->stream);
    if (dQd) (void)hipMemsetAsync(dQd, 0xFF, h->L * SS * sizeof(double), h->stream);
    (void)allreduce_results(h, lossd, dQd);
    (void)hipStreamSynchronize(h->stream);
    g_err = first_error;
    return rc;
  }
  if (rc != CB_OK) return rc;
  if ((rc = allreduce_results(h, lossd, dQd)) != CB_OK) return rc;
  if (h->profile) h->t_pending = true;
  if (!devp) {
    HIP_TRY(hipMemcpyAsync(loss, h->loss, h->L * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (dQ)
      HIP_TRY(hipMemcpyAsync(dQ, h->dQ, h->L * SS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  return finish_call(h, flags);
}

// the public entry honours documented flag bits only: CB_REUSE_EIGH is internal (a caller passing that bit with a
// different Q would silently get the previous matrix's bank)
extern "C" int cb_expm_bank(cb_handle h, const double *Q, const double *pi, int flags, double *P) {
  return cb_internal_expm_bank(h, Q, pi, flags & ~CB_REUSE_EIGH, P);
}

int cb_internal_expm_bank(cb_handle h, const double *Q, const double *pi, int flags, double *P) {
  if (!h || !Q || !P) return fail(CB_EINVAL, "cb_expm_bank: NULL argument");
  HIP_TRY(hipSetDevice(h->dev));
  const size_t SS = (size_t)h->S * h->S, nP = (size_t)h->L * h->B * SS;
  const bool devp = flags & CB_PTR_DEVICE;
  const double *Qd = Q, *pid = pi;
  double *Pd = P;
  double *Ptmp = nullptr;
  if (!devp) {
    HIP_TRY(hipMemcpyAsync(h->Q, Q, h->L * SS * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (pi)
      HIP_TRY(hipMemcpyAsync(h->pi, pi, (size_t)h->L * h->S * sizeof(double), hipMemcpyHostToDevice,
                             h->stream));
    HIP_TRY(hipMalloc((void **)&Ptmp, nP * sizeof(double)));
    Qd = h->Q;
    pid = h->pi;
    Pd = Ptmp;
  }
  int rc;
  if (!pi) {
    rc = general_run(h, Qd, 0, h->loss, nullptr, Pd);
  } else if (h->large) {
    rc = large_loss_grad(h, Qd, pid, false, h->loss, nullptr, Pd, (flags & CB_REUSE_EIGH) != 0);
  } else {
    SmallArgs a{};
    a.S = h->S;
    a.L = h->L;
    a.B = h->B;
    a.t = h->t;
    a.Ct = h->Ct;
    a.inv_n = h->ones;
    a.dirsum = h->dirsum;
    a.Q = Qd;
    a.pi = pid;
    a.loss = h->loss;
    a.P = Pd;
    a.status = h->status;
    // workgroups per site (SmallArgs::nchunk): ~32 buckets each while the launch stays below ~1024 workgroups
    a.nchunk = std::max(1, std::min((h->B + 31) / 32, std::max(1, 1024 / h->L)));
    rc = launch_small<SMALL_EXPM>(h, a);
  }
  if (rc == CB_OK && !devp) {
    hipError_t e = hipMemcpyAsync(P, Ptmp, nP * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail(CB_EHIP, "cb_expm_bank: copy back failed: %s", hipGetErrorString(e));
  }
  if (Ptmp) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(Ptmp);
  }
  if (rc != CB_OK) return rc;
  return finish_call(h, flags);
}

// internal (cb_internal.hip.h): new branch lengths for a counts-free handle, B <= the B it was created with --
// lets cb_tree_likelihood_batch push family after family through ONE handle (one eigendecomposition)
int cb_internal_spectral(cb_handle h, CbSpectral *out) {
  if (!h || !out) return fail(CB_EINVAL, "cb_internal_spectral: NULL argument");
  if (!h->large || !h->have_prev) return fail(CB_EINVAL, "cb_internal_spectral: no decomposition on this handle");
  out->S = h->S; out->LD = h->LD; out->A = h->A; out->U = h->U; out->lam = h->lam; out->dsq = h->dsq; out->sigma = h->sigma;
  return CB_OK;
}

int cb_internal_set_times(cb_handle h, const double *t_host, int B, const double *t_dev) {
  if (!h || !t_host) return fail(CB_EINVAL, "cb_internal_set_times: NULL argument");
  if (!h->expm_only || h->L != 1) return fail(CB_EINVAL, "cb_internal_set_times: counts-free single-bank handles only");
  if (B < 1 || B > h->B_cap) return fail(CB_EINVAL, "cb_internal_set_times: B = %d exceeds the handle's capacity %d", B, h->B_cap);
  HIP_TRY(hipSetDevice(h->dev));
  if (t_dev) {   // already resident: ordered behind the previous bank on the handle's stream, the host does not wait
    HIP_TRY(hipMemcpyAsync(h->t, t_dev, B * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else {
    HIP_TRY(hipStreamSynchronize(h->stream));     // the previous bank is done with h->t
    HIP_TRY(hipMemcpy(h->t, t_host, B * sizeof(double), hipMemcpyHostToDevice));
  }
  h->B = B;
  h->t_host.assign(t_host, t_host + B);
  return CB_OK;
}

extern "C" int cb_eigh(cb_handle h, const double *A, int flags, double *lam, double *U) {
  if (!h || !A || !lam || !U) return fail(CB_EINVAL, "cb_eigh: NULL argument");
  if (flags & CB_PTR_DEVICE) return fail(CB_EUNSUPPORTED, "cb_eigh: host pointers only (debug entry)");
  HIP_TRY(hipSetDevice(h->dev));
  const int S = h->S;
  const size_t SS = (size_t)S * S;
  if (!h->large) {
    double *lamd = nullptr, *Ud = nullptr;
    HIP_TRY(hipMalloc((void **)&lamd, (size_t)h->L * S * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&Ud, h->L * SS * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(h->Q, A, h->L * SS * sizeof(double), hipMemcpyHostToDevice, h->stream));
    SmallArgs a{};
    a.S = S;
    a.L = h->L;
    a.B = h->B;
    a.Q = h->Q;
    a.lam_out = lamd;
    a.U_out = Ud;
    a.status = h->status;
    int rc = launch_small<SMALL_EIGH>(h, a);
    if (rc == CB_OK) {
      hipError_t e = hipMemcpyAsync(lam, lamd, (size_t)h->L * S * sizeof(double), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(U, Ud, h->L * SS * sizeof(double), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) rc = fail(CB_EHIP, "cb_eigh: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(lamd);
    (void)hipFree(Ud);
    return rc;
  }
  // large: pad A into h->A
  const int LD = h->LD;
  std::vector<double> Ap((size_t)LD * LD, 0.0);
  for (int i = 0; i < S; ++i) memcpy(&Ap[(size_t)i * LD], A + (size_t)i * S, S * sizeof(double));
  HIP_TRY(hipMemcpyAsync(h->A, Ap.data(), Ap.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = large_eigh(h, false);
  if (rc != CB_OK) return rc;
  std::vector<double> Up((size_t)LD * LD), lp(LD);
  HIP_TRY(hipMemcpyAsync(Up.data(), h->U, Up.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(lp.data(), h->lam, LD * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  // padded eigenpairs are unit vectors on pad indices: drop them (columns whose
  // support is a pad index), keep the S columns living on real indices
  int kk = 0;
  for (int k = 0; k < LD && kk < S; ++k) {
    double nr = 0.0;
    for (int i = 0; i < S; ++i) nr += Up[(size_t)i * LD + k] * Up[(size_t)i * LD + k];
    if (nr > 0.5) {
      lam[kk] = lp[k];
      for (int i = 0; i < S; ++i) U[(size_t)i * S + kk] = Up[(size_t)i * LD + k];
      ++kk;
    }
  }
  if (kk != S) return fail(CB_ENUMERIC, "cb_eigh: found %d real eigenvectors, expected %d", kk, S);
  return CB_OK;
}

#include "general_host.hip.h"

#include "train_host.hip.h"

static int read_phase_times(cb_bank *h, double (&v)[CB_T_COUNT]);

// fold the previous profiled call (its events have normally completed long
// ago) into the running sums
static void fold_pending(cb_bank *h) {
  if (!h->t_pending) return;
  double v[CB_T_COUNT];
  if (read_phase_times(h, v) == CB_OK) {
    for (int i = 0; i < CB_T_COUNT; ++i) h->t_sum[i] += v[i];
    h->t_calls += 1;
  }
  h->t_pending = false;
}

extern "C" int cb_profile(cb_handle h, int enable) {
  if (!h) return fail(CB_EINVAL, "cb_profile: NULL handle");
  h->profile = enable != 0;
  h->profile_every = enable > 1 ? enable : 1;
  h->profile_now = h->profile;
  for (double &x : h->t_sum) x = 0.0;
  h->t_calls = 0;
  h->t_pending = false;
  h->t_pending2 = false;
  return CB_OK;
}

extern "C" int cb_timing_sums(cb_handle h, double *ms_sum, int n, int *calls) {
  if (!h || !ms_sum || !calls) return fail(CB_EINVAL, "cb_timing_sums: NULL argument");
  fold_pending(h);
  for (int i = 0; i < n; ++i) ms_sum[i] = i < CB_T_COUNT ? h->t_sum[i] : 0.0;
  *calls = h->t_calls;
  return CB_OK;
}

extern "C" int cb_last_sweeps(cb_handle h) { return h ? h->last_sweeps : 0; }
#ifdef CB_CLOCK_STAMP
// diagnostic build only (large_bank.hip.h): out[3][4096][6]: the last launch's stamps per workgroup of K1..K3
extern "C" int cb_debug_clock_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cb_clock_stamps), sizeof(unsigned long long) * 3 * 4096 * 6) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int cb_eigh_counters(cb_handle h, int *counts, int n) {
  if (!h || !counts || n < 0) return fail(CB_EINVAL, "cb_eigh_counters: NULL argument");
  const int v[4] = {h->planned_solves, h->planned_stalls, h->last_sweeps,
                    (int)std::min<long long>(h->record_spins, 0x7fffffff)};
  for (int i = 0; i < n && i < 4; ++i) counts[i] = v[i];
  return CB_OK;
}
extern "C" int cb_last_kernel_form(cb_handle h) { return h ? h->last_form : 0; }
extern "C" int cb_last_bank_form(cb_handle h) {
  return h ? (h->bank_fused ? 1 : 0) | (h->bank_kg == 2 ? 2 : 0) | (h->bank_accum ? 4 : 0) | (h->bank_tb ? 8 : 0) : 0;
}
extern "C" int cb_time_basis_info(cb_handle h, int *n, double *rho_max) {
  if (!h || !n || !rho_max) return fail(CB_EINVAL, "cb_time_basis_info: NULL argument");
  n[0] = h->tb.ns;
  n[1] = h->tb.nd;
  n[2] = h->tb.ng;
  n[3] = h->tb_builds;
  n[4] = h->tb_stale_epochs;
  *rho_max = h->tb.rho_max;
  return CB_OK;
}

extern "C" int cb_last_timings(cb_handle h, double *ms, int n) {
  if (!h || !ms) return fail(CB_EINVAL, "cb_last_timings: NULL argument");
  for (int i = 0; i < n; ++i) ms[i] = 0.0;
  double v[CB_T_COUNT];
  int rc = read_phase_times(h, v);
  if (rc != CB_OK) return rc;
  for (int i = 0; i < n && i < CB_T_COUNT; ++i) ms[i] = v[i];
  return CB_OK;
}

static int read_phase_times(cb_bank *h, double (&v)[CB_T_COUNT]) {
  for (double &x : v) x = 0.0;
  if (!h->ev_rec[EV_START]) return fail(CB_EINVAL, "cb_last_timings: no profiled call recorded");
  {
    int last = EV_START;
    for (int i = 0; i <= CB_T_COUNT; ++i)
      if (h->ev_rec[i] && i != EV_END && i != EV_AR) last = i;
    if (h->ev_rec[EV_AR]) last = EV_AR;
    HIP_TRY(hipEventSynchronize(h->ev[last]));
  }
  auto span = [&](int a, int b) -> double {
    if (!h->ev_rec[a] || !h->ev_rec[b]) return 0.0;
    float t = 0.f;
    if (hipEventElapsedTime(&t, h->ev[a], h->ev[b]) != hipSuccess) return 0.0;
    return (double)t;
  };
  if (h->large) {
    v[CB_T_EIGH] = span(EV_START, EV_EIGH);
    if (h->ev_rec[EV_K3] && !h->ev_rec[EV_K1] && !h->ev_rec[EV_K2]) {
      v[CB_T_K1] = span(EV_EIGH, EV_K3);   // fused bank launch (k123_bank): K1 + K2 + K3 (+ the spectral tables) in one span
    } else {
      v[CB_T_K1] = span(EV_EIGH, EV_K1);
      v[CB_T_K2] = span(EV_K1, EV_K2);
      v[CB_T_K3] = span(EV_K2, EV_K3);
    }
    v[CB_T_K4] = span(EV_K3, EV_K4);
    v[CB_T_ALLREDUCE] = span(EV_K4, EV_AR);
    v[CB_T_TOTAL] = span(EV_START, h->ev_rec[EV_K4] ? EV_K4 : EV_K1);
  } else {
    v[CB_T_SMALL] = span(EV_START, EV_SMALL);
    v[CB_T_TOTAL] = v[CB_T_SMALL];
  }
  return CB_OK;
}
