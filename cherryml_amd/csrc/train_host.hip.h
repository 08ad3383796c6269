// Host drivers of the fused trainers: which kernels run for which (S, L, parameterisation) and the C-driven epoch
// loop of the 400-state path (kernels in train_small.hip.h / train_large.hip.h).  Included by cherrybank.hip.
#pragma once
// ------------------------------------------------------------- fused trainers
// The one-kernel trainer (one workgroup per site for all epochs): what is left to it after the three-launch splits took
// S <= 24 (any L) and the single 25 .. 32-state bank -- SEVERAL sites, or the SiteRM parameterisation, at 25 .. 32 states.
template <int NW>
static int launch_train_nw(cb_bank *h, const TrainArgs &a) {
  static_assert(NW == 4, "25 .. 32 states: the four-wave form only (see the kernels' launch bounds)");
  const size_t lds = (SmallLds<NW>::TOTAL + 72) * sizeof(double);
  if (h->S <= 24) return fail(CB_EINVAL, "internal: %d states dispatched to the one-kernel trainer", h->S);
  auto kern = small_train_kernel<2, 8, NW>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(h->L), dim3(NW * 64), lds, h->stream, a);
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

// the 100 MHz wall clock at the head of a training call's epochs (the reference point of TrainArgs / LargeTrain::time_curve)
__global__ void tr_stamp(double *out) { *out = (double)__builtin_amdgcn_s_memrealtime(); }

// df_res's `time` column (trainer.py:207-217: seconds since the start at the end of every epoch) without returning to the
// host between epochs: the device stamps its constant 100 MHz clock at the end of every epoch's parameter step
// (time_curve[e]) and once at the head of the loop (time_curve[E], tr_stamp -- enqueued on an idle stream `t_first` seconds
// after the call was entered).
static void set_epoch_seconds(cb_bank *h, const char *ticks, int E, double t_first_s) {
  h->epoch_seconds.assign((size_t)std::max(E, 0), 0.0);
  if (!ticks || E <= 0) return;
  std::vector<double> tk((size_t)E + 1);
  memcpy(tk.data(), ticks, ((size_t)E + 1) * sizeof(double));
  for (int e = 0; e < E; ++e) h->epoch_seconds[e] = t_first_s + (tk[e] - tk[E]) * 1e-8;
}

// S > 32 (one bank, pande_reversible): the epoch loop driven from here, kernels of train_large.hip.h
static int run_fused_training_large(cb_bank *h, double *pi_param, double *up_param, const double *mask, int E,
                                    double lr, int do_adam, int flags, double *loss_curve, double *Q_best,
                                    double *Q_last, double *Q_pow2, int n_pow2) {
  HIP_TRY(hipSetDevice(h->dev));
  const bool dbg = getenv("CB_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_enter = now();
  const int S = h->S, LD = h->LD;
  const size_t SS = (size_t)S * S, nup = (size_t)S * (S - 1) / 2;
  // CB_TRAIN_RESUME: parameters, moments, best iterate and the bookkeeping words stay where the previous call left
  // them (workspace slots of unchanged size are never moved; the loss-curve slot may be, it carries no state)
  const bool resume = (flags & CB_TRAIN_RESUME) != 0;
  // what a resumed call must repeat exactly (else the best-loss word would compare losses of two different problems):
  // optimiser, learning rate, normalisation, and the CONTENTS of the mask (FNV-1a over its bytes)
  // (word-wise, four independent lanes: byte by byte the 1.28 MB mask cost 1.1 ms of host time at the head of EVERY call,
  // with the GPU idle -- 55 us per epoch of a 20-epoch call)
  uint64_t sig = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    uint64_t lane[4] = {sig, sig ^ 0x9e3779b97f4a7c15ull, sig ^ 0xc2b2ae3d27d4eb4full, sig ^ 0x165667b19e3779f9ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
      uint64_t w[4];
      memcpy(w, c + i, 32);
      for (int l = 0; l < 4; ++l) lane[l] = (lane[l] ^ w[l]) * 1099511628211ull;
    }
    sig = ((lane[0] * 31 + lane[1]) * 31 + lane[2]) * 31 + lane[3];
    for (; i < n; ++i) sig = (sig ^ c[i]) * 1099511628211ull;
  };
  {
    const int head[3] = {mask ? 1 : 0, do_adam ? 1 : 0, (flags & CB_NORMALIZE) ? 1 : 0};
    mix(head, sizeof head);
    mix(&lr, sizeof lr);
    if (mask) mix(mask, SS * sizeof(double));
    if (sig == 0) sig = 1;
  }
  if (resume && (h->tr_epochs <= 0 || h->tr_sig != sig))
    return fail(CB_EINVAL, "CB_TRAIN_RESUME: no finished training call with the same mask / optimiser / learning rate / "
                           "normalisation on this handle");
  if (resume && Q_pow2) return fail(CB_EINVAL, "CB_TRAIN_RESUME: Q_pow2 must be NULL");
  const int e0 = resume ? h->tr_epochs : 0;
  int slot = 0;
  auto alloc = [&](double **p, size_t n) -> bool { return ws_get(h, slot++, n, p); };
  // (every exit: h->sigma back where the other entry points expect it -- the epochs below use two words in turn, see lt_build)
  auto release = [&]() {
    (void)hipStreamSynchronize(h->stream);
    if (h->sigma_home) h->sigma = h->sigma_home;
    h->begin_folded = false;
    h->profile_now = h->profile;
  };
  double *d_pi = nullptr, *d_up = nullptr, *d_mom = nullptr, *d_mask = nullptr, *d_loss = nullptr, *d_Qb = nullptr,
         *d_Ql = nullptr, *d_Qp = nullptr, *d_vec = nullptr, *d_time = nullptr;
  const size_t nmom = 2 * (S + nup);
  // fixed slots (an optional buffer keeps its number): a resumed call finds the state where the first call put it
  auto at = [&](int s, double **p, size_t n) -> bool { return ws_get(h, s, n, p); };
  bool ok = at(0, &d_pi, S) && at(1, &d_up, nup) && at(2, &d_mom, nmom) && at(3, &d_loss, E) && at(4, &d_Qb, SS) &&
            at(5, &d_Ql, SS) && (!mask || at(6, &d_mask, SS)) &&
            (!(Q_pow2 && n_pow2 > 0) || at(7, &d_Qp, std::max<size_t>(n_pow2, 16) * SS)) && at(8, &d_vec, (size_t)LD + S + 8) &&
            at(9, &d_time, (size_t)E + 1);
  (void)alloc;
  if (!ok) {
    release();
    return fail(CB_ENOMEM, "fused training: device allocation failed");
  }
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: workspaces ready after %.2f ms\n", now() - t_enter);
  int rc = CB_OK;
#define TRYH(expr) \
  if (rc == CB_OK && (expr) != hipSuccess) rc = fail(CB_EHIP, "%s failed", #expr)
  {
    const size_t up_bytes = (S + nup + SS + 64) * sizeof(double);
    const size_t down_bytes = (S + nup + 2 * (size_t)E + 16 + (2 + (size_t)(d_Qp ? n_pow2 : 0)) * SS + 64) * sizeof(double);
    if (!pin_reserve(h, std::max(up_bytes, down_bytes) + 1024)) {
      release();
      return fail(CB_ENOMEM, "fused training: pinned staging allocation failed");
    }
  }
  if (!resume) {
    TRYH(h2d_staged(h, d_pi, pi_param, S * sizeof(double)));
    TRYH(h2d_staged(h, d_up, up_param, nup * sizeof(double)));
    TRYH(hipMemsetAsync(d_mom, 0, nmom * sizeof(double), h->stream));
    TRYH(hipMemsetAsync(d_Qb, 0, SS * sizeof(double), h->stream));
    TRYH(hipMemsetAsync(d_Ql, 0, SS * sizeof(double), h->stream));
  }
  if (mask && !resume) TRYH(h2d_staged(h, d_mask, mask, SS * sizeof(double)));   // (a resumed call: the same mask, see `sig`, still there)
  const double init_state[2] = {INFINITY, INFINITY};   // best loss so far, two words in turn (lt_step)
  LargeTrain a{};
  a.S = S; a.LD = LD; a.do_adam = do_adam; a.n_pow2 = d_Qp ? n_pow2 : 0;
  a.epoch0 = e0;
  a.p_pi = d_pi; a.p_up = d_up;
  a.m_pi = d_mom; a.v_pi = d_mom + S; a.m_up = d_mom + 2 * (size_t)S; a.v_up = a.m_up + nup;
  a.mask = d_mask; a.lr = lr; a.beta1 = 0.9; a.beta2 = 0.999; a.eps = 1e-8;
  a.pi = d_vec; a.gd = d_vec + LD; a.state = d_vec + LD + S;
  a.dsq = h->dsq; a.A = h->A; a.G = h->Mt; a.loss = h->loss;
  // sharded job (cb_allreduce_setup): this rank's buckets give partial sums; (loss, dL/dA) are
  // all-reduced every epoch below, the count margins and the normaliser are the job-wide ones
  a.dirsum = h->comm ? h->dirsum_g : h->dirsum;
  a.inv_n = (flags & CB_NORMALIZE) ? 1.0 / (h->comm ? h->n_global[0] : h->n_host[0]) : 1.0;
  a.loss_curve = d_loss; a.time_curve = d_time; a.Q_last = d_Ql; a.Q_best = d_Qb; a.Q_pow2 = d_Qp;
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: copies enqueued after %.2f ms\n", now() - t_enter);
  if (!resume) TRYH(h2d_staged(h, a.state, init_state, sizeof init_state));
  TRYH(hipStreamSynchronize(h->stream));  // init_state is on this stack frame
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: synced after %.2f ms\n", now() - t_enter);
  if (h->profile) fold_pending(h);
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: parameters uploaded after %.2f ms\n", now() - t_enter);
  double pow_b1 = resume ? h->tr_pow_b1 : 1.0, pow_b2 = resume ? h->tr_pow_b2 : 1.0;
  // sigma = max |A_ii| is folded into one of two words by lt_build (the other is cleared for the next epoch)
  if (!h->sigma2) {
    if (dev_alloc(h, &h->sigma2, 2) != CB_OK) {
      release();
      return CB_ENOMEM;
    }
  }
  if (!h->sigma_home) h->sigma_home = h->sigma;
  TRYH(hipMemsetAsync(h->sigma2, 0, 2 * sizeof(double), h->stream));
  h->last_form = 4000;
  h->tr_epochs = 0;   // (set again when this call succeeds)
  // fault injection for the tests of the collective failure protocol: this rank's evaluation "fails" at that epoch
  const int fault_epoch = cb_test_hook("CB_FAULT_INJECT") ? atoi(cb_test_hook("CB_FAULT_INJECT")) : -1000;
  // planned solves: float64 work on LD x LD matrices with LD a multiple of 16 and at least 8 column blocks (the hybrid
  // scheme's own condition); CB_EIGH_HOST=1 keeps the host-driven loop of rounds 1-3
  const bool planned = !cb_test_hook("CB_EIGH_HOST") && !cb_test_hook("CB_NO_HYBRID") && LD % 16 == 0 && LD / JB_W >= 8 && eigh_planned_setup(h);
  EighPlan &plan = h->eplan;
  if (!resume) eigh_plan_default(plan);
  // the bank's time basis (tbasis.hip.h) is part of the optimisation's state as well: a fresh optimisation builds its own from
  // its first matrix (the same bits whatever ran on the handle before), a resumed one continues with the one it has
  if (!resume) {
    tb_drop_next(h);
    h->tb = CbTimeBasisHost{};
    h->tb_failed = false;
  }
  // test hook: every plan cut down to one sweep, so that every solve stalls and is continued (tests/test_gpu_s400_full.py)
  // (= 2: that one sweep with the second-order polynomial only, so that it is also a DAMPED one -- exp(alpha X), alpha << 1)
  const int short_plans = cb_test_hook("CB_EIGH_SHORT_PLAN") ? std::max(1, atoi(cb_test_hook("CB_EIGH_SHORT_PLAN"))) : 0;
  auto shorten = [&]() {
    if (!short_plans) return;
    plan.nslots = 1;
    if (short_plans >= 2) plan.slot[0] = EighSlot{2, 0, plan.slot[0].band_after, 0};
  };
  shorten();
  // CB_TRACE_SLOW=<ms>: report every epoch whose HOST side took longer than that (where the host waited: the fold of an older
  // epoch's events, the enqueue, the planned solve's record) -- the tool for "one run in five is 3x slower"
  const double trace_slow = getenv("CB_TRACE_SLOW") ? atof(getenv("CB_TRACE_SLOW")) : 0.0;
  const double t_first_s = (now() - t_enter) * 1e-3;   // (the stream is idle: synchronised above)
  hipLaunchKernelGGL(tr_stamp, dim3(1), dim3(1), 0, h->stream, d_time + E);
  for (int e = 0; e < E && rc == CB_OK; ++e) {
    const double te0 = trace_slow > 0.0 ? now() : 0.0;
    double te_fold = 0.0, te_enq = 0.0, te_rec = 0.0;
    // (cb_profile(h, n): the phase events of every n-th epoch -- each completion event costs the epoch ~2.5 us, six of them 2 %)
    h->profile_now = h->profile && (e0 + e) % h->profile_every == 0;
    if (h->profile_now) {  // fold the profiled epoch before the previous one (its events are long complete), then re-record that set
      swap_event_sets(h);
      fold_pending(h);
    }
    if (trace_slow > 0.0) te_fold = now();
    clear_marks(h);
    const bool use_plan = planned && h->have_prev && e0 + e >= 3;
    {
      const int par = (e0 + e) & 1;
      h->sigma = h->sigma2 + par;
      a.sig_cur = reinterpret_cast<unsigned long long *>(h->sigma2 + par);
      a.sig_next = reinterpret_cast<unsigned long long *>(h->sigma2 + (par ^ 1));
      a.ectl = use_plan ? h->ectl : nullptr;                                        // the planned solve's prologue rides on the build
      a.eacc = use_plan ? reinterpret_cast<unsigned long long *>(h->epart) : nullptr;
      h->begin_folded = use_plan;
    }
    LAUNCH_STOP(stop_event(h, EV_START), lt_build, dim3(LD), dim3(256), 0, h->stream, a, e0 + e);
    // Every solve after the first is a PLAN (eigh_planned_host.hip.h): the device takes the sweep decisions, the host enqueues
    // the whole epoch and only then looks at the solve's record -- with K1 .. K4 queued behind it, so the GPU never waits.
    // (the first warm solves of an optimisation start far from converged -- cosines of 1e-2, every sweep damped --: they stay
    // with the host-driven solver, whose tournament sweeps are made for that; plans from the fourth epoch on)
    rc = large_eval(h, flags & CB_NORMALIZE, h->loss, h->Mt, true, nullptr, false, use_plan ? &plan : nullptr);
    h->begin_folded = false;   // (a solve enqueued later in this epoch starts with its own prologue)
    if (trace_slow > 0.0) te_enq = now();
    if (rc == CB_OK && use_plan) {
      EighRecord rec;
      rc = eigh_planned_record(h, h->eseq, rec);
      ++h->planned_solves;
      if (rc == CB_OK && rec.err == 2) rc = fail(CB_ENUMERIC, "eigensolver: non-finite input");
      int slots_done = plan.nslots;
      for (int attempt = 0; rc == CB_OK && rec.stall && attempt < 3; ++attempt) {
        // the plan ended before the solve converged: U, lambda untouched, the queued K1 .. K3 returned at once (EC_STALL).
        // The solve CONTINUES from its current state with more slots, and the epoch's kernels are enqueued again.
        ++h->planned_stalls;
        EighPlan more;
        eigh_plan_continue(rec, more);
        clear_marks(h);
        mark(h, EV_START);
        rc = large_eval(h, flags & CB_NORMALIZE, h->loss, h->Mt, true, nullptr, false, &more, slots_done);
        slots_done += more.nslots;
        if (rc == CB_OK) rc = eigh_planned_record(h, h->eseq, rec);
        if (rc == CB_OK && rec.err == 2) rc = fail(CB_ENUMERIC, "eigensolver: non-finite input");
      }
      bool host_solved = false;
      if (rc == CB_OK && rec.stall) {   // still not converged: the host-driven solver, from the previous eigenvectors
        host_solved = true;
        // (errors flow through rc: a sharded job's failure protocol and the clean-up below depend on it)
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(CB_EHIP, "hipStreamSynchronize failed (host-driven solver fallback)");
        else if (hipMemsetAsync(h->ectl, 0, sizeof(unsigned long long), h->stream) != hipSuccess)
          rc = fail(CB_EHIP, "hipMemsetAsync failed (host-driven solver fallback)");
        clear_marks(h);
        mark(h, EV_START);
        if (rc == CB_OK) rc = large_eval(h, flags & CB_NORMALIZE, h->loss, h->Mt, true, nullptr);
      }
      if (rc == CB_OK && !host_solved && h->bank_tb) {
        // The time basis against this solve's sigma.  Out of range (lge_norms saw it too: the bank, the reduction and K4 returned
        // at once): the evaluation is repeated on the finished decomposition with per-bucket products.  Close to the end of
        // the range, or far below it: a new basis, built by a helper thread beside the epochs (tb_maintain, cherrybank.hip).
        if (rec.tb_stale) {
          clear_marks(h);
          mark(h, EV_START);
          h->tb_block = true;
          rc = large_eval(h, flags & CB_NORMALIZE, h->loss, h->Mt, true, nullptr, true, nullptr);
          h->tb_block = false;
        }
        if (rc == CB_OK) rc = tb_maintain(h, h->Bl, e0 + e, 2.0 * rec.sigma, rec.tb_stale);
        if (rc == CB_OK && rec.tb_stale) ++h->tb_stale_epochs;
      }
      if (rc == CB_OK) {
        const EighPlan prev = plan;
        eigh_plan_from_record(rec, prev, plan);
        shorten();
        h->last_sweeps = rec.nsweep;
      }
    }
    if (trace_slow > 0.0) {
      te_rec = now();
      if (te_rec - te0 > trace_slow)
        fprintf(stderr, "[cherrybank] slow epoch %d (host): fold %.2f ms, enqueue %.2f ms, record / continuations %.2f ms\n", e0 + e,
                te_fold - te0, te_enq - te_fold, te_rec - te_enq);
    }
    if (rc == CB_OK && fault_epoch == e) rc = fail(CB_ENUMERIC, "injected fault at epoch %d (CB_FAULT_INJECT)", e);
    if (rc != CB_OK && h->comm) {
      // A rank that fails alone (its eigensolver met a non-finite matrix, say) must not leave its peers
      // waiting in this epoch's ncclAllReduce -- and a host-side status exchange per epoch would cost a
      // stream synchronisation.  So it keeps its place in EVERY remaining collective with NaN payloads:
      // the peers' parameters turn NaN with the next step, their own eigensolver reports "non-finite
      // input", they do the same, and all ranks return an error after the same number of collectives.
      const std::string first_error = g_err;
      for (int e2 = e; e2 < E; ++e2) {
        (void)hipMemsetAsync(h->loss, 0xFF, sizeof(double), h->stream);
        (void)hipMemsetAsync(h->Mt, 0xFF, (size_t)LD * LD * sizeof(double), h->stream);
        if (h->allreduce(h->loss, h->loss, 1, 8, 0, h->comm, h->stream) != 0 ||
            h->allreduce(h->Mt, h->Mt, (size_t)LD * LD, 8, 0, h->comm, h->stream) != 0)
          break;
      }
      g_err = first_error + " (this rank sent NaN to the remaining all-reduces so that its peers fail too)";
      break;
    }
    if (rc != CB_OK) break;
    if (h->comm) {  // one all-reduce of LD^2 + 1 doubles per epoch (RCCL, on this stream); identical Adam steps follow
      int ar = h->allreduce(h->loss, h->loss, 1, 8, 0, h->comm, h->stream);
      if (ar == 0) ar = h->allreduce(h->Mt, h->Mt, (size_t)LD * LD, 8, 0, h->comm, h->stream);
      if (ar != 0) {
        rc = fail(CB_EHIP, "ncclAllReduce failed with code %d", ar);
        break;
      }
      mark(h, EV_AR);   // CB_T_ALLREDUCE = the span from the end of K4 to here
    }
    if (h->profile_now) h->t_pending = true;
    pow_b1 *= a.beta1;
    pow_b2 *= a.beta2;
    hipLaunchKernelGGL(lt_gd, dim3((S + 3) / 4), dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL(lt_step, dim3(S + 1), dim3(256), 0, h->stream, a, e0 + e, 1.0 - pow_b1, std::sqrt(1.0 - pow_b2));
    if (hipGetLastError() != hipSuccess) rc = fail(CB_EHIP, "fused training launch failed");
  }
  if (E > 0 && h->sigma_home) TRYH(hipMemcpyAsync(h->sigma_home, h->sigma2 + ((e0 + E - 1) & 1), sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  TRYH(hipStreamSynchronize(h->stream));
  if (h->profile) {  // the older of the two event sets; the newest stays pending (cb_last_timings reads it)
    swap_event_sets(h);
    fold_pending(h);
    swap_event_sets(h);
  }
  h->pin_off = 0;  // uploads are consumed
  char *s_pi = nullptr, *s_up = nullptr, *s_loss = nullptr, *s_Qb = nullptr, *s_Ql = nullptr, *s_Qp = nullptr;
  TRYH(d2h_staged(h, d_pi, S * sizeof(double), &s_pi));
  TRYH(d2h_staged(h, d_up, nup * sizeof(double), &s_up));
  if (loss_curve && E > 0) TRYH(d2h_staged(h, d_loss, (size_t)E * sizeof(double), &s_loss));
  if (Q_best) TRYH(d2h_staged(h, d_Qb, SS * sizeof(double), &s_Qb));
  if (Q_last) TRYH(d2h_staged(h, d_Ql, SS * sizeof(double), &s_Ql));
  if (d_Qp) TRYH(d2h_staged(h, d_Qp, n_pow2 * SS * sizeof(double), &s_Qp));
  char *s_time = nullptr;
  TRYH(d2h_staged(h, d_time, ((size_t)E + 1) * sizeof(double), &s_time));
  TRYH(hipStreamSynchronize(h->stream));
  set_epoch_seconds(h, rc == CB_OK ? s_time : nullptr, E, t_first_s);
  if (rc == CB_OK) {
    memcpy(pi_param, s_pi, S * sizeof(double));
    memcpy(up_param, s_up, nup * sizeof(double));
    if (s_loss) memcpy(loss_curve, s_loss, (size_t)E * sizeof(double));
    if (s_Qb) memcpy(Q_best, s_Qb, SS * sizeof(double));
    if (s_Ql) memcpy(Q_last, s_Ql, SS * sizeof(double));
    if (s_Qp) memcpy(Q_pow2, s_Qp, n_pow2 * SS * sizeof(double));
  }
#undef TRYH
  release();
  if (rc == CB_OK) {   // what a CB_TRAIN_RESUME call continues
    h->tr_epochs = e0 + E;
    h->tr_sig = sig;
    h->tr_pow_b1 = pow_b1;
    h->tr_pow_b2 = pow_b2;
  }
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: %d epochs done after %.2f ms\n", E, now() - t_enter);
  return rc;
}

// shared host driver: parameters in, E epochs on the device, results out
static int run_fused_training(cb_bank *h, int kind, double *pi_param, double *up_param,
                              const double *mask, int E, double lr, int do_adam, int flags,
                              double *loss_curve, double *Q_best, double *Q_last, double *Q_pow2,
                              int n_pow2) {
  if (E < 0) return fail(CB_EINVAL, "fused training: num_epochs < 0");
  if (h->expm_only) return fail(CB_EINVAL, "fused training: the handle was created with CB_EXPM_ONLY (no counts)");
  if (h->large) {
    if (kind != 0) return fail(CB_EUNSUPPORTED, "fused SiteRM training: S <= 32 only (S = %d)", h->S);
    return run_fused_training_large(h, pi_param, up_param, mask, E, lr, do_adam, flags, loss_curve, Q_best, Q_last,
                                    Q_pow2, n_pow2);
  }
  if (flags & CB_TRAIN_RESUME)
    return fail(CB_EUNSUPPORTED, "CB_TRAIN_RESUME: S > 32 only (the small-state trainers run their epochs inside one launch)");
  if (h->comm)
    return fail(CB_EUNSUPPORTED, "fused training with cb_allreduce_setup: S > 32 only (a small bank does not shard; "
                                 "sites are independent)");
  HIP_TRY(hipSetDevice(h->dev));
  const int S = h->S, L = h->L;
  const size_t SS = (size_t)S * S, nup = kind == 0 ? (size_t)S * (S - 1) / 2 : SS;
  double *d_pi = nullptr, *d_up = nullptr, *d_mom = nullptr, *d_mask = nullptr, *d_loss = nullptr,
         *d_Qb = nullptr, *d_Ql = nullptr, *d_Qp = nullptr, *d_time = nullptr;
  const auto t_enter = std::chrono::steady_clock::now();
  int slot = 0;
  auto alloc = [&](double **p, size_t n) -> bool { return ws_get(h, slot++, n, p); };
  auto release = [&]() { (void)hipStreamSynchronize(h->stream); };
  const size_t nmom = 2 * ((size_t)L * S + (size_t)L * nup);
  bool ok = alloc(&d_pi, (size_t)L * S) && alloc(&d_up, L * nup) && alloc(&d_mom, nmom) &&
            alloc(&d_loss, (size_t)E * L) && alloc(&d_Qb, L * SS) && alloc(&d_Ql, L * SS) &&
            (!mask || alloc(&d_mask, SS)) && (!(Q_pow2 && n_pow2 > 0) || alloc(&d_Qp, std::max<size_t>(n_pow2, 16) * SS)) &&
            alloc(&d_time, (size_t)E + 1);
  if (!ok) {
    release();
    return fail(CB_ENOMEM, "fused training: device allocation failed");
  }
  int rc = CB_OK;
#define TRYH(expr)                                                                  \
  if (rc == CB_OK && (expr) != hipSuccess) rc = fail(CB_EHIP, "%s failed", #expr)
  {
    const size_t up_bytes = ((size_t)L * S + L * nup + SS + 64) * sizeof(double);
    const size_t down_bytes = ((size_t)L * S + L * nup + (size_t)E * L + E + 16 + 2 * L * SS + (size_t)(d_Qp ? n_pow2 : 0) * SS + 64) * sizeof(double);
    if (!pin_reserve(h, std::max(up_bytes, down_bytes) + 1024)) {
      release();
      return fail(CB_ENOMEM, "fused training: pinned staging allocation failed");
    }
  }
  TRYH(h2d_staged(h, d_pi, pi_param, (size_t)L * S * sizeof(double)));
  TRYH(h2d_staged(h, d_up, up_param, L * nup * sizeof(double)));
  TRYH(hipMemsetAsync(d_mom, 0, nmom * sizeof(double), h->stream));
  TRYH(hipMemsetAsync(d_Qb, 0, L * SS * sizeof(double), h->stream));
  TRYH(hipMemsetAsync(d_Ql, 0, L * SS * sizeof(double), h->stream));
  if (mask) TRYH(h2d_staged(h, d_mask, mask, SS * sizeof(double)));
  double t_first_s = 0.0;
  if (rc == CB_OK) {
    // (the uploads above are a few hundred kilobytes on an otherwise idle stream: the stamp runs microseconds after this)
    t_first_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count();
    hipLaunchKernelGGL(tr_stamp, dim3(1), dim3(1), 0, h->stream, d_time + E);
  }
  if (rc == CB_OK) {
    TrainArgs a{};
    a.S = S; a.L = L; a.B = h->Bl; a.E = E; a.kind = kind; a.do_adam = do_adam; a.n_pow2 = d_Qp ? n_pow2 : 0;
    a.nlive = h->nlive; a.t = h->t_live; a.Ct = h->Ct; a.Cq = h->Cq; a.nq = h->nq; a.inv_n = (flags & CB_NORMALIZE) ? h->inv_n : h->ones; a.dirsum = h->dirsum;
    a.p_pi = d_pi; a.p_up = d_up;
    a.m_pi = d_mom; a.v_pi = d_mom + (size_t)L * S;
    a.m_up = d_mom + 2 * (size_t)L * S; a.v_up = a.m_up + L * nup;
    a.mask = d_mask; a.lr = lr; a.beta1 = 0.9; a.beta2 = 0.999; a.eps = 1e-8;
    a.loss_curve = d_loss; a.time_curve = d_time; a.Q_best = d_Qb; a.Q_last = d_Ql; a.Q_pow2 = d_Qp;
    a.sym = (S <= 24 && h->sym_counts) ? 1 : 0;
    for (bool &b : h->ev_rec) b = false;
    mark(h, EV_START);
    // which kernels (cb_last_kernel_form): S <= 24 -- the site-parallel split, any L, both parameterisations;
    // 24 < S <= 32 -- one pande_reversible bank: the LG split; several sites or SiteRM: the one-kernel trainer
    const bool site_split = S <= 24;
    const bool split = !site_split && kind == 0 && L == 1;
    if (E > 0 && site_split) {
      // three launches per epoch over all sites (train_small.hip.h: sp_prepare / sp_bank / sp_finish)
      SpSplit g{};
      int max_live = 1;
      for (int l = 0; l < L; ++l) max_live = std::max(max_live, h->nlive_host[l]);
      const int nquads = (max_live + 3) / 4;
      // few sites: one quad per wave spreads a single bank over the chip; many sites: longer chunks
      const int qpw = L < 64 ? 1 : 3;
      g.quads_per_chunk = std::min(nquads, 4 * qpw);
      g.nchunk = (nquads + g.quads_per_chunk - 1) / g.quads_per_chunk;
      g.quads_per_chunk = (nquads + g.nchunk - 1) / g.nchunk;  // even split
      double *buf = nullptr;
      const size_t nbuf = (size_t)L * LGS_TOTAL + (size_t)L * g.nchunk * 577 + L + 24;
      if (!alloc(&buf, nbuf)) rc = fail(CB_ENOMEM, "fused training: device allocation failed");
      if (rc == CB_OK) {
        g.frames = buf;
        g.Mpart = buf + (size_t)L * LGS_TOTAL;
        g.lpart = g.Mpart + (size_t)L * g.nchunk * 576;
        g.best = g.lpart + (size_t)L * g.nchunk;
        const size_t lds_p = SPP_TOTAL * sizeof(double), lds_f = SPF_TOTAL * sizeof(double);
        const int TS = quad_ts(S);
        const size_t lds_s = std::max(lds_p, lds_f);
        // few sites (one LG-sized bank): finish(e - 1) and prepare(e) are ONE launch (sp_step), two launches per epoch
        const bool fuse = L < 64;
        double pow_b1 = 1.0, pow_b2 = 1.0, bc1_prev = 0.0, bc2s_prev = 0.0;
        const dim3 gb((unsigned)((size_t)L * g.nchunk));
        const bool w3 = (size_t)L * g.nchunk > 512;   // more workgroups than two per CU can hold at once: the three-per-CU form
        h->last_form = 1000 + 100 * TS + (a.sym ? 10 : 0) + (w3 ? 1 : 0);
#define SPK(T)                                                                                         \
  do {                                                                                                 \
    if (fuse && e > 0) hipLaunchKernelGGL((sp_step<T>), dim3(L), dim3(256), lds_s, h->stream, a, g, e, bc1_prev, bc2s_prev);                      \
    else if (fuse) hipLaunchKernelGGL(sp_prepare<256>, dim3(L), dim3(256), lds_p, h->stream, a, g, e);                                           \
    else hipLaunchKernelGGL(sp_prepare<64>, dim3(L), dim3(64), lds_p, h->stream, a, g, e);                                                       \
    if (a.sym && w3) hipLaunchKernelGGL((sp_bank<T, true, true>), gb, dim3(256), spb_total(T, true, true) * sizeof(double), h->stream, a, g);      \
    else if (a.sym) hipLaunchKernelGGL((sp_bank<T, true, false>), gb, dim3(256), spb_total(T, true, false) * sizeof(double), h->stream, a, g);  \
    else if (w3) hipLaunchKernelGGL((sp_bank<T, false, true>), gb, dim3(256), spb_total(T, false, true) * sizeof(double), h->stream, a, g);     \
    else hipLaunchKernelGGL((sp_bank<T, false, false>), gb, dim3(256), spb_total(T, false, false) * sizeof(double), h->stream, a, g);           \
    if (!fuse || e == E - 1) hipLaunchKernelGGL((sp_finish<T>), dim3(L), dim3(256), lds_f, h->stream, a, g, e, bc1, bc2s);                       \
  } while (0)
        for (int e = 0; e < E && rc == CB_OK; ++e) {
          pow_b1 *= a.beta1;
          pow_b2 *= a.beta2;
          const double bc1 = 1.0 - pow_b1, bc2s = std::sqrt(1.0 - pow_b2);
          switch (TS) {
            case 1: SPK(1); break;
            case 2: SPK(2); break;
            case 4: SPK(4); break;
            case 5: SPK(5); break;
            default: SPK(6); break;
          }
          bc1_prev = bc1;
          bc2s_prev = bc2s;
          if ((e & 63) == 63 && hipGetLastError() != hipSuccess) rc = fail(CB_EHIP, "fused training launch failed");
        }
#undef SPK
#ifdef CB_SP_STAMPS
        if (fuse && E > 301) {
          unsigned long long st[10];
          (void)hipStreamSynchronize(h->stream);
          (void)hipMemcpy(st, g.best, sizeof st, hipMemcpyDeviceToHost);
          // (stamps 0-4: finish(300), inside the launch of epoch 301; stamps 5-8: prepare(300), inside the launch of epoch 300)
          const char *nm[] = {"finish: frames in", "finish: M sum + loss", "finish: dA = U M U^T", "finish: tr_update", "-",
                              "prepare: tr_build", "prepare: eigensolver", "prepare: pad + frames out"};
          for (int i = 0; i < 8; ++i)
            if (i != 4) fprintf(stderr, "[cherrybank] sp_step epoch 300: %-30s %7.2f us\n", nm[i], (double)(st[2 + i] - st[1 + i]) * 0.01);
        }
#endif
      }
    } else if (E > 0 && split) {
      // one LG-sized bank: the epoch spread over the chip, three small launches per epoch
      h->last_form = 2000;
      LgSplit g{};
      double *buf = nullptr;
      const size_t nbuf = LGS_TOTAL + (size_t)h->Bl * 1025 + 8;  // L == 1: nlive[0] == Bl
      if (!alloc(&buf, nbuf)) rc = fail(CB_ENOMEM, "fused training: device allocation failed");
      if (rc == CB_OK) {
        g.frames = buf;
        g.Mpart = buf + LGS_TOTAL;
        g.lpart = g.Mpart + (size_t)h->Bl * 1024;
        g.best = g.lpart + h->Bl;
        const size_t lds_pf = (SmallLds<4>::TOTAL + 72) * sizeof(double);
        const size_t lds_b = SmallLds<4>::TOTAL * sizeof(double);
        const unsigned nblk = (unsigned)((h->Bl + 3) / 4);
        double pow_b1 = 1.0, pow_b2 = 1.0;
        for (int e = 0; e < E && rc == CB_OK; ++e) {
          pow_b1 *= a.beta1;
          pow_b2 *= a.beta2;
          hipLaunchKernelGGL(lg_prepare, dim3(1), dim3(256), lds_pf, h->stream, a, g, e);
#define LGB(NT, KS) hipLaunchKernelGGL((lg_bank<NT, KS>), dim3(nblk), dim3(256), lds_b, h->stream, a, g)
          if (S <= 4) LGB(1, 1);
          else if (S <= 8) LGB(1, 2);
          else if (S <= 16) LGB(1, 4);
          else if (S <= 20) LGB(2, 5);
          else if (S <= 24) LGB(2, 6);
          else LGB(2, 8);
#undef LGB
          if (S <= 16)
            hipLaunchKernelGGL(lg_finish<1>, dim3(1), dim3(256), lds_pf, h->stream, a, g, e, 1.0 - pow_b1,
                               std::sqrt(1.0 - pow_b2));
          else
            hipLaunchKernelGGL(lg_finish<2>, dim3(1), dim3(256), lds_pf, h->stream, a, g, e, 1.0 - pow_b1,
                               std::sqrt(1.0 - pow_b2));
          if ((e & 63) == 63 && hipGetLastError() != hipSuccess) rc = fail(CB_EHIP, "fused training launch failed");
        }
      }
    } else if (E > 0) {
      h->last_form = 3000;
      rc = launch_train_nw<4>(h, a);
    }
    mark(h, EV_SMALL);  // cb_last_timings(): CB_T_SMALL = all E epochs
  }
  TRYH(hipStreamSynchronize(h->stream));
  h->pin_off = 0;  // uploads are consumed
  char *s_pi = nullptr, *s_up = nullptr, *s_loss = nullptr, *s_Qb = nullptr, *s_Ql = nullptr, *s_Qp = nullptr;
  TRYH(d2h_staged(h, d_pi, (size_t)L * S * sizeof(double), &s_pi));
  TRYH(d2h_staged(h, d_up, L * nup * sizeof(double), &s_up));
  if (loss_curve && E > 0) TRYH(d2h_staged(h, d_loss, (size_t)E * L * sizeof(double), &s_loss));
  if (Q_best) TRYH(d2h_staged(h, d_Qb, L * SS * sizeof(double), &s_Qb));
  if (Q_last) TRYH(d2h_staged(h, d_Ql, L * SS * sizeof(double), &s_Ql));
  if (d_Qp) TRYH(d2h_staged(h, d_Qp, n_pow2 * SS * sizeof(double), &s_Qp));
  char *s_time = nullptr;
  TRYH(d2h_staged(h, d_time, ((size_t)E + 1) * sizeof(double), &s_time));
  TRYH(hipStreamSynchronize(h->stream));
  set_epoch_seconds(h, rc == CB_OK ? s_time : nullptr, E, t_first_s);
  if (rc == CB_OK) {
    memcpy(pi_param, s_pi, (size_t)L * S * sizeof(double));
    memcpy(up_param, s_up, L * nup * sizeof(double));
    if (s_loss) memcpy(loss_curve, s_loss, (size_t)E * L * sizeof(double));
    if (s_Qb) memcpy(Q_best, s_Qb, L * SS * sizeof(double));
    if (s_Ql) memcpy(Q_last, s_Ql, L * SS * sizeof(double));
    if (s_Qp) memcpy(Q_pow2, s_Qp, n_pow2 * SS * sizeof(double));
  }
  if (rc == CB_OK) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(CB_EHIP, "fused training kernel failed: %s", hipGetErrorString(e));
  }
#undef TRYH
  release();
  return rc;
}

extern "C" int cb_train_pande_reversible(cb_handle h, double *upper_diag, double *log_pi,
                                         const double *mask, int num_epochs, double lr, int do_adam,
                                         int flags, double *loss_curve, double *Q_best,
                                         double *Q_last, double *Q_pow2, int n_pow2) {
  if (!h || !upper_diag || !log_pi) return fail(CB_EINVAL, "cb_train_pande_reversible: NULL argument");
  // L > 1: L independent problems with the reference's pande_reversible parameterisation each -- the
  // per-site SiteRM loop (_site_specific_rate_matrix.py:43-84, 659-684) as one batched launch;
  // upper_diag [L][S(S-1)/2], log_pi [L][S], loss_curve [E][L], Q_best / Q_last [L][S][S], one shared mask
  if (h->L != 1 && (Q_pow2 || n_pow2 > 0))
    return fail(CB_EUNSUPPORTED, "cb_train_pande_reversible: power-of-two snapshots exist for L == 1 only");
  if (mask)
    for (int i = 0; i < h->S; ++i)
      for (int j = 0; j < i; ++j)
        if (mask[i * h->S + j] != mask[j * h->S + i])
          return fail(CB_EUNSUPPORTED, "cb_train_pande_reversible: mask must be symmetric "
                                       "(a non-symmetric mask makes Q non-reversible)");
  return run_fused_training(h, 0, log_pi, upper_diag, mask, num_epochs, lr, do_adam, flags,
                            loss_curve, Q_best, Q_last, Q_pow2, n_pow2);
}

extern "C" int cb_train_epoch_times(cb_handle h, double *seconds, int n) {
  if (!h || !seconds || n < 0) return fail(CB_EINVAL, "cb_train_epoch_times: NULL argument");
  if ((size_t)n > h->epoch_seconds.size())
    return fail(CB_EINVAL, "cb_train_epoch_times: the last training call on this handle ran %zu epoch(s), %d asked for",
                h->epoch_seconds.size(), n);
  if (n > 0) memcpy(seconds, h->epoch_seconds.data(), (size_t)n * sizeof(double));
  return CB_OK;
}

extern "C" int cb_train_siterm(cb_handle h, double *theta, double *Theta, int num_epochs, double lr,
                               int flags, double *res, double *loss_per_epoch_per_site) {
  if (!h || !theta || !Theta) return fail(CB_EINVAL, "cb_train_siterm: NULL argument");
  return run_fused_training(h, 1, theta, Theta, nullptr, num_epochs, lr, 1, flags | CB_NORMALIZE,
                            loss_per_epoch_per_site, res, nullptr, nullptr, 0);
}
