// Host side of the planned (device-controlled) warm solves of eigh_planned.hip.h: the plan, its launches, the record a
// finished solve leaves in pinned memory and the next plan derived from it.  Nothing here waits for the GPU: the only
// reader of the record is the trainer loop (train_host.hip.h), which looks at it while an epoch of bank kernels is queued.
// Included by cherrybank.hip after eigh_large_host.hip.h (launch_sg).
#pragma once

// (EighSlot / EighPlan: handle_host.hip.h -- the handle keeps the plan between CB_TRAIN_RESUME calls)
struct EighRecord {
  bool stall = false;
  int err = 0, nsweep = 0, final_slot = -1;
  double sigma = 0.0;      // max |A_ii| of the matrix the solve was given
  bool tb_stale = false;   // ... left the range of the bank's time basis: the bank behind the solve returned at once
  struct {
    double c, rs, rsf;
    int order, sq, slot;
    bool masked, damped;
  } sweep[EC_MAXREC];
};

// (round 6: 1 instead of 2 -- the two cross rounds of block pairs (i, i + 2) cost 4 launches of ~10 us per solve and buy 0.14
// sweeps on the recorded trajectory (profiles/tools/r6_eigh_proto.py: 3.52 against 3.38 sweeps per solve); measured on the bench
// bank: eigh phase 0.393 -> 0.373 ms in the driver's window, 0.262 -> 0.252 over 200 epochs, profiles/r06_eigh_anatomy.json)
#ifndef CB_PLANNED_BAND
#define CB_PLANNED_BAND 1
#endif
#ifndef CB_LEAD_INNER
#define CB_LEAD_INNER 2
#endif
static const int kPlannedBand = CB_PLANNED_BAND;   // blocks: pairs nearer than this are rotated exactly by the band passes

static void eigh_plan_default(EighPlan &p, int extra = 0) {
  p = EighPlan{};
  p.lead_band = 1;
  p.nslots = std::min<int>(EC_MAXREC, 7 + extra);
  for (int i = 0; i < p.nslots; ++i) {
    p.slot[i].cap = i < 5 + extra ? 12 : 4;
    p.slot[i].nsq = i == 0 ? 2 : (i < 5 + extra ? 1 : 0);
    p.slot[i].band_after = i < 3 + extra ? 1 : 0;   // (a masked sweep without a band pass behind it gets nowhere)
    p.slot[i].so = i >= 1 ? 1 : 0;
  }
}

// The next solve is planned like the last one went, with margins: one spare sweep, the twelfth-order launches in every slot but
// the last, at least one squaring launch in each of them, a band pass behind every sweep that was masked or close to it.
static void eigh_plan_from_record(const EighRecord &r, const EighPlan &prev, EighPlan &p) {
  (void)prev;
  if (r.stall || r.nsweep <= 0) {
    eigh_plan_default(p, 2);
    return;
  }
  p = EighPlan{};
  const int n = std::min<int>(r.nsweep, EC_MAXREC - 1);
  p.nslots = n + 1;
  for (int i = 0; i < n; ++i) {
    const auto &s = r.sweep[i];
    const double rsu = s.masked ? s.rsf : s.rs;
    EighSlot &q = p.slot[i];
    // (the generator's norm at a given position moves by orders of magnitude from one epoch to the next -- a solve that needs a
    // sweep more, an isolated near-degenerate pair --, and a slot that cannot evaluate the order it meets is a lost (damped)
    // sweep and, as a rule, a stalled solve: every slot but the last one gets the twelfth-order launches, the last one the
    // two-product fourth-order form when the last generator there was 4x below its limit)
    q.cap = (i + 1 < n || i < 2 || s.order >= 8 || rsu > 5e-4 || s.damped) ? 12 : 4;   // (never the first two: a short solve
                                                                                        // is followed by longer ones)
    if (q.cap == 12) {
      int need = s.sq + (s.damped ? 1 : 0);
      if (rsu * std::ldexp(1.0, -s.sq) > 0.3) ++need;
      // (at least one: with Jacobi angles an isolated near-degenerate pair puts up to pi / 4 into a row sum from one epoch
      // to the next -- twice the twelfth-order limit; without the squaring that sweep is a damped one and the plan runs out)
      q.nsq = std::min(2, std::max(1, need));
    }
    // a band pass behind the sweep when it is expected to be a masked one (near the rule's thresholds counts)
    q.band_after = (s.masked || s.c > 2e-4 || s.rs > 0.3) ? 1 : 0;
    q.so = (i >= 1 && !s.masked) ? 1 : 0;   // (never in the first sweep: profiles/tools/r6_eigh_proto.py, and the default plan)
    q.expect_run = 1;
    q.expect_order = std::min(s.order, q.cap);
    q.expect_sq = s.sq;
  }
  // the spare (runs when the solve needs a sweep more than last time: as a rule the final one, |X| ~ 1e-7 .. 1e-4)
  p.slot[n] = EighSlot{4, 0, 0, 1, 0, 2, 0};
  // the band pass in front pays while near-degenerate neighbours are far from separated (profiles/tools/eigh_proto.py)
  p.lead_band = (r.sweep[0].masked && r.sweep[0].c > 1e-3) ? 1 : 0;
}

// More slots for a solve whose plan ended before convergence (it continues from its current state).
static void eigh_plan_continue(const EighRecord &r, EighPlan &p) {
  eigh_plan_default(p, 0);
  const bool masked = r.nsweep > 0 && r.sweep[std::min(r.nsweep, (int)EC_MAXREC) - 1].masked;
  p.lead_band = masked ? 1 : 0;
}

static bool eigh_planned_setup(cb_bank *h) {
  if (h->ectl) return true;
  // a failed attempt is not repeated: what it did allocate stays with the handle (freed with it), and a second attempt
  // would allocate everything again -- the pinned block without anyone left to free the first one
  if (h->planned_unavailable) return false;
  if (h->LD / 16 > LGE_ACC_NT) return false;   // (the statistics lines have room for 32 block rows: beyond, the host-driven solver)
  h->planned_unavailable = true;   // (cleared at the end)
  const int LD = h->LD;
  void *q = nullptr;
  if (hipMalloc(&q, EC_WORDS * sizeof(unsigned long long)) != hipSuccess) return false;
  h->allocs.push_back(q);
  unsigned long long *ctl = static_cast<unsigned long long *>(q);
  if (hipMalloc(&q, ((size_t)2 * LGE_ACC_WORDS + LD) * sizeof(double)) != hipSuccess) return false;   // statistics lines, Gamma's diagonal
  h->allocs.push_back(q);
  h->epart = static_cast<double *>(q);
  if (!h->epin) {
    if (hipHostMalloc(&q, 2 * (EC_WORDS + 16) * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    memset(q, 0, 2 * (EC_WORDS + 16) * sizeof(unsigned long long));
    h->epin = static_cast<unsigned long long *>(q);
  }
  const int RS = LD + ((2 - LD % 32 + 32) % 32);
  const size_t lds = (size_t)(16 * RS + 3 * 16 * 17 + JB_WAVES * 256) * sizeof(double);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(lgj_round), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return false;
  if (hipMemsetAsync(ctl, 0, EC_WORDS * sizeof(unsigned long long), h->stream) != hipSuccess) return false;
  h->ectl = ctl;
  h->planned_unavailable = false;
  return true;
}

// Enqueue one warm solve (h->U / h->Vc hold the previous eigenvectors, h->A the new matrix).  `seq` is what lge_norms
// leaves in the record's sequence word.  first_slot > 0: the CONTINUATION of a stalled solve -- its G buffers hold a valid,
// partly converged state (every rotation applied so far was orthogonal), so the new slots simply carry on from it.
#ifdef CB_NANCHECK
#define NANCHECK(buf, tag) hipLaunchKernelGGL(lge_nancheck, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream, LD, buf, (unsigned long long)(tag), ctl)
#else
#define NANCHECK(buf, tag) ((void)0)
#endif
static int enqueue_planned_solve(cb_bank *h, const EighPlan &p, unsigned long long seq, int first_slot = 0, double tb_rho_max = 0.0,
                                 const TbTableArgs &tbt = TbTableArgs{}) {
  const int LD = h->LD, nt = LD / 16, nb = LD / JB_W;
  const size_t LL = (size_t)LD * LD;
  unsigned long long *ctl = h->ectl;
  unsigned long long *eacc = reinterpret_cast<unsigned long long *>(h->epart);   // two sets of statistics lines (slot parity)
  double *Gb[2] = {h->Gc, h->Gc2};
  double *X = h->gx, *Xf = h->gx + LL, *P2 = h->gx + 2 * LL, *P3 = h->gx + 3 * LL, *P4 = h->gx + 4 * LL, *B0 = h->gx + 5 * LL,
         *B1 = h->gx + 6 * LL, *B2 = h->gx + 7 * LL, *T = h->gx + 8 * LL, *R0 = h->gx + 9 * LL, *R0t = h->gx + 10 * LL,
         *R1 = h->gx + 11 * LL, *R1t = h->gx + 6 * LL,   // (B1 is dead once R_0 exists)
         *Rfin = h->gx + 3 * LL;   // (the finished rotation lives where Gamma was: lge_so has read that before any product runs, and X^3 is
                                   // never stored -- lge_p34 uses it in registers)
  const int RS = LD + ((2 - LD % 32 + 32) % 32);
  const size_t lds = (size_t)(16 * RS + 3 * 16 * 17 + JB_WAVES * 256) * sizeof(double);
  if (first_slot > 0) {
    hipLaunchKernelGGL(lge_resume, dim3(1), dim3(64), 0, h->stream, ctl);
  } else {
    // (the trainer's lt_build has reset the control block and left sigma: h->begin_folded)
    if (!h->begin_folded) hipLaunchKernelGGL(lge_begin, dim3(1), dim3(256), 0, h->stream, LD, h->A, h->sigma, ctl, eacc);
    NANCHECK(h->U, 1);
    NANCHECK(h->A, 2);
    // warm start G = A' U_prev:  Gc[k][r] = sum_j U_prev[j][k] A[j][r] - sigma Ut_prev[k][r]
    hipLaunchKernelGGL(lge_plain, dim3((unsigned)(nt * nt)), dim3(512), 0, h->stream, LD, h->U, h->A, h->Vc, h->sigma, Gb[0], nullptr);
  }
  unsigned long long *jstate = ctl + EC_JSTATE;   // lgj_round's own two words (zeroed by lge_begin)
  int shift = 0;
  auto band_pass = [&](double *G, const unsigned long long *must_nonzero) {
    for (int w = 0; w < 2; ++w)
      hipLaunchKernelGGL(lgj_round, dim3(nb / 2), dim3(JB_THREADS), lds, h->stream, LD, ((shift + w) & 1) && nb > 2 ? -2 : -1,
                         must_nonzero ? 1 : CB_LEAD_INNER, G,   // (inner sweeps: two in front of the first sweep, one behind a masked one)
                         jstate, ctl + EC_STALL, must_nonzero);
    for (int k = 2; k <= kPlannedBand && k < nb; ++k)
      for (int par = 0; par < 2; ++par)
        hipLaunchKernelGGL(lgj_round, dim3((unsigned)(((nb + 2 * k - 1) / (2 * k)) * k)), dim3(JB_THREADS), lds, h->stream, LD,
                           -(10 + 2 * (k - 2) + par), 0, G, jstate, ctl + EC_STALL, must_nonzero);
    ++shift;
  };
  if (p.lead_band) band_pass(Gb[first_slot & 1], nullptr);
  const dim3 tiles((unsigned)(nt * nt));
  for (int i = 0; i < p.nslots; ++i) {
    const EighSlot &q = p.slot[i];
    const int s = first_slot + i;
    double *Gin = Gb[s & 1], *Gout = Gb[(s + 1) & 1];
    // (Gamma goes to P3's buffer -- dead until the powers are formed --, its diagonal behind the statistics partials)
    unsigned long long *acc = eacc + (size_t)(s & 1) * LGE_ACC_WORDS;
    GramArgs ga{LD, kPlannedBand, s, q.cap, q.cap == 12 ? q.nsq : 0, Gin, X, Xf, P3, h->epart + (size_t)2 * LGE_ACC_WORDS,
                q.so, acc, ctl, 3e-4};
    NANCHECK(Gin, 1000 + s * 100 + 1);
    hipLaunchKernelGGL(lge_gram, tiles, dim3(512), 0, h->stream, ga);
    NANCHECK(X, 1000 + s * 100 + 2);
    // the sweep's decision is taken by the first launch behind lge_gram
    DecArgs dec;
    dec.on = 1; dec.cap = ga.cap; dec.nsq = ga.nsq; dec.so = q.so; dec.acc = acc; dec.trigger = ga.trigger;
#ifdef CB_DECIDE_KERNEL
    hipLaunchKernelGGL(lge_decide_k, dim3(1), dim3(64), 0, h->stream, s, dec, ctl);
    dec.on = 0;
#endif
    if (q.so) {
      SoArgs so{LD, s, ctl, ga.Gm, ga.dg, X, Xf, q.expect_run && i >= 1, dec};
#ifdef CB_NO_EARLY
      so.early = 0;
#endif
      hipLaunchKernelGGL(lge_so, tiles, dim3(512), 0, h->stream, so);
      NANCHECK(Xf, 1000 + s * 100 + 3);
    }
    EgArgs e{};
    if (!q.so) e.dec = dec;
    e.zacc = acc;
    e.LD = LD; e.slot = s; e.cap = q.cap; e.ctl = ctl; e.X = X; e.Xf = Xf; e.P2 = P2; e.P4 = P4; e.B0 = B0; e.B1 = B1; e.B2 = B2; e.T = T;
    e.R[0] = R0; e.R[1] = R1; e.Rt[0] = R0t; e.Rt[1] = R1t; e.Rfin = Rfin; e.Gin = Gin; e.Gout = Gout;
    auto gemm = [&](int kind, int qq = 0) {
      EgArgs c = e;
      c.kind = kind;
      c.q = qq;
      const int xo = q.expect_order;
      c.early = !q.expect_run ? 0 : kind == EG_P34 ? xo >= 4 : kind == EG_T1 ? xo == 12 : kind == EG_RP ? xo >= 8 :
                kind == EG_SQ ? qq < q.expect_sq : kind == EG_R4 ? xo == 4 : 1;
#ifdef CB_NO_EARLY
      c.early = 0;
#endif
      switch (kind) {
        case EG_P2: hipLaunchKernelGGL(lge_gemm<EG_P2>, tiles, dim3(512), 0, h->stream, c); break;
        case EG_P34: hipLaunchKernelGGL(lge_p34, tiles, dim3(512), 0, h->stream, c); break;
        case EG_T1: hipLaunchKernelGGL(lge_gemm<EG_T1>, tiles, dim3(512), 0, h->stream, c); break;
        case EG_RP: hipLaunchKernelGGL(lge_gemm<EG_RP>, tiles, dim3(512), 0, h->stream, c); break;
        case EG_SQ: hipLaunchKernelGGL(lge_gemm<EG_SQ>, tiles, dim3(512), 0, h->stream, c); break;
        case EG_R4: hipLaunchKernelGGL(lge_gemm<EG_R4>, tiles, dim3(512), 0, h->stream, c); break;
        default: hipLaunchKernelGGL(lge_gemm<EG_GR>, tiles, dim3(512), 0, h->stream, c); break;
      }
    };
    gemm(EG_P2);
    if (q.cap == 12) {
      gemm(EG_P34);    // X^3, X^4 and the polynomial's coefficient matrices (no launch of their own any more)
      gemm(EG_T1);
      gemm(EG_RP);
      for (int k = 0; k < q.nsq; ++k) gemm(EG_SQ, k);
    } else if (q.cap == 4) {
      gemm(EG_R4);   // (second order: EG_P2 has written R already and this launch returns)
    }
    NANCHECK(Rfin, 1000 + s * 100 + 4);
#ifdef CB_NANCHECK
    hipLaunchKernelGGL(lge_orthcheck, dim3((unsigned)LD), dim3(256), 0, h->stream, LD, Rfin, (unsigned long long)s, ctl);
#endif
    gemm(EG_GR);
    NANCHECK(Gout, 1000 + s * 100 + 5);
    if (q.band_after) band_pass(Gout, ctl + EC_MASKED);
  }
  volatile unsigned long long *pin = h->epin + (size_t)(seq & 1ull) * (EC_WORDS + 16);
  hipLaunchKernelGGL(lge_norms, dim3((LD + 3) / 4), dim3(256), 0, h->stream, LD, Gb[0], Gb[1], h->X, ctl, pin, seq, h->sigma, tb_rho_max);
  LAUNCH_STOP(stop_event(h, EV_EIGH), lge_finish, dim3((LD + 3) / 4 + 1), dim3(256), 0, h->stream, LD, Gb[0], Gb[1], h->X, h->sigma, h->lam, h->U, h->Vc,
                     ctl, tbt, pin, seq);
#ifdef CB_NANCHECK
  hipLaunchKernelGGL(lge_residual, dim3((unsigned)LD), dim3(256), 0, h->stream, LD, h->A, h->Vc, h->lam, ctl);
#endif
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

// The record of solve `seq`, once lge_norms has published it.  The caller has an epoch of bank kernels queued behind the
// solve, so looking here is not on the GPU's critical path; after 200 ms without the word the stream is synchronised.
static int eigh_planned_record(cb_bank *h, unsigned long long seq, EighRecord &r) {
  volatile unsigned long long *pin = h->epin + (size_t)(seq & 1ull) * (EC_WORDS + 16);
  const auto t0 = std::chrono::steady_clock::now();
  bool got = false;
  for (unsigned it = 0;; ++it) {
    if (pin[EC_WORDS] == seq) {
      got = true;
      break;
    }
    if (it == 0) ++h->record_spins;   // (cb_eigh_counters: the host looked before the record was there)
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((it & 4095u) == 4095u && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 200.0) break;
  }
  if (!got) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (pin[EC_WORDS] != seq) return fail(CB_EHIP, "eigensolver: the record of the planned solve never reached the host");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  r = EighRecord{};
  r.stall = pin[EC_STALL] != 0ull;
  r.err = (int)pin[EC_ERR];
  {
    const unsigned long long ns = pin[EC_NSWEEP];
    r.nsweep = (int)(ns < (unsigned long long)EC_MAXREC ? ns : (unsigned long long)EC_MAXREC);
  }
  r.final_slot = pin[EC_FINAL] == EC_NONE ? -1 : (int)pin[EC_FINAL];
  {
    const unsigned long long sb = pin[EC_SIGMA];
    memcpy(&r.sigma, &sb, 8);
  }
  r.tb_stale = pin[EC_TBSTALE] != 0ull;
  for (int k = 0; k < r.nsweep; ++k) {
    unsigned long long w[4] = {pin[EC_REC + 4 * k], pin[EC_REC + 4 * k + 1], pin[EC_REC + 4 * k + 2], pin[EC_REC + 4 * k + 3]};
    memcpy(&r.sweep[k].c, &w[0], 8);
    memcpy(&r.sweep[k].rs, &w[1], 8);
    memcpy(&r.sweep[k].rsf, &w[2], 8);
    r.sweep[k].order = (int)(w[3] & 255ull);
    r.sweep[k].masked = (w[3] & 256ull) != 0ull;
    r.sweep[k].damped = (w[3] & 512ull) != 0ull;
    r.sweep[k].sq = (int)((w[3] >> 24) & 255ull);
    r.sweep[k].slot = (int)(w[3] >> 32);
  }
  if (getenv("CB_DEBUG")) {
    fprintf(stderr, "[cherrybank] planned eigh %llu:%s", seq, r.stall ? " STALL" : "");
    for (int k = 0; k < r.nsweep; ++k)
      fprintf(stderr, " %s%d%s/%d c=%.1e |X|<=%.1e(%.1e)", r.sweep[k].masked ? "M" : "L", r.sweep[k].order, r.sweep[k].damped ? "d" : "",
              r.sweep[k].sq, r.sweep[k].c, r.sweep[k].rs, r.sweep[k].rsf);
    // where the solve's time went on the device's own 100 MHz clock: begin -> each decision -> lge_norms, in microseconds
    fprintf(stderr, " final slot %d (slots", r.final_slot);
    for (int k = 0; k < r.nsweep; ++k) fprintf(stderr, " %d", r.sweep[k].slot);
    fprintf(stderr, ")  | us:");
    unsigned long long tp = pin[EC_T0];
    for (int k = 0; k < r.nsweep; ++k) {
      fprintf(stderr, " %.1f", (double)(pin[EC_TSWEEP + k] - tp) * 0.01);
      tp = pin[EC_TSWEEP + k];
    }
    fprintf(stderr, " %.1f = %.1f\n", (double)(pin[EC_TEND] - tp) * 0.01, (double)(pin[EC_TEND] - pin[EC_T0]) * 0.01);
#ifdef CB_NANCHECK
    {
      double rr;
      const unsigned long long rb = pin[92];
      memcpy(&rr, &rb, 8);
      double oo;
      const unsigned long long ob = pin[91];
      memcpy(&oo, &ob, 8);
      fprintf(stderr, "  | nancheck tag %llu, largest residual since the last print %.2e; largest |R^T R - I| of this solve %.2e (slot %llu)\n", pin[94], rr, oo, pin[90]);
    }
#endif
#ifdef CB_EIGH_STAMPS
    {   // launch by launch: kernel id : microseconds since the previous entry
      const unsigned long long ns = pin[EC_NSTAMP];
      const int n = (int)(ns < 96ull ? ns : 96ull);
      fprintf(stderr, "  | entries (id:us since the previous entry; 3 band, 4 gram, 5 decide, 6 so, 10 P2, 12 T1, 13 RP, 14 SQ, 15 GR, 16 R4, 17 P34, 20 norms):");
      unsigned long long prev = pin[EC_T0];
      for (int i = 0; i < n; ++i) {
        const unsigned long long w = pin[EC_STAMPS + i];
        fprintf(stderr, " %d:%.1f", (int)(w & 255ull), (double)((w >> 8) - prev) * 0.01);
        prev = w >> 8;
      }
      fprintf(stderr, "\n");
    }
#endif
  }
  return CB_OK;
}
