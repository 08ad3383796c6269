// Host driver of the general (non-reversible) path: scaling and squaring + exact adjoint (kernels in
// general_small.hip.h / general_large.hip.h).  Included by cherrybank.hip.
#pragma once
// ------------------------------------------------------------- general (non-reversible) path
// S > 32: batched 80 x 80-tile GEMMs over the buckets (general_large.hip.h has the algebra)
static int general_large_run(cb_bank *h, const double *Qd, int flags, double *lossd, double *dQd, double *Pd) {
  if (h->L != 1) return fail(CB_EUNSUPPORTED, "general path, S > 32: L == 1 banks only");
  if (!Pd && h->dtype == CB_F32)
    return fail(CB_EUNSUPPORTED, "general (non-reversible) path: CB_F64 / CB_MIXED handles only (the counts of a CB_F32 handle are float32)");
  const int S = h->S, LD = h->LD, B = Pd ? h->B : h->Bl, Bcap = h->B_cap;
  // counts-free handles never run the adjoint: the Horner iterates and the squarings ping-pong between two
  // slots instead of keeping all 17 + s_max of them (a 2047-node family at 400 states: 10 GB instead of 100)
  const bool lean = h->expm_only;
  const int n_horner = lean ? 2 : GL_DEG - 1;
  const std::vector<double> &th = Pd ? h->t_host : h->t_live_host;
  const size_t LL = (size_t)LD * LD, BL = (size_t)B * LL, capBL = (size_t)Bcap * LL;
  const int nt32 = (LD + 31) / 32;
  auto &w = h->gl;
  if (!w.Qn) {
    bool ok = dev_alloc(h, &w.Qn, LL) == CB_OK && dev_alloc(h, &w.QT, LL) == CB_OK && dev_alloc(h, &w.colsum, LD) == CB_OK &&
              dev_alloc(h, &w.alpha, Bcap) == CB_OK && dev_alloc(h, &w.nsq, Bcap) == CB_OK &&
              dev_alloc(h, &w.R, n_horner * capBL) == CB_OK && dev_alloc(h, &w.RT, n_horner * capBL) == CB_OK &&
              (lean || (dev_alloc(h, &w.G, 2 * capBL) == CB_OK && dev_alloc(h, &w.GT, 2 * capBL) == CB_OK &&
                        dev_alloc(h, &w.Xbar, capBL) == CB_OK && dev_alloc(h, &w.lpart, (size_t)Bcap * nt32 * nt32) == CB_OK));
    if (!ok) return CB_ENOMEM;
  }
  hipLaunchKernelGGL(gl_prep, dim3(LD), dim3(256), 0, h->stream, S, LD, Qd, w.Qn, w.QT, w.colsum);
  std::vector<double> cs(LD);
  HIP_TRY(hipMemcpyAsync(cs.data(), w.colsum, LD * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  double norm1 = 0.0;
  for (double v : cs) norm1 = (v == v) ? std::max(norm1, v) : INFINITY;
  if (!std::isfinite(norm1)) return fail(CB_ENUMERIC, "general path: non-finite rate matrix");
  std::vector<double> alpha(B);
  std::vector<int> nsq(B);
  int smax = 0;
  for (int b = 0; b < B; ++b) {
    const double x = th[b] * norm1;
    int sq = 0;
    if (x > 1.0) sq = (int)std::ceil(std::log2(x));
    if (sq > 60) return fail(CB_ENUMERIC, "general path: |t Q|_1 = %g needs %d squarings", x, sq);
    nsq[b] = sq;
    alpha[b] = std::ldexp(th[b], -sq);
    smax = std::max(smax, sq);
  }
  const int need_slots = lean ? std::min(smax + 1, 2) : smax + 1;
  if (w.cap_slots < need_slots) {   // (an outgrown stack stays allocated until cb_destroy)
    w.E = w.ET = nullptr;
    if (dev_alloc(h, &w.E, (size_t)need_slots * capBL) != CB_OK || dev_alloc(h, &w.ET, (size_t)need_slots * capBL) != CB_OK)
      return CB_ENOMEM;
    w.cap_slots = need_slots;
  }
  HIP_TRY(hipMemcpyAsync(w.alpha, alpha.data(), B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(w.nsq, nsq.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));   // alpha / nsq live on this stack frame
  const int tn = (LD + LG_TN - 1) / LG_TN;
  const dim3 grid((unsigned)(tn * tn * B)), blk(LG4_THREADS);
  const unsigned nel = (unsigned)((BL + 255) / 256);
  auto Rn = [&](int k) { return w.R + (size_t)(lean ? k & 1 : k - 2) * BL; };    // R_k, k = 2..18
  auto Rt = [&](int k) { return w.RT + (size_t)(lean ? k & 1 : k - 2) * BL; };
  auto En = [&](int i) { return w.E + (size_t)(lean ? i & 1 : i) * BL; };
  auto Et = [&](int i) { return w.ET + (size_t)(lean ? i & 1 : i) * BL; };
  auto gemm = [&](BgArgs a) {
    a.LD = LD;
    a.B = B;
    hipLaunchKernelGGL(bg_gemm, grid, blk, 0, h->stream, a);
  };
  // ---- forward: Horner, then the squarings
  hipLaunchKernelGGL(gl_first, dim3(nel), dim3(256), 0, h->stream, LD, B, w.Qn, w.alpha, Rn(GL_DEG), Rt(GL_DEG));
  for (int k = GL_DEG - 1; k >= 1; --k) {   // R_k = I + (a_b / k) Q R_{k+1}
    BgArgs a{};
    a.A1 = w.QT; a.sA1 = 0; a.B1 = Rn(k + 1); a.sB1 = LL;
    a.C = k >= 2 ? Rn(k) : En(0); a.CT = k >= 2 ? Rt(k) : Et(0); a.sC = LL;
    a.alpha = w.alpha; a.scale = 1.0 / k; a.add_identity = 1.0;
    gemm(a);
  }
  for (int i = 1; i <= smax; ++i) {         // E_i = E_{i-1} E_{i-1} for the buckets with i <= s_b
    BgArgs a{};
    a.A1 = Et(i - 1); a.sA1 = LL; a.B1 = En(i - 1); a.sB1 = LL; a.C = En(i); a.CT = Et(i); a.sC = LL;
    a.scale = 1.0; a.nsq = w.nsq; a.round = i;
    gemm(a);
  }
  const double inv_n = (flags & CB_NORMALIZE) ? 1.0 / (h->comm ? h->n_global[0] : h->n_host[0]) : 1.0;
  GlLoss gl{S, LD, B, w.E, w.ET, w.nsq, lean ? 1 : ~0, h->Ct, inv_n, w.G, w.GT, BL, w.lpart, Pd};
  hipLaunchKernelGGL(gl_loss, dim3(nt32, nt32, B), dim3(32, 8), 0, h->stream, gl);
  if (Pd) {
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  hipLaunchKernelGGL(gl_finish_loss, dim3(1), dim3(256), 0, h->stream, w.lpart, B * nt32 * nt32, inv_n, lossd);
  if (dQd) {
    // ---- backward through the squarings: round i reads half i & 1 of the ping-pong buffers, writes half (i - 1) & 1
    for (int i = smax; i >= 1; --i) {       // Ebar_{i-1} = Ebar_i E_{i-1}^T + E_{i-1}^T Ebar_i
      const size_t in = (size_t)(i & 1) * BL, out = (size_t)((i - 1) & 1) * BL;
      BgArgs a{};
      a.A1 = w.GT + in; a.sA1 = LL; a.B1 = Et(i - 1); a.sB1 = LL;
      a.A2 = En(i - 1); a.sA2 = LL; a.B2 = w.G + in; a.sB2 = LL;
      a.C = w.G + out; a.CT = w.GT + out; a.sC = LL; a.scale = 1.0; a.nsq = w.nsq; a.round = i;
      gemm(a);
    }
    // ---- backward through Horner: Hbar_1 = Ebar_0 sits in half 0 for every bucket
    int cur = 0;
    for (int k = 1; k <= GL_DEG - 1; ++k) {
      const size_t in = (size_t)cur * BL, out = (size_t)(cur ^ 1) * BL;
      BgArgs x{};                              // Xbar_b (+)= (a_b / k) Hbar_k R_{k+1}^T
      x.A1 = w.GT + in; x.sA1 = LL; x.B1 = Rt(k + 1); x.sB1 = LL; x.C = w.Xbar; x.sC = LL;
      x.alpha = w.alpha; x.scale = 1.0 / k; x.accumulate = k > 1;
      gemm(x);
      BgArgs g{};                              // Hbar_{k+1} = (a_b / k) Q^T Hbar_k
      g.A1 = w.Qn; g.sA1 = 0; g.B1 = w.G + in; g.sB1 = LL; g.C = w.G + out; g.CT = w.GT + out; g.sC = LL;
      g.alpha = w.alpha; g.scale = 1.0 / k;
      gemm(g);
      cur ^= 1;
    }
    hipLaunchKernelGGL(gl_last, dim3(nel), dim3(256), 0, h->stream, LD, B, w.G + (size_t)cur * BL, w.alpha, w.Xbar);
    hipLaunchKernelGGL(gl_reduce, dim3((unsigned)((S * S + 255) / 256)), dim3(256), 0, h->stream, S, LD, B, w.Xbar, dQd);
  }
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

static int general_run(cb_bank *h, const double *Qd, int flags, double *lossd, double *dQd,
                       double *Pd) {
  if (h->large) return general_large_run(h, Qd, flags, lossd, dQd, Pd);
  const int NW = h->L < 512 ? 8 : 4;
  if (!h->gn_scratch) {
    const size_t waves = (size_t)h->L * NW;
    int rc = dev_alloc(h, &h->gn_scratch, waves * GN_SLOTS * GN_MAT);
    if (rc != CB_OK) return rc;
    rc = dev_alloc(h, &h->gn_partial, waves * (GN_MAT + 1));
    if (rc != CB_OK) return rc;
    h->gn_nw = NW;
  }
  GeneralArgs a{};
  a.S = h->S; a.L = h->L;
  if (Pd) { a.B = h->B; a.t = h->t; a.nlive = nullptr; }            // expm: every bucket, original order
  else { a.B = h->Bl; a.t = h->t_live; a.nlive = h->nlive; }        // loss: live buckets only
  a.Ct = h->Ct; a.inv_n = (flags & CB_NORMALIZE) ? (h->comm ? h->inv_n_global : h->inv_n) : h->ones;
  a.Q = Qd; a.loss = lossd; a.dQ = dQd; a.P = Pd;
  a.scratch = h->gn_scratch; a.partial = h->gn_partial;
  if (NW == 8) hipLaunchKernelGGL(general_bank_kernel<8>, dim3(h->L), dim3(512), 0, h->stream, a);
  else hipLaunchKernelGGL(general_bank_kernel<4>, dim3(h->L), dim3(256), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  return CB_OK;
}

extern "C" int cb_loss_grad_general(cb_handle h, const double *Q, int flags, double *loss,
                                    double *dQ) {
  if (!h || !Q || !loss) return fail(CB_EINVAL, "cb_loss_grad_general: NULL argument");
  if (h->expm_only) return fail(CB_EINVAL, "cb_loss_grad_general: the handle was created with CB_EXPM_ONLY (no counts)");
  HIP_TRY(hipSetDevice(h->dev));
  const size_t SS = (size_t)h->S * h->S;
  const bool devp = flags & CB_PTR_DEVICE;
  const double *Qd = Q;
  double *lossd = loss, *dQd = dQ;
  if (!devp) {
    HIP_TRY(hipMemcpyAsync(h->Q, Q, h->L * SS * sizeof(double), hipMemcpyHostToDevice, h->stream));
    Qd = h->Q;
    lossd = h->loss;
    dQd = dQ ? h->dQ : nullptr;
  }
  int rc = general_run(h, Qd, flags, lossd, dQd, nullptr);
  if (rc != CB_OK) return rc;
  if ((rc = allreduce_results(h, lossd, dQd)) != CB_OK) return rc;
  if (!devp) {
    HIP_TRY(hipMemcpyAsync(loss, h->loss, h->L * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (dQ)
      HIP_TRY(hipMemcpyAsync(dQ, h->dQ, h->L * SS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  return finish_call(h, flags);
}
