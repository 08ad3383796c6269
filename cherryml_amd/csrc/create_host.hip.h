// cb_create / cb_destroy and the kernels that lay the counts out once (transposed per bucket, live buckets first,
// quad order for S <= 24).  Included by cherrybank.hip after handle_host.hip.h.
#pragma once
// ---------------------------------------------------------------- create
// per-site totals and (colsum - rowsum) of sum_b C; one block per site
__global__ void prep_counts(int S, int B, const double *C, double *n, double *inv_n, double *ones,
                            double *dirsum) {
  extern __shared__ double sm[];  // tot[S*S]
  const int l = blockIdx.x;
  const double *Cl = C + (size_t)l * B * S * S;
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    double acc = 0.0;
    for (int b = 0; b < B; ++b) acc += Cl[(size_t)b * S * S + e];
    sm[e] = acc;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < S; k += blockDim.x) {
    double cs = 0.0, rs = 0.0;
    for (int i = 0; i < S; ++i) {
      cs += sm[i * S + k];
      rs += sm[k * S + i];
    }
    dirsum[(size_t)l * S + k] = cs - rs;
  }
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int e = 0; e < S * S; ++e) tot += sm[e];
    n[l] = tot;
    inv_n[l] = 1.0 / tot;
    ones[l] = 1.0;
  }
}

// large S: the S*S totals do not fit LDS comfortably; two simple kernels
__global__ void prep_counts_large_tot(int S, int B, const double *C, double *tot) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * S) return;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) acc += C[(size_t)b * S * S + e];
  tot[e] = acc;
}
__global__ void prep_counts_large_fin(int S, const double *tot, double *n, double *inv_n,
                                      double *ones, double *dirsum) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int e = threadIdx.x; e < S * S; e += 256) acc += tot[e];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    n[0] = s[0];
    inv_n[0] = 1.0 / s[0];
    ones[0] = 1.0;
  }
  for (int k = threadIdx.x; k < S; k += 256) {
    double cs = 0.0, rs = 0.0;
    for (int i = 0; i < S; ++i) {
      cs += tot[(size_t)i * S + k];
      rs += tot[(size_t)k * S + i];
    }
    dirsum[k] = cs - rs;
  }
}

// sum |C_b| per (site, bucket): buckets with C_b == 0 add nothing to the loss or its gradient
__global__ void bucket_mass(size_t SS, const double *C, double *mass) {
  __shared__ double s[256];
  const double *Cm = C + (size_t)blockIdx.x * SS;
  double acc = 0.0;
  for (size_t e = threadIdx.x; e < SS; e += 256) acc += fabs(Cm[e]);
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) mass[blockIdx.x] = s[0];
}

// S <= 24: quad order.  Block (l, quad): Cq[(I*TS + J)*64 + lane] = C[l, src[l, 4 quad + blk]][4J + r][4I + q]
// (lane = 16 q + 4 blk + r; transposed like Ct), zero where the slot / row / column does not exist.
__global__ void pack_counts_quad(int S, int B, int Bl, int nq, int TS, const int *nlive, const int *src,
                                 const double *C, double *Cq) {
  const int l = blockIdx.x / nq, quad = blockIdx.x - l * nq;
  double *dst = Cq + (size_t)blockIdx.x * TS * TS * 64;
  for (int e = threadIdx.x; e < TS * TS * 64; e += blockDim.x) {
    const int tile = e >> 6, lane = e & 63, I = tile / TS, J = tile - I * TS;
    const int q = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
    const int k = 4 * quad + blk, row = 4 * I + q, col = 4 * J + r;
    double v = 0.0;
    if (k < nlive[l] && row < S && col < S)
      v = C[((size_t)l * B + src[(size_t)l * Bl + k]) * S * S + (size_t)col * S + row];
    dst[e] = v;
  }
}

// flag[0] |= 1 when some matrix of C [nmat][S][S] is not symmetric (small path)
__global__ void small_sym_check(int S, const double *C, int *flag) {
  const double *M = C + (size_t)blockIdx.x * S * S;
  bool bad = false;
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    const int i = e / S, j = e - i * S;
    if (j < i && M[e] != M[(size_t)j * S + i]) bad = true;
  }
  if (bad) atomicOr(flag, 1);
}

// small path: Ct[l,k][j][i] = C[l,src[l,k]][i][j]  for the live slots k < nlive[l]
__global__ void transpose_small(int S, int B, int Bl, const int *nlive, const int *src, const double *C,
                                double *Ct) {
  const size_t m = blockIdx.x;
  const int l = (int)(m / Bl), k = (int)(m - (size_t)l * Bl);
  if (k >= nlive[l]) return;
  const double *Cs = C + ((size_t)l * B + src[m]) * S * S;
  for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
    const int j = e / S, i = e - j * S;
    Ct[m * S * S + e] = Cs[(size_t)i * S + j];
  }
}

extern "C" int cb_create(int device, int S, int L, int B, int dtype, const double *t, const double *C,
                         int flags, cb_handle *out) {
  if (!out) return fail(CB_EINVAL, "cb_create: out is NULL");
  *out = nullptr;
  if (dtype != CB_F64 && dtype != CB_F32 && dtype != CB_MIXED)
    return fail(CB_EINVAL, "cb_create: dtype must be CB_F64, CB_F32 or CB_MIXED (got %d)", dtype);
  // CB_F32 / CB_MIXED live in the tile kernels of the large path (K1-K3 templated on the element type).  A single bank
  // of ANY size can take that path (LD = 32 at 20 states: the reference's own float32 LG arithmetic, opt-in, slower
  // than the float64 small-state kernels -- an arithmetic mode, not a fast path); batches of sites (L > 1) are float64.
  const bool narrow = dtype != CB_F64 && !(flags & CB_EXPM_ONLY);
  if (narrow && S <= 32 && L != 1)
    return fail(CB_EUNSUPPORTED, "cb_create: CB_F32 / CB_MIXED with S <= 32 are built for single banks only (L == 1; got L=%d): "
                                 "the site-batched small-state kernels are float64", L);
  if (S < 2 || L < 1 || B < 1) return fail(CB_EINVAL, "cb_create: need S>=2, L>=1, B>=1 (got %d,%d,%d)", S, L, B);
  const bool expm_only = (flags & CB_EXPM_ONLY) != 0;
  if (!t || (!C && !expm_only)) return fail(CB_EINVAL, "cb_create: t and C must not be NULL");
  if (S > 32 && L != 1)
    return fail(CB_EUNSUPPORTED, "cb_create: S > 32 is supported for L == 1 only (got L=%d)", L);
  if (narrow && S < 4) return fail(CB_EUNSUPPORTED, "cb_create: CB_F32 / CB_MIXED need S >= 4 (got %d)", S);
  if (S > 1024) return fail(CB_EUNSUPPORTED, "cb_create: S > 1024 not supported");
  int ndev = cb_device_count();
  if (ndev <= 0) return fail(CB_EHIP, "cb_create: no HIP device visible");
  if (device < 0 || device >= ndev) return fail(CB_EINVAL, "cb_create: device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  cb_bank *h = new cb_bank();
  h->dev = device;
  h->S = S;
  h->L = L;
  h->B = B;
  h->B_cap = B;
  h->large = S > 32 || narrow;
  h->dtype = expm_only ? CB_F64 : dtype;   // a counts-free handle has no bank products to narrow
  h->expm_only = expm_only;
  h->per_bucket_products = (flags & CB_PER_BUCKET_PRODUCTS) != 0;
  h->LD = (S + 15) / 16 * 16;
  auto cleanup = [&](int rc) {
    cb_destroy(h);
    return rc;
  };
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
    return cleanup(fail(CB_EHIP, "hipStreamCreate failed"));
  h->stream = h->own_stream;
  const size_t nmat = (size_t)L * B, SS = (size_t)S * S;
  int rc;
#define TRY_ALLOC(ptr, count) \
  if ((rc = dev_alloc(h, &(ptr), (count))) != CB_OK) return cleanup(rc)
#define TRY_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return cleanup(fail(CB_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)));            \
  } while (0)
  TRY_ALLOC(h->t, nmat);
  TRY_ALLOC(h->n_dev, L);
  TRY_ALLOC(h->inv_n, L);
  TRY_ALLOC(h->ones, L);
  TRY_ALLOC(h->dirsum, (size_t)L * S);
  TRY_ALLOC(h->Q, (size_t)L * SS);
  TRY_ALLOC(h->pi, (size_t)L * S);
  TRY_ALLOC(h->loss, L);
  TRY_ALLOC(h->dQ, (size_t)L * SS);
  TRY_ALLOC(h->status, L);
  // raw counts: device copy (temporary when they come from the host)
  const double *Cdev = C;
  double *Ctmp = nullptr;
  const hipMemcpyKind kind = (flags & CB_PTR_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  TRY_HIP(hipMemcpyAsync(h->t, t, nmat * sizeof(double), kind, h->stream));
  if (!(flags & CB_PTR_DEVICE) && !expm_only) {
    hipError_t e = hipMalloc((void **)&Ctmp, nmat * SS * sizeof(double));
    if (e != hipSuccess) return cleanup(fail(CB_ENOMEM, "hipMalloc(C staging) failed"));
    e = hipMemcpyAsync(Ctmp, C, nmat * SS * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) {
      (void)hipFree(Ctmp);
      return cleanup(fail(CB_EHIP, "copy of C failed"));
    }
    Cdev = Ctmp;
  }
  auto free_tmp = [&]() {
    if (Ctmp) {
      (void)hipStreamSynchronize(h->stream);
      (void)hipFree(Ctmp);
      Ctmp = nullptr;
    }
  };
  // ---- live buckets: an exact work reduction: C_b == 0 contributes nothing -------
  int *src_idx = nullptr;
  {
    double *mass_d = nullptr;
    if ((rc = dev_alloc(h, &mass_d, nmat)) != CB_OK || (rc = dev_alloc(h, &h->nlive, L)) != CB_OK) {
      free_tmp();
      return cleanup(rc);
    }
    std::vector<double> mass(nmat, 1.0), th(nmat);   // expm-only: every bucket "live"
    hipError_t e = hipSuccess;
    if (!expm_only) {
      hipLaunchKernelGGL(bucket_mass, dim3((unsigned)nmat), dim3(256), 0, h->stream, SS, Cdev, mass_d);
      e = hipMemcpyAsync(mass.data(), mass_d, nmat * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(th.data(), h->t, nmat * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
      free_tmp();
      return cleanup(fail(CB_EHIP, "cb_create: bucket scan failed: %s", hipGetErrorString(e)));
    }
    h->t_host = th;
    h->nlive_host.assign(L, 0);
    for (int l = 0; l < L; ++l)
      for (int b = 0; b < B; ++b)
        if (mass[(size_t)l * B + b] != 0.0) h->nlive_host[l]++;   // NaN counts stay live (and fail later)
    h->Bl = 1;
    for (int l = 0; l < L; ++l) h->Bl = std::max(h->Bl, h->nlive_host[l]);
    const size_t nl = (size_t)L * h->Bl;
    std::vector<int> src(nl, 0);
    std::vector<double> tl(nl, 1.0);
    for (int l = 0; l < L; ++l) {
      int k = 0;
      for (int b = 0; b < B; ++b)
        if (mass[(size_t)l * B + b] != 0.0) {
          src[(size_t)l * h->Bl + k] = b;
          tl[(size_t)l * h->Bl + k] = th[(size_t)l * B + b];
          ++k;
        }
    }
    h->t_live_host = tl;
    {   // kphi_combine (large_bank.hip.h): |t dlam / 2| <= 0.1 for every bucket
      double tmax = 0.0;
      for (double v : tl) tmax = std::max(tmax, v);
      h->phi_delta = tmax > 0.0 ? 0.2 / tmax : 0.0;
    }
    if ((rc = dev_alloc(h, &src_idx, nl)) != CB_OK || (rc = dev_alloc(h, &h->t_live, nl)) != CB_OK) {
      free_tmp();
      return cleanup(rc);
    }
    e = hipMemcpyAsync(src_idx, src.data(), nl * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->t_live, tl.data(), nl * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->nlive, h->nlive_host.data(), L * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // src / tl are stack-owned
    if (e != hipSuccess) {
      free_tmp();
      return cleanup(fail(CB_EHIP, "cb_create: upload failed: %s", hipGetErrorString(e)));
    }
  }
  const int Bl = h->Bl;
  if (expm_only) {   // placeholders instead of the count statistics
    std::vector<double> one(L, 1.0);
    TRY_HIP(hipMemcpy(h->n_dev, one.data(), L * sizeof(double), hipMemcpyHostToDevice));
    TRY_HIP(hipMemcpy(h->inv_n, one.data(), L * sizeof(double), hipMemcpyHostToDevice));
    TRY_HIP(hipMemcpy(h->ones, one.data(), L * sizeof(double), hipMemcpyHostToDevice));
    TRY_HIP(hipMemset(h->dirsum, 0, (size_t)L * S * sizeof(double)));
  }
  if (!h->large && expm_only) {
    // nothing else: the expm / eigh modes of the small kernels touch no counts
  } else if (!h->large) {
    if ((rc = dev_alloc(h, &h->Ct, (size_t)L * Bl * SS)) != CB_OK) {
      free_tmp();
      return cleanup(rc);
    }
    hipLaunchKernelGGL(prep_counts, dim3(L), dim3(256), SS * sizeof(double), h->stream, S, B, Cdev,
                       h->n_dev, h->inv_n, h->ones, h->dirsum);
    hipLaunchKernelGGL(transpose_small, dim3((unsigned)((size_t)L * Bl)), dim3(256), 0, h->stream, S, B, Bl,
                       h->nlive, src_idx, Cdev, h->Ct);
    if (S <= 24) {
      const int TS = S <= 4 ? 1 : S <= 8 ? 2 : S <= 16 ? 4 : S <= 20 ? 5 : 6;   // = quad_ts(S), the kernels' instantiation
      h->nq = (Bl + 3) / 4;
      if ((rc = dev_alloc(h, &h->Cq, (size_t)L * h->nq * TS * TS * 64)) != CB_OK) {
        free_tmp();
        return cleanup(rc);
      }
      hipLaunchKernelGGL(pack_counts_quad, dim3((unsigned)((size_t)L * h->nq)), dim3(256), 0, h->stream, S, B, Bl,
                         h->nq, TS, h->nlive, src_idx, Cdev, h->Cq);
      // symmetric counts (cherry counting, SiteRM assembly with reverse transitions): sp_bank's symmetric form
      int *flag = h->status;
      (void)hipMemsetAsync(flag, 0, sizeof(int), h->stream);
      hipLaunchKernelGGL(small_sym_check, dim3((unsigned)nmat), dim3(256), 0, h->stream, S, Cdev, flag);
      int hf = 1;
      if (hipMemcpyAsync(&hf, flag, sizeof hf, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
          hipStreamSynchronize(h->stream) == hipSuccess)
        h->sym_counts = hf == 0 && !cb_test_hook("CB_NO_SYM");
      (void)hipMemsetAsync(flag, 0, sizeof(int), h->stream);
    }
  } else {
    const size_t LL = (size_t)h->LD * h->LD;
    const int tiles = ((h->LD + LG_TM - 1) / LG_TM) * ((h->LD + LG_TN - 1) / LG_TN);
    h->k3_chunk = 4;
    h->k3_nchunks = (B + h->k3_chunk - 1) / h->k3_chunk;
    double *tot = nullptr;
    const bool f32 = h->dtype == CB_F32, mixed = h->dtype == CB_MIXED, narrow = f32 || mixed;
    const size_t per_bucket = (expm_only || narrow) ? 0 : (size_t)Bl * LL;   // Ct / Gt / T exist for the loss only
    const size_t per_bucket32 = narrow ? (size_t)Bl * LL : 0;
    bool ok = dev_alloc(h, &h->Ct, mixed ? (size_t)Bl * LL : per_bucket) == CB_OK && dev_alloc(h, &tot, SS) == CB_OK &&
              dev_alloc(h, &h->Ct32, f32 ? per_bucket32 : 0) == CB_OK && dev_alloc(h, &h->Gt32, per_bucket32) == CB_OK &&
              dev_alloc(h, &h->T32, per_bucket32) == CB_OK && dev_alloc(h, &h->Uf, narrow ? LL : 0) == CB_OK &&
              dev_alloc(h, &h->Utf, narrow ? LL : 0) == CB_OK && dev_alloc(h, &h->Af, narrow ? LL : 0) == CB_OK &&
              dev_alloc(h, &h->Ff, narrow ? (size_t)B * h->LD : 0) == CB_OK &&
              dev_alloc(h, &h->A, LL) == CB_OK && dev_alloc(h, &h->dsq, h->LD) == CB_OK &&
              dev_alloc(h, &h->Gc, LL) == CB_OK && dev_alloc(h, &h->Vc, LL) == CB_OK &&
              dev_alloc(h, &h->Gc2, LL) == CB_OK && dev_alloc(h, &h->gx, 12 * LL + (size_t)h->LD + 16 + 3 * 256 + 8) == CB_OK &&
              dev_alloc(h, &h->U, LL) == CB_OK && dev_alloc(h, &h->lam, h->LD) == CB_OK &&
              dev_alloc(h, &h->sigma, 8) == CB_OK && dev_alloc(h, &h->off_bits, 64) == CB_OK &&
              dev_alloc(h, &h->F, (size_t)B * h->LD) == CB_OK &&
              dev_alloc(h, &h->E, (size_t)B * h->LD) == CB_OK &&
              dev_alloc(h, &h->H, (size_t)B * h->LD) == CB_OK &&
              dev_alloc(h, &h->Gt, per_bucket) == CB_OK &&
              dev_alloc(h, &h->T, per_bucket) == CB_OK &&
              dev_alloc(h, &h->Mt, LL) == CB_OK && dev_alloc(h, &h->X, LL) == CB_OK &&
              dev_alloc(h, &h->loss_part, (size_t)B * tiles) == CB_OK &&
              dev_alloc(h, &h->Yk, expm_only ? 0 : (1 + CB_PHI_TERMS) * LL) == CB_OK &&
              dev_alloc(h, &h->Lk, expm_only ? 0 : (1 + CB_PHI_TERMS) * LL) == CB_OK;
    if (ok && !expm_only && !f32)   // the time basis (tbasis.hip.h): interpolation matrices, bucket kinds, virtual branch lengths
      for (int k = 0; k < 2 && ok; ++k)
        ok = dev_alloc(h, &h->tb_Ls[k], (size_t)Bl * CB_TB_RS_MAX) == CB_OK && dev_alloc(h, &h->tb_Lg[k], (size_t)Bl * CB_TB_RG_MAX) == CB_OK &&
             dev_alloc(h, &h->tb_tf[k], Bl) == CB_OK && dev_alloc(h, &h->tb_tg[k], Bl) == CB_OK;
    if (!ok) {
      free_tmp();
      return cleanup(CB_ENOMEM);
    }
    // Everything an evaluation will need is allocated HERE, not on first use: a hipHostMalloc in the middle of an epoch (the
    // 64 bytes the first-order sweeps publish their statistics to; found with CB_TRACE_SLOW) blocks the host for ~80 ms --
    // and when its first use fell into a timed region, one bench run in five read 4.5 ms per epoch instead of 1.3.
    {
      void *q = nullptr;
      if (hipHostMalloc(&q, 8 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
        h->poll = (unsigned long long *)q;
        memset(q, 0, 8 * sizeof(unsigned long long));
      } else {
        (void)hipGetLastError();   // (the sweeps then read their statistics through the stream)
      }
    }
    if (!expm_only) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
      h->bank_slots = 4 * std::max(cus, 1);
      h->bank_claims = (h->bank_slots + LG_NQ - 1) / LG_NQ;   // reserved first tickets per queue (k123_bank)
      if (dev_alloc(h, &h->bank_queue, (size_t)LG_NQ + 2 * (size_t)B + (size_t)LG_NQ * h->bank_claims) != CB_OK ||
          dev_alloc(h, &h->bank_args, sizeof(K123Args<double, double>)) != CB_OK) {
        free_tmp();
        return cleanup(CB_ENOMEM);
      }
    }
    if (!expm_only) {
    hipLaunchKernelGGL(prep_counts_large_tot, dim3((unsigned)((SS + 255) / 256)), dim3(256), 0,
                       h->stream, S, B, Cdev, tot);
    hipLaunchKernelGGL(prep_counts_large_fin, dim3(1), dim3(256), 0, h->stream, S, tot, h->n_dev,
                       h->inv_n, h->ones, h->dirsum);
    const int nt32 = (h->LD + 31) / 32;
    if (f32) hipLaunchKernelGGL(lg_transpose_pad<float>, dim3(nt32, nt32, Bl), dim3(32, 8), 0, h->stream, S, h->LD,
                                Cdev, h->Ct32, src_idx);
    else hipLaunchKernelGGL(lg_transpose_pad<double>, dim3(nt32, nt32, Bl), dim3(32, 8), 0, h->stream, S, h->LD,
                            Cdev, h->Ct, src_idx);
    {
      int *flag = reinterpret_cast<int *>(h->status);  // [L] ints, unused by the large path
      (void)hipMemsetAsync(flag, 0, sizeof(int), h->stream);
      if (f32) hipLaunchKernelGGL(lg_sym_check<float>, dim3(nt32, nt32, Bl), dim3(32, 32), 0, h->stream, h->LD, h->Ct32, flag);
      else hipLaunchKernelGGL(lg_sym_check<double>, dim3(nt32, nt32, Bl), dim3(32, 32), 0, h->stream, h->LD, h->Ct, flag);
      int hf = 1;
      if (hipMemcpyAsync(&hf, flag, sizeof hf, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
          hipStreamSynchronize(h->stream) == hipSuccess)
        h->sym_counts = hf == 0 && !cb_test_hook("CB_NO_SYM");
    }
    }
  }
  h->n_host.resize(L);
  hipError_t e = hipMemcpyAsync(h->n_host.data(), h->n_dev, L * sizeof(double), hipMemcpyDeviceToHost,
                                h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (Ctmp) (void)hipFree(Ctmp);
  if (e != hipSuccess) return cleanup(fail(CB_EHIP, "cb_create: upload failed: %s", hipGetErrorString(e)));
  e = hipGetLastError();
  if (e != hipSuccess) return cleanup(fail(CB_EHIP, "cb_create: kernel failed: %s", hipGetErrorString(e)));
  for (int l = 0; l < L; ++l)
    if (!(h->n_host[l] > 0.0) || !std::isfinite(h->n_host[l]))
      return cleanup(fail(CB_ENUMERIC, "cb_create: site %d has total count %g", l, h->n_host[l]));
  *out = h;
  return CB_OK;
#undef TRY_ALLOC
#undef TRY_HIP
}

extern "C" void cb_destroy(cb_handle h) {
  if (!h) return;
  if (h->tb_next_pending) (void)h->tb_next.get();   // (a helper thread still building the next time basis)
  (void)hipSetDevice(h->dev);
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  for (hipEvent_t e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev2)
    if (e) (void)hipEventDestroy(e);
  for (void *p : h->allocs) (void)hipFree(p);
  for (double *p : h->ws_ptr)
    if (p) (void)hipFree(p);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->poll) (void)hipHostFree(h->poll);
  if (h->epin) (void)hipHostFree(h->epin);
  for (hipStream_t x : h->xstream)
    if (x) {
      (void)hipStreamSynchronize(x);
      (void)hipStreamDestroy(x);
    }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (hipEvent_t e : h->ev_join)
    if (e) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}
