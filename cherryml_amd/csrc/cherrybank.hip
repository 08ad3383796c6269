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

// ------------------------------------------------------------------ handle
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
  int k3_chunk = 0, k3_nchunks = 0;
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
enum { EV_START = 0, EV_EIGH, EV_K1, EV_K2, EV_K3, EV_K4, EV_SMALL, EV_END };
static void mark(cb_bank *h, int which) {
  if (!h->profile) return;
  if (!h->ev[which]) (void)hipEventCreate(&h->ev[which]);
  (void)hipEventRecord(h->ev[which], h->stream);
  h->ev_rec[which] = true;
}

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
        h->sym_counts = hf == 0 && !getenv("CB_NO_SYM");
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
              dev_alloc(h, &h->loss_part, (size_t)B * tiles) == CB_OK;
    if (!ok) {
      free_tmp();
      return cleanup(CB_ENOMEM);
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
        h->sym_counts = hf == 0 && !getenv("CB_NO_SYM");
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
    hipLaunchKernelGGL(kern, dim3(h->L), dim3(NW * 64), lds, h->stream, a);                     \
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

// h->A (padded, symmetric) and h->dsq are filled.  Output: dQ (S x S, dQ = D^1/2 dA D^-1/2) when
// `dA_padded` is false, else dL/dA itself as a padded LD x LD matrix.
static int large_eval(cb_bank *h, bool normalize, double *lossd, double *out, bool dA_padded, double *Pd,
                      bool reuse_eigh = false) {
  const int S = h->S, LD = h->LD;
  const int B = Pd ? h->B : h->Bl;                 // the loss visits live buckets only
  const double *tb = Pd ? h->t : h->t_live;
  const size_t LL = (size_t)LD * LD;
  double *dQd = out;
  int rc = CB_OK;
  if (!(reuse_eigh && h->have_prev)) rc = large_eigh(h, true);   // reuse: same matrix as the previous call (CB_REUSE_EIGH)
  if (rc != CB_OK) return rc;
  mark(h, EV_EIGH);
  hipLaunchKernelGGL(lg_tables, dim3((unsigned)(((size_t)B * LD + 255) / 256)), dim3(256), 0,
                     h->stream, LD, B, tb, h->lam, h->sigma, h->F, h->E, h->H);
  const int tm = (LD + LG_TM - 1) / LG_TM, tn = (LD + LG_TN - 1) / LG_TN, tiles = tm * tn;
  const double inv_n = normalize ? 1.0 / (h->comm ? h->n_global[0] : h->n_host[0]) : 1.0;
  const int tiles_k1 = tn * (tn + 1) / 2;  // Pt is symmetric: upper-triangular tiles only
  const int tiles_k3 = h->sym_counts ? tiles_k1 : tiles;
  // float32 bank (cb_create(dtype = CB_F32)): the loss / gradient products run on the f32 MFMA from f32
  // copies of this epoch's U, U^T, A and F; cb_expm_bank (Pd) always takes the float64 kernels
  // CB_MIXED: P_b, the loss and G_b in float64 (the O(t^2) entries of P_b keep their relative accuracy),
  // G_b rounded to float32 once, the two contractions on the float32 MFMA
  const bool f32 = h->dtype == CB_F32 && !Pd, mixed = h->dtype == CB_MIXED && !Pd;
  static const int n_parts = getenv("CB_BANK_STREAMS") ? std::min(4, std::max(1, atoi(getenv("CB_BANK_STREAMS")))) : 1;
  if (f32 || mixed)
    hipLaunchKernelGGL(lg_cast_f32, dim3((unsigned)((std::max(LL, (size_t)B * LD) + 255) / 256)), dim3(256), 0, h->stream, LL,
                       (size_t)B * LD, h->U, h->Vc, h->A, h->F, h->Uf, h->Utf, h->Af, h->Ff);
  mark(h, EV_END);  // (re-used as "before K1" marker)
  if (f32) {
    K1Args<float> k1{S, LD, B, h->Utf, h->Af, tb, h->Ff, h->sigma, h->Ct32, h->Gt32, h->loss_part, inv_n, h->dsq, nullptr};
    hipLaunchKernelGGL(k1_pt_loss_gt<float>, dim3(tiles_k1 * B), dim3(LG4_THREADS), 0, h->stream, k1);
  } else if (mixed) {
    K1Args<double, float> k1{S, LD, B, h->Vc, h->A, tb, h->F, h->sigma, h->Ct, h->Gt32, h->loss_part, inv_n, h->dsq, nullptr};
    hipLaunchKernelGGL((k1_pt_loss_gt<double, float>), dim3(tiles_k1 * B), dim3(LG4_THREADS), 0, h->stream, k1);
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
      hipLaunchKernelGGL((k1_pt_loss_gt<double, double, false>), dim3(tiles_k1 * Bs), dim3(LG4_THREADS), 0, st, k1);
    }
    for (int p = 0; p < n_parts; ++p) {
      int b0, Bs; hipStream_t st;
      part_of(p, b0, Bs, st);
      K2Args<double> k2{LD, h->Gt + b0 * LL, h->U, h->T + b0 * LL};
      hipLaunchKernelGGL(k2_t_eq_g_u<double>, dim3(tiles * Bs), dim3(LG4_THREADS), 0, st, k2);
    }
    for (int p = 0; p < n_parts; ++p) {
      int b0, Bs; hipStream_t st;
      part_of(p, b0, Bs, st);
      K3Args<double> k3{LD, Bs, h->T + b0 * LL, h->U, tb + b0, h->lam, h->E + (size_t)b0 * LD, h->H + (size_t)b0 * LD,
                        h->Gt + b0 * LL, h->sym_counts ? 1 : 0};
      hipLaunchKernelGGL(k3_w_phi<double>, dim3(tiles_k3 * Bs), dim3(LG4_THREADS), 0, st, k3);
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
    K1Args<double> k1{S, LD, B, h->Vc, h->A, tb, h->F, h->sigma, h->Ct, h->Gt, h->loss_part, inv_n, h->dsq, Pd};
    if (Pd) hipLaunchKernelGGL((k1_pt_loss_gt<double, double, true>), dim3(tiles_k1 * B), dim3(LG4_THREADS), 0, h->stream, k1);
    else hipLaunchKernelGGL((k1_pt_loss_gt<double, double, false>), dim3(tiles_k1 * B), dim3(LG4_THREADS), 0, h->stream, k1);
  }
  mark(h, EV_K1);
  if (Pd) {
    HIP_TRY(hipGetLastError());
    return CB_OK;
  }
  hipLaunchKernelGGL(lg_finish_loss, dim3(1), dim3(256), 0, h->stream, h->loss_part, B * tiles_k1, S,
                     h->dsq, h->dirsum, inv_n, lossd);
  if (dQd) {
    if (f32 || mixed) {
      K2Args<float> k2{LD, h->Gt32, h->Uf, h->T32};
      hipLaunchKernelGGL(k2_t_eq_g_u<float>, dim3(tiles * B), dim3(LG4_THREADS), 0, h->stream, k2);
      mark(h, EV_K2);
      K3Args<float> k3{LD, B, h->T32, h->Uf, tb, h->lam, h->E, h->H, h->Gt32, h->sym_counts ? 1 : 0};
      hipLaunchKernelGGL(k3_w_phi<float>, dim3(tiles_k3 * B), dim3(LG4_THREADS), 0, h->stream, k3);
      mark(h, EV_K3);
      hipLaunchKernelGGL(k3_reduce<float>, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream,
                         h->Gt32, B, LL, h->Mt, h->sym_counts ? LD : 0);
    } else {
      K2Args<double> k2{LD, h->Gt, h->U, h->T};
      hipLaunchKernelGGL(k2_t_eq_g_u<double>, dim3(tiles * B), dim3(LG4_THREADS), 0, h->stream, k2);
      mark(h, EV_K2);
      K3Args<double> k3{LD, B, h->T, h->U, tb, h->lam, h->E, h->H, h->Gt, h->sym_counts ? 1 : 0};
      hipLaunchKernelGGL(k3_w_phi<double>, dim3(tiles_k3 * B), dim3(LG4_THREADS), 0, h->stream, k3);
      mark(h, EV_K3);
      hipLaunchKernelGGL(k3_reduce<double>, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, h->stream,
                         h->Gt, B, LL, h->Mt, h->sym_counts ? LD : 0);
    }
    K4Args k4a{S, LD, h->Mt, h->Vc, h->X, nullptr, nullptr, nullptr};
    launch_sg(h, k4a, 0);
    K4Args k4b{S, LD, h->Vc, h->X, dQd, dA_padded ? nullptr : h->dsq, nullptr, nullptr};
    launch_sg(h, k4b, 0);
    mark(h, EV_K4);
  }
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
  if (rc == CB_OK && getenv("CB_FAULT_INJECT") && atoi(getenv("CB_FAULT_INJECT")) == -1)
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
  uint64_t sig = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) sig = (sig ^ c[i]) * 1099511628211ull;
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
  auto release = [&]() { (void)hipStreamSynchronize(h->stream); };
  double *d_pi = nullptr, *d_up = nullptr, *d_mom = nullptr, *d_mask = nullptr, *d_loss = nullptr, *d_Qb = nullptr,
         *d_Ql = nullptr, *d_Qp = nullptr, *d_vec = nullptr;
  const size_t nmom = 2 * (S + nup);
  // fixed slots (an optional buffer keeps its number): a resumed call finds the state where the first call put it
  auto at = [&](int s, double **p, size_t n) -> bool { return ws_get(h, s, n, p); };
  bool ok = at(0, &d_pi, S) && at(1, &d_up, nup) && at(2, &d_mom, nmom) && at(3, &d_loss, E) && at(4, &d_Qb, SS) &&
            at(5, &d_Ql, SS) && (!mask || at(6, &d_mask, SS)) &&
            (!(Q_pow2 && n_pow2 > 0) || at(7, &d_Qp, std::max<size_t>(n_pow2, 16) * SS)) && at(8, &d_vec, (size_t)LD + S + 8);
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
    const size_t down_bytes = (S + nup + (size_t)E + (2 + (size_t)(d_Qp ? n_pow2 : 0)) * SS + 64) * sizeof(double);
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
  if (mask) TRYH(h2d_staged(h, d_mask, mask, SS * sizeof(double)));
  const double init_state[2] = {INFINITY, 0.0};
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
  a.loss_curve = d_loss; a.Q_last = d_Ql; a.Q_best = d_Qb; a.Q_pow2 = d_Qp;
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: copies enqueued after %.2f ms\n", now() - t_enter);
  if (!resume) TRYH(h2d_staged(h, a.state, init_state, sizeof init_state));
  TRYH(hipStreamSynchronize(h->stream));  // init_state is on this stack frame
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: synced after %.2f ms\n", now() - t_enter);
  if (h->profile) fold_pending(h);
  if (dbg) fprintf(stderr, "[cherrybank] large trainer: parameters uploaded after %.2f ms\n", now() - t_enter);
  double pow_b1 = resume ? h->tr_pow_b1 : 1.0, pow_b2 = resume ? h->tr_pow_b2 : 1.0;
  h->last_form = 4000;
  h->tr_epochs = 0;   // (set again when this call succeeds)
  // fault injection for the tests of the collective failure protocol: this rank's evaluation "fails" at that epoch
  const int fault_epoch = getenv("CB_FAULT_INJECT") ? atoi(getenv("CB_FAULT_INJECT")) : -1000;
  for (int e = 0; e < E && rc == CB_OK; ++e) {
    if (h->profile) {  // fold the epoch before the previous one (its events are long complete), then re-record that set
      swap_event_sets(h);
      fold_pending(h);
    }
    for (bool &b : h->ev_rec) b = false;
    hipLaunchKernelGGL(lt_pi, dim3(1), dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL(lt_build, dim3(LD), dim3(256), 0, h->stream, a, e0 + e);
    mark(h, EV_START);
    rc = large_eval(h, flags & CB_NORMALIZE, h->loss, h->Mt, true, nullptr);
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
    }
    if (h->profile) h->t_pending = true;
    pow_b1 *= a.beta1;
    pow_b2 *= a.beta2;
    hipLaunchKernelGGL(lt_gd, dim3((S + 3) / 4), dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL(lt_step_pi, dim3(1), dim3(256), 0, h->stream, a, e0 + e, 1.0 - pow_b1, std::sqrt(1.0 - pow_b2));
    hipLaunchKernelGGL(lt_step_up, dim3(S), dim3(256), 0, h->stream, a, 1.0 - pow_b1, std::sqrt(1.0 - pow_b2));
    if (hipGetLastError() != hipSuccess) rc = fail(CB_EHIP, "fused training launch failed");
  }
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
  TRYH(hipStreamSynchronize(h->stream));
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
         *d_Qb = nullptr, *d_Ql = nullptr, *d_Qp = nullptr;
  int slot = 0;
  auto alloc = [&](double **p, size_t n) -> bool { return ws_get(h, slot++, n, p); };
  auto release = [&]() { (void)hipStreamSynchronize(h->stream); };
  const size_t nmom = 2 * ((size_t)L * S + (size_t)L * nup);
  bool ok = alloc(&d_pi, (size_t)L * S) && alloc(&d_up, L * nup) && alloc(&d_mom, nmom) &&
            alloc(&d_loss, (size_t)E * L) && alloc(&d_Qb, L * SS) && alloc(&d_Ql, L * SS) &&
            (!mask || alloc(&d_mask, SS)) && (!(Q_pow2 && n_pow2 > 0) || alloc(&d_Qp, std::max<size_t>(n_pow2, 16) * SS));
  if (!ok) {
    release();
    return fail(CB_ENOMEM, "fused training: device allocation failed");
  }
  int rc = CB_OK;
#define TRYH(expr)                                                                  \
  if (rc == CB_OK && (expr) != hipSuccess) rc = fail(CB_EHIP, "%s failed", #expr)
  {
    const size_t up_bytes = ((size_t)L * S + L * nup + SS + 64) * sizeof(double);
    const size_t down_bytes = ((size_t)L * S + L * nup + (size_t)E * L + 2 * L * SS + (size_t)(d_Qp ? n_pow2 : 0) * SS + 64) * sizeof(double);
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
  if (rc == CB_OK) {
    TrainArgs a{};
    a.S = S; a.L = L; a.B = h->Bl; a.E = E; a.kind = kind; a.do_adam = do_adam; a.n_pow2 = d_Qp ? n_pow2 : 0;
    a.nlive = h->nlive; a.t = h->t_live; a.Ct = h->Ct; a.Cq = h->Cq; a.nq = h->nq; a.inv_n = (flags & CB_NORMALIZE) ? h->inv_n : h->ones; a.dirsum = h->dirsum;
    a.p_pi = d_pi; a.p_up = d_up;
    a.m_pi = d_mom; a.v_pi = d_mom + (size_t)L * S;
    a.m_up = d_mom + 2 * (size_t)L * S; a.v_up = a.m_up + L * nup;
    a.mask = d_mask; a.lr = lr; a.beta1 = 0.9; a.beta2 = 0.999; a.eps = 1e-8;
    a.loss_curve = d_loss; a.Q_best = d_Qb; a.Q_last = d_Ql; a.Q_pow2 = d_Qp;
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
  TRYH(hipStreamSynchronize(h->stream));
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

extern "C" int cb_train_siterm(cb_handle h, double *theta, double *Theta, int num_epochs, double lr,
                               int flags, double *res, double *loss_per_epoch_per_site) {
  if (!h || !theta || !Theta) return fail(CB_EINVAL, "cb_train_siterm: NULL argument");
  return run_fused_training(h, 1, theta, Theta, nullptr, num_epochs, lr, 1, flags | CB_NORMALIZE,
                            loss_per_epoch_per_site, res, nullptr, nullptr, 0);
}

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
extern "C" int cb_last_kernel_form(cb_handle h) { return h ? h->last_form : 0; }

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
      if (h->ev_rec[i] && i != EV_END) last = i;
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
    v[CB_T_K1] = span(EV_END, EV_K1);
    v[CB_T_K2] = span(EV_K1, EV_K2);
    v[CB_T_K3] = span(EV_K2, EV_K3);
    v[CB_T_K4] = span(EV_K3, EV_K4);
    v[CB_T_TOTAL] = span(EV_START, h->ev_rec[EV_K4] ? EV_K4 : EV_K1);
  } else {
    v[CB_T_SMALL] = span(EV_START, EV_SMALL);
    v[CB_T_TOTAL] = v[CB_T_SMALL];
  }
  return CB_OK;
}
