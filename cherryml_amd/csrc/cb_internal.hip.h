// Shared by the translation units of libcherrybank: error reporting and small host-side helpers.
// (cherrybank.hip: handle, bank entry points, trainers; cb_bank_fused.hip: the one-launch bank kernel; cb_counting.hip;
// cb_ble.hip; cb_likelihood.hip; cb_host_io.hip.)
#pragma once
#include "../../include/cherrybank.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// sets the thread-local message cb_last_error() returns; returns `code` (defined in cherrybank.hip)
int cb_fail(int code, const char *fmt, ...);
#define fail cb_fail

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(e_ == hipErrorOutOfMemory ? CB_ENOMEM : CB_EHIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                    \
  } while (0)

// cb_internal_expm_bank flag: the matrix is the one of the previous call on this handle -- skip the eigensolve
// (the public cb_expm_bank masks it out)
#define CB_REUSE_EIGH 256
int cb_internal_expm_bank(cb_handle h, const double *Q, const double *pi, int flags, double *P);
// new branch lengths for a counts-free (CB_EXPM_ONLY) single-bank handle, B <= its creation B (cherrybank.hip);
// t_dev != NULL: the same values already on the device -- copied on the handle's stream, no host wait
int cb_internal_set_times(cb_handle h, const double *t_host, int B, const double *t_dev);
// the symmetric eigendecomposition a large (S > 32) handle's last reversible expm bank left on the device: A = D^1/2 Q D^-1/2
// [LD][LD], U [LD][LD] row-major (columns = eigenvectors), lam [LD], dsq = sqrt(pi) [LD] (pad 1), sigma = max |A_ii| (a scalar)
struct CbSpectral {
  int S = 0, LD = 0;
  const double *A = nullptr, *U = nullptr, *lam = nullptr, *dsq = nullptr, *sigma = nullptr;
};
int cb_internal_spectral(cb_handle h, CbSpectral *out);

// the fused bank launch k123_bank (cb_bank_fused.hip): variant 0 = float64, 1 = CB_F32, 2 = CB_MIXED; kg = 1 (four-wave
// tiles) or 2 (eight-wave tiles); `args` = the argument block in device memory; stop = null or the event that takes the
// launch's end time.  Returns 0 or -1.
int cb_launch_bank_fused(int variant, int kg, const void *args, int grid, hipStream_t stream, hipEvent_t stop);

// ---- the bank in a time basis (cb_tbasis.hip, tbasis.hip.h): host builder, spectral tables, the elementwise kernel ----
#define CB_TB_RS_MAX 24   // skeleton buckets of the short-branch forward family (psi), at most
#define CB_TB_RG_MAX 40   // skeleton buckets of the gradient family (e^{t mu}), at most
struct CbTimeBasisHost {
  int B = 0, ns = 0, nd = 0, ng = 0;   // live buckets; forward skeleton, direct (long-branch) buckets, gradient skeleton
  double rho_max = 0.0;                // the basis serves spectra inside [-rho_max, 0]
  double res_s = 0.0, res_g = 0.0;     // largest interpolation residuals on the sample grid
  std::vector<int> kind;               // [B]: -1 = expanded in the psi family, k >= 0 = direct bucket k
  std::vector<int> skel_s, direct, skel_g;   // live-bucket indices
  std::vector<double> tf, tg;          // branch lengths of the virtual buckets: [ns + nd], [ng]
  std::vector<double> Ls, Lg;          // [B][CB_TB_RS_MAX], [B][CB_TB_RG_MAX] (zero padded; Lg carries t_b / t_skeleton)
};
// false: the grid / range needs more skeleton buckets than the maxima (the caller keeps the per-bucket products)
bool cb_tb_build(int B, const double *t, double rho_max, CbTimeBasisHost &out);
struct CbTbEwArgs {
  int S, LD, B, ns, nd, ng;
  int nsmall;           // buckets 0 .. nsmall - 1 are expanded in the psi family, bucket b >= nsmall is direct bucket b - nsmall
  const double *Ct;     // [B][LD][LD] counts (transposed per bucket; symmetric banks only)
  const double *Psi;    // [ns + nd][LD][LD]: Psi_r of the forward skeleton, then P_b of the direct buckets
  const double *A;      // [LD][LD]
  const double *t;      // [B]
  const double *Ls, *Lg;
  double *Gh;           // [ng][LD][LD] out (float64 bank)
  float *Gh32;          // ... or, when not null, the same rounded to float32 (CB_MIXED)
  double *loss_part;    // [LD * LD / 256] out
  double inv_n;
  const unsigned long long *skip;   // device word: non-zero => return at once
};
int cb_tb_launch_tables(int LD, int ns, int nd, int ng, const double *tf, const double *tg, const double *lam, double *F, double *E,
                        double *H, const unsigned long long *skip, hipStream_t stream);
// LDS bytes tb_ew needs for a bank of B live buckets (its interpolation matrices live there); above CB_TB_LDS_MAX the bank keeps
// the per-bucket forms
#define CB_TB_LDS_MAX (144u << 10)
size_t cb_tb_ew_lds_bytes(int B, int ns, int ng);
bool cb_tb_supported(int B, int LD, int ns, int ng);   // (+ the buffer loads' offset range: B LD^2 below 2^28 doubles)
// raises tb_ew's dynamic-LDS limit on the CURRENT device for the instantiation (ns, ng) selects; cache: four words of the handle
int cb_tb_prepare_ew(int B, int ns, int ng, size_t *cache);
// (*nparts: the loss partials the launch writes)
int cb_tb_launch_ew(const CbTbEwArgs &a, hipStream_t stream, hipEvent_t stop, int *nparts);

// Test hooks (tests/, profiles/): environment variables that change WHICH kernels run or inject faults are honoured only
// when CB_TEST_HOOKS=1 is set as well -- a stray CB_NO_SYM in a user's shell must not change the path.  (CB_DEBUG and
// CB_TRACE_SLOW only log; CB_BANK_STREAMS is a documented opt-in.)
static inline const char *cb_test_hook(const char *name) {
  const char *on = getenv("CB_TEST_HOOKS");
  return (on && on[0] == '1') ? getenv(name) : nullptr;
}

// device buffers of one call of the per-family entry points (uploaded on the default stream, freed on return)
struct CbDevBufs {
  std::vector<void *> ptrs;
  ~CbDevBufs() {
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
  }
  template <typename T>
  T *up(const T *host, size_t count, int &rc) {
    void *q = nullptr;
    if (rc != CB_OK) return nullptr;
    if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) {
      rc = fail(CB_ENOMEM, "device allocation failed");
      return nullptr;
    }
    ptrs.push_back(q);
    if (host && count && hipMemcpyAsync(q, host, count * sizeof(T), hipMemcpyHostToDevice, 0) != hipSuccess)
      rc = fail(CB_EHIP, "upload failed");
    return static_cast<T *>(q);
  }
};
