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

// the fused bank launch k123_bank (cb_bank_fused.hip): variant 0 = float64, 1 = CB_F32, 2 = CB_MIXED; kg = 1 (four-wave
// tiles) or 2 (eight-wave tiles); `args` = the argument block in device memory; stop = null or the event that takes the
// launch's end time.  Returns 0 or -1.
int cb_launch_bank_fused(int variant, int kg, const void *args, int grid, hipStream_t stream, hipEvent_t stop);

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
