// The fused bank kernel k123_bank (large_bank.hip.h: K1 -> K2 -> K3 of the co-evolution epoch as one persistent launch) in a
// translation unit of its own, because it is compiled with -mllvm -disable-machine-licm (cherryml_amd/_build.py): the ticket
// loop around the three stage bodies is a loop to the compiler, and its machine-level loop-invariant code motion hoists the
// ~30 floating-point constants of the three epilogues (polynomial coefficients of log / the divided differences) in front of
// it, where they stay live through every stage: 26 spilled VGPRs, and a spill reload behind an epilogue's stores waits for
// those stores (vmcnt counts them in order) -- K3's epilogue took 41 us instead of 11.  Without that pass: no spill.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#define CB_BANK_FUSED_TU 1
#include "common.hip.h"
#include "large_bank.hip.h"
#include "cb_internal.hip.h"

// variant: 0 = float64 bank, 1 = CB_F32, 2 = CB_MIXED; kg: 1 = four-wave tiles (four workgroups per CU), 2 = eight-wave tiles
// (two K-groups, two workgroups per CU: large_bank.hip.h, lg4_gemm_tile); `args` = the argument block in device memory
// (written by lg_tables); stop: null, or the event that takes the launch's end time (the phase timer: handle_host.hip.h)
template <typename T1, typename TG, int KG>
static void launch(const void *args, int grid, hipStream_t stream, hipEvent_t stop) {
  const K123Args<T1, TG> *a = static_cast<const K123Args<T1, TG> *>(args);
  if (stop) hipExtLaunchKernelGGL((k123_bank<T1, TG, KG>), dim3(grid), dim3(LG4_THREADS * KG), 0, stream, nullptr, stop, 0, a);
  else hipLaunchKernelGGL((k123_bank<T1, TG, KG>), dim3(grid), dim3(LG4_THREADS * KG), 0, stream, a);
}
template <int KG>
static void launch_variant(int variant, const void *args, int grid, hipStream_t stream, hipEvent_t stop) {
  if (variant == 1) launch<float, float, KG>(args, grid, stream, stop);
  else if (variant == 2) launch<double, float, KG>(args, grid, stream, stop);
  else launch<double, double, KG>(args, grid, stream, stop);
}
int cb_launch_bank_fused(int variant, int kg, const void *args, int grid, hipStream_t stream, hipEvent_t stop) {
  if (kg == 2) launch_variant<2>(variant, args, grid, stream, stop);
  else launch_variant<1>(variant, args, grid, stream, stop);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

#ifdef CB_CLOCK_STAMP
// diagnostic build only: the stamps of the FUSED launch (this translation unit's copy of cb_clock_stamps)
extern "C" int cb_debug_clock_stamps_fused(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cb_clock_stamps), sizeof(unsigned long long) * 3 * 4096 * 6) == hipSuccess ? 0 : -1;
}
#endif
