// Issue rate of the two f64 MFMA shapes: cycles per instruction with N independent accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int N> __global__ void k16(double *out, int iters) {
  d4 acc[N]; for (int i = 0; i < N; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < N; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / ((double)iters * N);
}
template <int N> __global__ void k4(double *out, int iters) {
  double acc[N]; for (int i = 0; i < N; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < N; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / ((double)iters * N);
}
int main() {
  double *d; hipMalloc(&d, ((1 << 20) + 8) * 8);
  double r;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(K, N, W)                                                                                      \
  {                                                                                                       \
    hipLaunchKernelGGL((K<N>), dim3(1), dim3(64 * W), 0, 0, d, 1000);                                      \
    hipEventRecord(e0);                                                                                   \
    hipLaunchKernelGGL((K<N>), dim3(1), dim3(64 * W), 0, 0, d, 200000);                                    \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                          \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                           \
    hipMemcpy(&r, d + (1 << 20), 8, hipMemcpyDeviceToHost);                                               \
    printf(#K " chains %d, waves/WG %d: %.2f ns per instr per wave (memtime ticks %.2f)\\n", N, W, ms * 1e6 / (200000.0 * N), r); \
  }
  RUN(k16, 1, 1) RUN(k16, 4, 1) RUN(k16, 8, 1) RUN(k16, 4, 4) RUN(k16, 4, 8)
  RUN(k4, 1, 1) RUN(k4, 4, 1) RUN(k4, 8, 1) RUN(k4, 16, 1) RUN(k4, 8, 4) RUN(k4, 8, 8)
  return 0;
}
