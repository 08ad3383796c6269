// What do the start / stop events of hipExtLaunchKernelGGL measure, and what does a launch with them cost between two
// dependent kernels compared with hipEventRecord?   hipcc --offload-arch=gfx950 -O2 ext_event_probe.hip -o ext_event_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = t0;
}
int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  unsigned long long *d;
  hipMalloc(&d, 64);
  hipEvent_t e[6];
  for (auto &x : e) hipEventCreate(&x);
  for (int it = 0; it < 3; ++it) {
    // A: 100 us, B: 50 us; e0 = start of A, e1 = stop of A, e2 = start of B, e3 = stop of B
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, e[0], e[1], 0, 10000ull, d);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, e[2], e[3], 0, 5000ull, d);
    hipStreamSynchronize(s);
    float a, b, c, g;
    hipEventElapsedTime(&a, e[0], e[1]);
    hipEventElapsedTime(&b, e[2], e[3]);
    hipEventElapsedTime(&c, e[0], e[3]);
    hipEventElapsedTime(&g, e[1], e[2]);
    printf("ext: start->stop A %.1f us, B %.1f us, startA->stopB %.1f us, stopA->startB %.1f us\n", a * 1e3, b * 1e3, c * 1e3, g * 1e3);
  }
  // cost between dependent kernels: N pairs of 20 us kernels with (a) nothing, (b) hipEventRecord between, (c) ext stop events
  const int N = 200;
  for (int mode = 0; mode < 4; ++mode) {
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, nullptr, e[i & 3], 0, 2000ull, d);
      else if (mode == 3) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, e[i & 3], nullptr, 0, 2000ull, d);
      else hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 2000ull, d);
      if (mode == 1) hipEventRecord(e[i & 3], s);
    }
    hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d (%s): %.2f us per 20 us kernel\n", mode, mode == 0 ? "plain" : mode == 1 ? "hipEventRecord after each" : mode == 2 ? "ext stop event on each" : "ext start event on each", us / N);
  }
  return 0;
}
