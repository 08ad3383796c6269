// Round 4: what does a PREDICATED launch cost?  Chains of dependent same-stream launches of (a) an empty kernel, (b) a kernel with a
// 200-byte argument block, (c) a kernel that reads one device word and returns (the "skip" of eigh_planned.hip.h), (d) the same where the
// PREVIOUS kernel wrote that word (the realistic case: a cross-XCD miss), (e) reads the word and then streams 16 doubles per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double *p[20]; int a[10]; };
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b) { if (b.a[0] == 12345 && threadIdx.x == 9999) *b.p[0] = 1; }
__global__ void k_skip(const unsigned long long *w) { if (*w != 0ull) return; }
__global__ void k_write_then(unsigned long long *w) { if (blockIdx.x == 0 && threadIdx.x == 0) w[1] = w[1] + 1; if (w[0] != 0ull) return; }
__global__ void k_read_stream(const unsigned long long *w, const double *src, double *dst, int n) {
  if (*w != 0ull) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i] + 1.0;
}
__global__ void k_stream(const double *src, double *dst, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i] + 1.0;
}
template <class F> static void run(const char *name, hipStream_t s, F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int N = 2000;
  for (int i = 0; i < 50; ++i) f();
  hipEventRecord(a, s);
  for (int i = 0; i < N; ++i) f();
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-60s %.2f us per launch\n", name, ms * 1e3 / N);
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long *w; hipMalloc(&w, 64); hipMemset(w, 0, 64);
  double *d; hipMalloc(&d, 2 * 160000 * 8); hipMemset(d, 0, 2 * 160000 * 8);
  Big b{}; 
  for (int g : {1, 625}) for (int t : {256, 512}) {
    char nm[128];
    snprintf(nm, sizeof nm, "empty %dx%d", g, t); run(nm, s, [&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(t), 0, s, nullptr); });
    snprintf(nm, sizeof nm, "200-byte args %dx%d", g, t); run(nm, s, [&] { hipLaunchKernelGGL(k_big, dim3(g), dim3(t), 0, s, b); });
    snprintf(nm, sizeof nm, "read word, return %dx%d", g, t); run(nm, s, [&] { hipLaunchKernelGGL(k_skip, dim3(g), dim3(t), 0, s, w); });
    snprintf(nm, sizeof nm, "prev kernel wrote next to the word %dx%d", g, t); run(nm, s, [&] { hipLaunchKernelGGL(k_write_then, dim3(g), dim3(t), 0, s, w); });
  }
  run("stream 160000 doubles 625x256", s, [&] { hipLaunchKernelGGL(k_stream, dim3(625), dim3(256), 0, s, d, d + 160000, 160000); });
  run("read word then stream 625x256", s, [&] { hipLaunchKernelGGL(k_read_stream, dim3(625), dim3(256), 0, s, w, d, d + 160000, 160000); });
  return 0;
}
