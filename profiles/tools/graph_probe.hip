// Round 4: does a hipGraph of dependent tiny kernels run them closer together than the same launches on a stream?
// 50 dependent launches (the planned eigensolve's count) of (a) a kernel that reads one device word and returns, (b) a kernel that
// streams 160000 doubles; stream launches vs one graph launch; GPU time per kernel from events around the batch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_skip(const unsigned long long *w) { if (*w != 0ull) return; }
__global__ void k_stream(const unsigned long long *w, const double *src, double *dst, int n) {
  if (*w != 0ull) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i] + 1.0;
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long *w; hipMalloc(&w, 64); hipMemset(w, 0, 64);
  double *d; hipMalloc(&d, 2 * 160000 * 8); hipMemset(d, 0, 2 * 160000 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int CH = 50, REP = 200;
  for (int kind = 0; kind < 2; ++kind) {
    auto one = [&](hipStream_t st) {
      if (kind == 0) hipLaunchKernelGGL(k_skip, dim3(625), dim3(256), 0, st, w);
      else hipLaunchKernelGGL(k_stream, dim3(625), dim3(256), 0, st, w, d, d + 160000, 160000);
    };
    // stream
    for (int i = 0; i < CH; ++i) one(s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < REP; ++r) for (int i = 0; i < CH; ++i) one(s);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: stream launches  %.2f us per kernel\n", kind ? "stream 160000 doubles" : "read word, return  ", ms * 1e3 / (CH * REP));
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < CH; ++i) one(s);
    hipStreamEndCapture(s, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("%s: graph of %d       %.2f us per kernel\n", kind ? "stream 160000 doubles" : "read word, return  ", CH, ms * 1e3 / (CH * REP));
  }
  return 0;
}
