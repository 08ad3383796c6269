// How long does a chain of dependent (same-stream) kernel launches take per kernel on this box?
// empty kernels of several grid shapes: the floor under every small launch of the eigensolver.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void touch_kernel(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  double *d; hipMalloc(&d, 160000 * 8); hipMemset(d, 0, 160000 * 8);
  const int N = 2000;
  struct { int grid, block; } shapes[] = {{1, 64}, {25, 1024}, {125, 512}, {256, 256}, {625, 256}, {1024, 256}};
  for (auto sh : shapes) {
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.grid), dim3(sh.block), 0, s, nullptr);
    hipEventRecord(a, s);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.grid), dim3(sh.block), 0, s, nullptr);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("empty kernel grid %4d x %4d threads: %.2f us per launch\n", sh.grid, sh.block, ms * 1e3 / N);
  }
  hipEventRecord(a, s);
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(touch_kernel, dim3(625), dim3(256), 0, s, d, 160000);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("touch 160000 doubles (read+write 2.5 MB): %.2f us per launch\n", ms * 1e3 / N);
  return 0;
}
