#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, int spin) {
  extern __shared__ double lds[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // keep the wave alive a bit so that several workgroups co-reside
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 0.5;
  lds[threadIdx.x] = x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw | (lds[threadIdx.x ^ 1] > 1e300 ? 1u << 31 : 0);
}
int main(int argc, char **argv) {
  int threads = argc > 1 ? atoi(argv[1]) : 320;
  int nb = 2048;
  unsigned *d; hipMalloc(&d, nb * 16 * 4); hipMemset(d, 0xff, nb * 16 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 40960, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 16); hipMemcpy(h.data(), d, nb * 16 * 4, hipMemcpyDeviceToHost);
  int nw = threads / 64;
  // histogram of SIMD patterns per workgroup
  int pat[256] = {0};
  long simd_count[4] = {0, 0, 0, 0};
  for (int b = 0; b < nb; ++b) {
    int key = 0;
    for (int w = 0; w < nw; ++w) { int s = (h[b * 16 + w] >> 4) & 3; simd_count[s]++; if (w < 4) key = key * 4 + s; }
    pat[key]++;
  }
  printf("threads %d: waves per SIMD over all WGs: %ld %ld %ld %ld\n", threads, simd_count[0], simd_count[1], simd_count[2], simd_count[3]);
  for (int k = 0; k < 256; ++k) if (pat[k] > 20) printf("  first-4-wave SIMD pattern %d%d%d%d : %d WGs\n", (k >> 6) & 3, (k >> 4) & 3, (k >> 2) & 3, k & 3, pat[k]);
  for (int b = 0; b < 6; ++b) { printf("  WG %d:", b); for (int w = 0; w < nw; ++w) printf(" simd%u/cu%u", (h[b*16+w] >> 4) & 3, (h[b*16+w] >> 8) & 15); printf("\n"); }
  return 0;
}
