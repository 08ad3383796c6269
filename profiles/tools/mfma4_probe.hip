// Layout probe for v_mfma_f64_4x4x4f64 (4 blocks): which lane holds which A / B / D element.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double *out) {
  const int l = threadIdx.x;
  // test 1: A = one-hot at lane la (value 1), B = lane index + 1 -> D shows where products land
  for (int la = 0; la < 64; ++la) {
    double a = (l == la) ? 1.0 : 0.0;
    double b = 100.0 + l;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[la * 64 + l] = d;
  }
}
int main() {
  double *d; hipMalloc(&d, 64 * 64 * 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  static double h[64 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // For A one-hot at lane la: D[lane] nonzero entries tell: D lane set = {lanes with same block and row i(la)},
  // and the value (100 + lb) tells which B lane (k(la), j(lane)) was used.
  for (int la : {0, 1, 2, 3, 4, 5, 8, 12, 15, 16, 17, 20, 37, 63}) {
    printf("A one-hot lane %2d -> ", la);
    for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0.0) printf("D[lane %2d]=B[lane %2d] ", l, (int)(h[la * 64 + l] - 100.0));
    printf("\n");
  }
  return 0;
}
