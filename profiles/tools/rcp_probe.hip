// accuracy of v_rcp_f64 and of one / two Newton steps on top of it, and of v_rsq/v_log-free alternatives (round 5, sp_bank's
// instruction diet):  hipcc --offload-arch=gfx950 -O3 rcp_probe.hip -o rcp_probe && ./rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y0 = __builtin_amdgcn_rcp(v);
  double y1 = fma(fma(-v, y0, 1.0), y0, y0);
  double y2 = fma(fma(-v, y1, 1.0), y1, y1);
  o[3 * i] = y0; o[3 * i + 1] = y1; o[3 * i + 2] = y2;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(3 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = std::ldexp(0.5 + u, (int)(s % 61) - 40);   // 2^-40 .. 2^20, the range of Pt entries and eigenvalue gaps
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost);
  double e[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j) {
      long double r = (long double)o[3 * i + j] * (long double)x[i] - 1.0L;
      e[j] = std::fmax(e[j], (double)fabsl(r));
    }
  printf("max |x * y - 1|: v_rcp_f64 %.3e, + 1 Newton step %.3e, + 2 Newton steps %.3e\n", e[0], e[1], e[2]);
  return 0;
}
