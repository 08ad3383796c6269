// Timing of the single-matrix 400^3 product (sg_gemm) in several shapes, 400 dependent launches each:
// the eigensolver issues ~30 of these per epoch.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
#include "../../cherryml_amd/csrc/large_bank.hip.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int NW, int UU, int NJ, int PAIR = 1>
static float run(hipStream_t s, K4Args a, int n) {
  const unsigned nwg = (unsigned)((a.LD / 16) * ((a.LD + 16 * NJ - 1) / (16 * NJ)));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((sg_gemm<NW, UU, NJ>), dim3(nwg, PAIR), dim3(NW * 64), 0, s, a, a, 0, 0.0, 0.0);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((sg_gemm<NW, UU, NJ>), dim3(nwg, PAIR), dim3(NW * 64), 0, s, a, a, 0, 0.0, 0.0);
  (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("sg_gemm<NW=%2d, UU=%d, NJ=%d> x%d: %4u workgroups of %4d threads: %.2f us per launch\n", NW, UU, NJ, PAIR, nwg, NW * 64, ms * 1e3 / n);
  return ms;
}
int main() {
  const int LD = 400; const size_t LL = (size_t)LD * LD;
  std::vector<double> h(LL);
  for (size_t i = 0; i < LL; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  double *A, *B, *O; CK(hipMalloc(&A, LL * 8)); CK(hipMalloc(&B, LL * 8)); CK(hipMalloc(&O, LL * 8));
  CK(hipMemcpy(A, h.data(), LL * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), LL * 8, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  K4Args a{LD, LD, A, B, O, nullptr, nullptr, nullptr};
  const int n = 400;
  run<8, 4, 5, 2>(s, a, n); run<8, 4, 3, 2>(s, a, n);
  run<4, 4, 5>(s, a, n); run<8, 4, 5>(s, a, n); run<8, 7, 5>(s, a, n); run<16, 4, 5>(s, a, n); run<16, 7, 5>(s, a, n);
  run<8, 4, 3>(s, a, n); run<16, 7, 3>(s, a, n); run<8, 7, 2>(s, a, n); run<16, 7, 2>(s, a, n); run<4, 7, 1>(s, a, n); run<8, 7, 1>(s, a, n);
  return 0;
}
