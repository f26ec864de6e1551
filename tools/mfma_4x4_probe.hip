// mfma_4x4_probe.hip -- lane layout and rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks, one double per lane for
// A, B and D).  For a one-hot A (lane la) and B = lane index + 1, every non-zero D lane names the B lane it was multiplied with.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_4x4_probe.hip -o tools/mfma_4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void layout(double* out) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    const double a = (l == la) ? 1.0 : 0.0, b = (double)(l + 1);
    out[la * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  }
}
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) rate(double* out, int iters) {
  const int t = threadIdx.x;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double a = 1e-3 * t, b = 1e-3 * (t + 1);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m & 7], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 64 * WAVES + t] = s;
}
int main() {
  double* d; hipMalloc(&d, 256 * 512 * 8);
  static double h[64 * 64];
  layout<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d ->", la);
    for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0.0) printf("  D lane %2d x B lane %2d", l, (int)h[la * 64 + l] - 1);
    printf("\n");
  }
  for (int w = 0; w < 2; ++w) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (w == 0) { rate<4><<<256, 256>>>(d, 10); hipDeviceSynchronize(); hipEventRecord(e0); rate<4><<<256, 256>>>(d, iters); }
    else { rate<8><<<256, 512>>>(d, 10); hipDeviceSynchronize(); hipEventRecord(e0); rate<8><<<256, 512>>>(d, iters); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d wave(s) per SIMD: %.2f ns per 4x4x4_4b instruction per SIMD\n", w + 1, ms * 1e6 / ((double)iters * 32 * (w + 1)));
  }
  return 0;
}
