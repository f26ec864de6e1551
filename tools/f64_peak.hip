// f64_peak.hip -- measures the fp64 MFMA (v_mfma_f64_16x16x4_f64) and fp64 VALU (v_fma_f64) peaks of the GPU; these are
// the roofline denominators bench.py uses for the psi kernels (the local microarchitecture guide lists no fp64 peak).
//   hipcc -O3 --offload-arch=gfx950 tools/f64_peak.hip -o tools/f64_peak && tools/f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mfma_kernel(double* out, int iters, double a0, double b0) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double a0, double b0) {
  double x[8];
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1e-9 + k;
  const double a = a0, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = fma(x[k], a, b);
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) exp_kernel(double* out, int iters, double a0) {
  double x[4];
  for (int k = 0; k < 4; ++k) x[k] = -1.0 - threadIdx.x * 1e-3 - k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = exp(x[k]) - a0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s CUs %d clock %d MHz\n", p.name, cus, p.clockRate / 1000);
  double* out; hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 8; wpc *= 2) {             // 4, 8, 16, 32 waves per CU
    const int blocks = cus * wpc, iters = 20000;
    float ms;
    mfma_kernel<<<blocks, 256>>>(out, 100, 1.0, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); mfma_kernel<<<blocks, 256>>>(out, iters, 1.0, 0.5); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 4 * (2.0 * 16 * 16 * 4);
    printf("mfma_f64_16x16x4  %d waves/CU: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at %d MHz)\n", 4 * wpc, ms,
           flops / ms * 1e-9, ms * 1e-3 * p.clockRate * 1e3 / (iters * 4.0 * wpc), p.clockRate / 1000);
    fma_kernel<<<blocks, 256>>>(out, 100, 1.0, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); fma_kernel<<<blocks, 256>>>(out, iters * 4, 0.999, 0.001); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    flops = (double)blocks * 256 * (iters * 4.0) * 8 * 2;
    printf("v_fma_f64         %d waves/CU: %.3f ms  %.2f TFLOP/s\n", 4 * wpc, ms, flops / ms * 1e-9);
    exp_kernel<<<blocks, 256>>>(out, 10, 0.0); hipDeviceSynchronize();
    hipEventRecord(e0); exp_kernel<<<blocks, 256>>>(out, iters / 4, 1.5); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double ex = (double)blocks * 256 * (iters / 4.0) * 4;
    printf("exp(double)       %d waves/CU: %.3f ms  %.2f Gexp/s\n", 4 * wpc, ms, ex / ms * 1e-6);
  }
  return 0;
}
