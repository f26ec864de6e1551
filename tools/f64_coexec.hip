// f64_coexec.hip -- do v_mfma_f64_16x16x4_f64 and v_fma_f64 run concurrently on gfx950?  Times three loops with the same
// instruction counts: MFMA only, FMA only, and both interleaved (independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 1 = mfma, 2 = fma, 3 = both
__global__ void __launch_bounds__(256) k(double* out, int iters, double a0, double b0) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0;
  double x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-9 + j;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; ++i) {
    if (MODE & 1) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    }
    if (MODE & 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = fma(x[j], a0, b0);
    }
  }
  double s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> float run(double* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10, 0.999, 0.001); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters, 0.999, 0.001); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  double* out; hipMalloc(&out, sizeof(double) * 256 * p.multiProcessorCount * 8);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    const int blocks = p.multiProcessorCount * wpc, iters = 20000;
    float m = run<1>(out, blocks, iters), f = run<2>(out, blocks, iters), b = run<3>(out, blocks, iters);
    printf("%d waves/SIMD: 2 MFMA per iter %.3f ms | 32 FMA per iter %.3f ms | both %.3f ms (sum %.3f, max %.3f)\n", wpc, m, f, b, m + f,
           m > f ? m : f);
  }
  return 0;
}
