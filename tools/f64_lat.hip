// f64_lat.hip -- v_fma_f64 issue cadence and dependent-issue latency on gfx950: throughput of one wave per SIMD (and two)
// as a function of the number of independent accumulation chains.
//   hipcc -O3 --offload-arch=gfx950 tools/f64_lat.hip -o tools/f64_lat && tools/f64_lat
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int C>
__global__ void __launch_bounds__(256) chains(double* out, int iters, double a0, double b0) {
  double x[C];
#pragma unroll
  for (int k = 0; k < C; ++k) x[k] = threadIdx.x * 1e-9 + k;
  const double a = a0;
  double b = b0 + threadIdx.x * 1e-12;      // VGPR addend, scalar multiplicand (as in the psi2 sweep)
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 32 / C; ++r) {
#pragma unroll
      for (int k = 0; k < C; ++k) x[k] = fma(x[k], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < C; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C>
static void run(double* out, int cus, int wpc, int mhz) {
  const int blocks = cus * wpc, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  chains<C><<<blocks, 256>>>(out, 100, 1.0, 1.0); hipDeviceSynchronize();
  hipEventRecord(e0); chains<C><<<blocks, 256>>>(out, iters, 0.999, 0.001); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fmas_per_wave = (double)iters * 32;
  const double cyc = ms * 1e-3 * mhz * 1e6;
  printf("chains %2d  waves/SIMD %d: %.3f ms  %.2f cycles per FMA per wave  (%.1f TFLOP/s)\n", C, wpc, ms, cyc / fmas_per_wave,
         (double)blocks * 256 * fmas_per_wave * 2 / ms * 1e-9);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, mhz = p.clockRate / 1000;
  printf("device %s CUs %d clock %d MHz\n", p.name, cus, mhz);
  double* out; hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    run<1>(out, cus, wpc, mhz); run<2>(out, cus, wpc, mhz); run<4>(out, cus, wpc, mhz); run<8>(out, cus, wpc, mhz);
    run<16>(out, cus, wpc, mhz); run<32>(out, cus, wpc, mhz);
  }
  return 0;
}
