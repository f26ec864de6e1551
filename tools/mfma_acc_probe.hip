// mfma_acc_probe.hip -- does the rate of v_mfma_f64_16x16x4_f64 depend on how many accumulator tiles a wave rotates over, and on
// how many registers the wave holds?  NACC accumulator tiles, 48 matrix instructions per iteration, operands in registers.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_acc_probe.hip -o tools/mfma_acc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES, int PAD>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) probe(double* out, int iters) {
  const int t = threadIdx.x;
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a[4], b[3];
  for (int i = 0; i < 4; ++i) a[i] = 1e-3 * (t + i);
  for (int i = 0; i < 3; ++i) b[i] = 1e-3 * (t - i);
  double pad[PAD > 0 ? PAD : 1];                       // extra live registers
#pragma unroll
  for (int i = 0; i < PAD; ++i) pad[i] = out[t + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 48; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m & 3], b[m % 3], acc[m % NACC], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < PAD; ++i) asm volatile("" : "+v"(pad[i]));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < PAD; ++i) s += pad[i];
  out[blockIdx.x * 64 * WAVES + t] = s;
}

template <int NACC, int WAVES, int PAD>
static void run(double* out) {
  const int iters = 2000, cus = 256;
  probe<NACC, WAVES, PAD><<<cus, 64 * WAVES>>>(out, 10); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); probe<NACC, WAVES, PAD><<<cus, 64 * WAVES>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 48 * (WAVES / 4.0);           // matrix instructions per SIMD
  printf("accumulators %2d  waves/SIMD %d  extra live doubles %3d: %8.3f ms  %6.1f ns per instruction per SIMD  %5.1f TFLOP/s\n", NACC, WAVES / 4, PAD, ms,
         ms * 1e6 / n, 2048.0 * n * 4 * cus / (ms * 1e9));
}

int main() {
  double* out; hipMalloc(&out, 256 * 512 * sizeof(double) + 4096);
  hipMemset(out, 0, 256 * 512 * sizeof(double) + 4096);
  run<1, 4, 0>(out); run<3, 4, 0>(out); run<4, 4, 0>(out); run<12, 4, 0>(out);
  run<1, 8, 0>(out); run<3, 8, 0>(out); run<4, 8, 0>(out); run<12, 8, 0>(out);
  run<12, 8, 40>(out); run<12, 8, 70>(out); run<12, 4, 70>(out); run<4, 8, 100>(out);
  hipFree(out);
  return 0;
}
