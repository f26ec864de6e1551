// barrier_probe.hip -- does a wave that waits at s_barrier slow down the other wave of its SIMD?
// One 512-thread workgroup per CU (8 waves, two per SIMD).  Waves 0-3 run a chain of v_fma_f64; waves 4-7 either run
// it too (mode 0), wait at the workgroup barrier (mode 1), have left the kernel (mode 2) or sleep-poll an LDS flag (mode 3).
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o tools/barrier_probe && tools/barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void __launch_bounds__(512) probe(double* out, long long* cyc, int iters, int mode, double a0, double b0) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) lds[0] = 0;
  __syncthreads();
  double x[4];
  for (int k = 0; k < 4; ++k) x[k] = threadIdx.x * 1e-9 + k;
  const bool worker = wave < 4 || mode == 0;
  long long t0 = __builtin_readcyclecounter();
  if (worker) {
    const double a = a0; const double b = b0 + threadIdx.x * 1e-12;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fma(x[k], a, b);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (mode == 2 && !worker) return;
  if (mode == 3) {
    if (worker) { if (threadIdx.x == 0) atomicExch(&lds[0], 1); }
    else {
      int spins = 0;
      while (atomicAdd(&lds[0], 0) == 0 && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(8); ++spins; }
    }
  } else if (mode != 2) {
    __syncthreads();
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; long long* cyc; hipMalloc(&out, sizeof(double) * 512 * cus); hipMalloc(&cyc, sizeof(long long) * cus);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 20000;
  static long long h[1024];
  const char* names[4] = {"both waves of a SIMD work", "partner waits at s_barrier", "partner has exited", "partner sleep-polls LDS"};
  // does the rate per CU depend on how many CUs work?  (a power or clock limit would show here)
  for (int grid : {cus, cus / 2, cus / 4, cus / 8, 8}) {
    for (int mode = 0; mode < 2; ++mode) {
      probe<<<grid, 512, 100 * 1024>>>(out, cyc, 100, mode, 1.0, 1.0); hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); probe<<<grid, 512, 100 * 1024>>>(out, cyc, iters, mode, 0.999, 0.001); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
      double c = 0; for (int i = 0; i < grid; ++i) c += (double)h[i] / grid;
      printf("grid %3d mode %d: %.3f ms  (%.2f ns per FMA per working wave, %.2f counter ticks; counter rate %.0f MHz)\n", grid, mode, ms,
             ms * 1e6 / ((double)iters * 32), c / ((double)iters * 32), c / (ms * 1e3));
    }
  }
  for (int mode = 0; mode < 4; ++mode) {
    probe<<<cus, 512, 100 * 1024>>>(out, cyc, 100, mode, 1.0, 1.0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); probe<<<cus, 512, 100 * 1024>>>(out, cyc, iters, mode, 0.999, 0.001); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, sizeof(long long) * cus, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < cus; ++i) c += (double)h[i] / cus;
    printf("mode %d (%s): %.3f ms, %.2f counter ticks per FMA per working wave\n", mode, names[mode], ms, c / ((double)iters * 32));
  }
  return 0;
}
