// lds_atomic_probe.hip -- throughput of fp64 LDS atomics (ds_add_f64 without return) on gfx950 against plain 64-bit LDS
// writes, as a function of the number of active lanes and of the number of waves per CU that issue them.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_atomic_probe.hip -o tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(512) probe(double* out, int iters, int active, int workers) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 0.0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x = 1e-9 * threadIdx.x;
  if (wave < workers && lane < active) {
    double* p = lds + wave * 1024 + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (MODE == 0) atomicAdd(p + 64 * r, x);          // ds_add_f64
        else if (MODE == 1) p[64 * r] = x;                // ds_write_b64
        else x += atomicAdd(p + 64 * r, x);               // ds_add_rtn_f64
        x += 1e-12;
      }
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = lds[threadIdx.x] + x;
}

template <int MODE>
static void run(const char* name, double* out, int cus, int active, int workers) {
  const int iters = 20000;
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  probe<MODE><<<cus, 512, 100 * 1024>>>(out, 10, active, workers); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); probe<MODE><<<cus, 512, 100 * 1024>>>(out, iters, active, workers); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 8 * workers;      // wave instructions per CU
  printf("%-16s workers %d active lanes %2d: %.3f ms  %.1f ns per wave instruction per CU  (%.2f lane-ops per ns per CU)\n", name, workers,
         active, ms, ms * 1e6 / instr, instr * active / (ms * 1e6));
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; (void)hipMalloc(&out, sizeof(double) * 512 * cus);
  for (int workers : {8, 4, 1}) {
    for (int active : {64, 32, 16, 4, 1}) run<0>("ds_add_f64", out, cus, active, workers);
    run<1>("ds_write_b64", out, cus, 64, workers);
    run<2>("ds_add_rtn_f64", out, cus, 64, workers);
  }
  return 0;
}
