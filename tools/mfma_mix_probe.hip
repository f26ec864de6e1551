// mfma_mix_probe.hip -- what do ordinary vector instructions cost next to v_mfma_f64_16x16x4_f64 on one SIMD?
// Per group a wave issues 3 matrix instructions, NDP independent v_mul_f64, NI independent integer shift+add pairs and NA LDS
// fp64 adds (the NL LDS reads are volatile, each waited for: they time latency, not issue cost, and are not in the default runs).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_mix_probe.hip -o tools/mfma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NDP, int NI, int WAVES, int NM, int NL = 0, int NA = 0>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) probe(double* out, int iters) {
  const int t = threadIdx.x;
  __shared__ double sh[64 * WAVES * 4];
  for (int i = 0; i < 4; ++i) sh[t + 64 * WAVES * i] = 1e-6 * t;
  __syncthreads();
  const volatile double* shv = sh + ((t * 9) & (64 * WAVES - 1));
  double ld[NL > 0 ? NL : 1];
  for (int i = 0; i < NL; ++i) ld[i] = 0.0;
  d4 acc[3];
  for (int i = 0; i < 3; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a = 1e-3 * t, b[3] = {1e-3 * (t + 1), 1e-3 * (t + 2), 1e-3 * (t + 3)};
  double x[NDP > 0 ? NDP : 1];
  int n[NI > 0 ? NI : 1];
  for (int i = 0; i < NDP; ++i) x[i] = 1.0 + 1e-9 * (t + i);
  for (int i = 0; i < NI; ++i) n[i] = t + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < NDP; ++i) x[i] = x[i] * 1.0000001;
#pragma unroll
      for (int i = 0; i < NI; ++i) n[i] = n[i] + (n[i] >> 3);
#pragma unroll
      for (int i = 0; i < NL; ++i) { ld[i] = shv[64 * WAVES * (i & 3) + 8 * (i >> 2)]; }
#pragma unroll
      for (int i = 0; i < NL; ++i) asm volatile("" :: "v"(ld[i]));
#pragma unroll
      for (int i = 0; i < NA; ++i) atomicAdd(&sh[((t + 5 * i) & (64 * WAVES - 1)) + 64 * WAVES * (i & 3)], a);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[m % 3], acc[m % 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0.0;
  for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NDP; ++i) s += x[i];
  for (int i = 0; i < NI; ++i) s += n[i];
  for (int i = 0; i < NL; ++i) s += ld[i];
  s += sh[t];
  out[blockIdx.x * 64 * WAVES + t] = s;
}

template <int NDP, int NI, int WAVES, int NM, int NL = 0, int NA = 0>
static void run(double* out) {
  const int iters = 1000, cus = 256;
  probe<NDP, NI, WAVES, NM, NL, NA><<<cus, 64 * WAVES>>>(out, 10); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); probe<NDP, NI, WAVES, NM, NL, NA><<<cus, 64 * WAVES>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)iters * 4 * (WAVES / 4.0);           // groups per SIMD
  printf("matrix %d  fp64 vector %2d  integer %2d  LDS reads %2d  LDS adds %2d  waves/SIMD %d: %7.1f ns per group per SIMD\n", NM, NDP, NI, NL, NA, WAVES / 4, ms * 1e6 / groups);
}

int main() {
  double* out; hipMalloc(&out, 256 * 512 * sizeof(double) + 4096);
  run<0, 0, 8, 3, 0, 3>(out); run<0, 0, 8, 3, 0, 6>(out); run<0, 0, 8, 0, 0, 6>(out);
  run<0, 0, 8, 3>(out); run<3, 0, 8, 3>(out); run<6, 0, 8, 3>(out); run<12, 0, 8, 3>(out); run<0, 6, 8, 3>(out); run<0, 12, 8, 3>(out); run<0, 24, 8, 3>(out);
  run<12, 0, 8, 0>(out); run<0, 24, 8, 0>(out); run<6, 12, 8, 3>(out);
  run<0, 0, 4, 3>(out); run<3, 0, 4, 3>(out); run<12, 0, 4, 3>(out); run<0, 12, 4, 3>(out);
  hipFree(out);
  return 0;
}
