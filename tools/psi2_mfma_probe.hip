// psi2_mfma_probe.hip -- throughput of the per-tile work of a pair-indexed fp64 MFMA form of the fused psi2 pass on gfx950
// (no result is checked: operands are synthetic; the instruction pattern per 16 x 16 tile of (data row, pair of inducing rows) is
// the one such a kernel would issue at Q = 40):
//   Gram   10 v_mfma_f64_16x16x4 on one accumulator, A operand read from an LDS row tile, B operand in registers
//   exp    4 exponentials per lane, three LDS reads (g0, h[n,a], h[n,b]) and two LDS fp64 atomic adds (v) per element
//   C      12 MFMAs (3 tiles of 16 window columns x 4), A = the weights as they are, B read from an LDS row tile
//   T      the weights transposed through LDS (4 writes + 4 reads), 12 MFMAs, 12 LDS atomic adds of the result
// STAGES selects how much of it runs: 1 = Gram only, 2 = + exp, 3 = + C, 4 = everything.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops tools/psi2_mfma_probe.hip -o tools/psi2_mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef double d4 __attribute__((ext_vector_type(4)));
#define R 32            // data rows of the LDS tiles
#define KS 10           // Gram K steps (Q = 40)
#define QT 3            // tiles of 16 window columns (Q padded to 48)
#define MH 101          // row stride of the h / v tiles (M = 100)

template <int STAGES>
__global__ void __launch_bounds__(512) probe(double* out, int iters) {
  __shared__ double dmL[R * 41], d2L[R * 49], g0L[R], HL[R * MH], VL[R * MH], TL[R * 49], TB[8 * 16 * 17];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, c = lane & 15;
  for (int i = t; i < R * 41; i += 512) dmL[i] = 1e-3 * (i % 7);
  for (int i = t; i < R * 49; i += 512) { d2L[i] = 1e-2 * (i % 5); TL[i] = 0.0; }
  for (int i = t; i < R * MH; i += 512) { HL[i] = -1e-2 * (i % 11); VL[i] = 0.0; }
  if (t < R) g0L[t] = -0.5;
  __syncthreads();
  double zb[KS], ztb[4][QT];
#pragma unroll
  for (int k = 0; k < KS; ++k) zb[k] = 1e-2 * (k + c);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < QT; ++q) ztb[r][q] = 1e-2 * (r + q + g);
  const int pa = (wave * 16 + c) % 100, pb = (wave * 16 + c + 37) % 100;
  const double dl = 1e-3 * (c + 1);
  d4 cacc[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) cacc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  double psum = 0.0;
  double* tb = TB + wave * (16 * 17);
  for (int it = 0; it < iters; ++it) {
    const int rs = it & 1;                                   // row sub-tile of the block
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < KS; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dmL[(rs * 16 + c) * 41 + 4 * k + g], zb[k], acc, 0, 0, 0);
    double wc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wc[r] = acc[r];
    if (STAGES >= 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = rs * 16 + g + 4 * r;
        const double e = exp(acc[r] * 1e-3 + g0L[n] + HL[n * MH + pa] + HL[n * MH + pb]);
        psum += e;
        const double w = e * dl;
        atomicAdd(&VL[n * MH + pa], w);
        atomicAdd(&VL[n * MH + pb], w);
        wc[r] = 2.0 * w;
      }
    }
    if (STAGES >= 3) {
#pragma unroll
      for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[r], d2L[(rs * 16 + g + 4 * r) * 49 + 16 * q + c], cacc[q], 0, 0, 0);
    }
    if (STAGES >= 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) tb[(g + 4 * r) * 17 + c] = wc[r];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double at[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) at[r] = tb[c * 17 + g + 4 * r];
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        d4 tacc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(at[r], ztb[r][q], tacc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&TL[(rs * 16 + g + 4 * r) * 49 + 16 * q + c], tacc[r]);
      }
    } else {
      psum += wc[0] + wc[1] + wc[2] + wc[3];
    }
  }
  __syncthreads();
  double s = psum + VL[t % (R * MH)] + TL[t % (R * 49)];
#pragma unroll
  for (int q = 0; q < QT; ++q) s += cacc[q][0] + cacc[q][1] + cacc[q][2] + cacc[q][3];
  out[blockIdx.x * 512 + t] = s;
}

template <int STAGES>
static void run(const char* name, double* out, int mfma_per_tile) {
  const int iters = 4000, cus = 256;
  probe<STAGES><<<cus, 512>>>(out, 10); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); probe<STAGES><<<cus, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)iters * 8 * cus;               // 16 x 16 tiles processed on the chip
  const double ns_tile_simd = ms * 1e6 / (iters * 2.0);       // two waves per SIMD: ns per tile per SIMD
  printf("%-34s %8.3f ms  %7.1f ns per tile per SIMD  %6.2f G element pairs/s  MFMA %5.1f TFLOP/s executed\n", name, ms, ns_tile_simd,
         tiles * 256 / (ms * 1e6), tiles * mfma_per_tile * 2048 / (ms * 1e9));
}

int main() {
  double* out; hipMalloc(&out, 256 * 512 * sizeof(double));
  run<1>("Gram (10 MFMA)", out, 10);
  run<2>("Gram + exp + v atomics", out, 10);
  run<3>("Gram + exp + C (22 MFMA)", out, 22);
  run<4>("all: + transpose + T (34 MFMA)", out, 34);
  // the fused VALU kernel of this library needs 5.66 ms for N x M x M = 130432 x 100 x 100 ordered pairs at Q = 40
  printf("for comparison: the fused VALU sweep processes %.2f G unordered pairs/s at Q = 40 (130432 x 5050 pairs in 5.66 ms)\n", 130432.0 * 5050 / 5.66e6);
  hipFree(out);
  return 0;
}
