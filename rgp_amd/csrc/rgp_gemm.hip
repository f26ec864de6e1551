// rgp_gemm.hip -- fp64 GEMM for gfx950 on v_mfma_f64_16x16x4_f64 (see rgp_gemm.h for the contract).
//
// Workgroup = 256 threads = 2 x 2 waves, block tile 64 x 64, K step 16; every wave owns a 32 x 32 sub-tile = 2 x 2 MFMA
// tiles of 16 x 16 (four independent accumulators per wave cover the instruction's latency).  Operand tiles are staged
// through LDS in K-major order (As[kk][i], Bs[kk][j]) so that each MFMA operand is one conflict-free ds_read_b64 per lane:
//   A operand: lane l holds op(A)(i = l & 15, k = l >> 4);  B operand: op(B)(k = l >> 4, j = l & 15);
//   result   : register r of lane l is C(row = (l >> 4) + 4 r, col = l & 15)         (the fp64 layout, not the f32 one).
// The next K step's global loads are issued before the current step's MFMAs and land in registers under them.
// Out-of-range rows / columns / K are zero-filled on load and masked on store, so every shape is legal.
#include "rgp_gemm.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define GB_M 64
#define GB_N 64
#define GB_K 16
#define GB_LD 68          // LDS row stride in doubles (64 + 4: rows of consecutive kk start 8 banks apart)

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_f64_kernel(const GemmArgs g) {
  __shared__ double As[GB_K * GB_LD];
  __shared__ double Bs[GB_K * GB_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = blockIdx.x * GB_M, j0 = blockIdx.y * GB_N;
  int kbeg = 0, kend = g.k;
  if (g.splits > 1) {
    const int kper = ((g.k + g.splits - 1) / g.splits + GB_K - 1) / GB_K * GB_K;
    kbeg = (int)blockIdx.z * kper;
    kend = (kbeg + kper < g.k) ? kbeg + kper : g.k;
  }
  const double* __restrict__ A = g.A;
  const double* __restrict__ B = g.B;
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  double va[4], vb[4];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      {
        int i, kk;
        if (TA) { kk = e >> 6; i = e & 63; } else { i = e >> 4; kk = e & 15; }
        const bool ok = (i0 + i < g.m) && (k0 + kk < kend);
        const size_t idx = TA ? (size_t)(k0 + kk) * g.lda + (i0 + i) : (size_t)(i0 + i) * g.lda + (k0 + kk);
        const double v = A[ok ? idx : 0];
        va[u] = ok ? v : 0.0;
      }
      {
        int jx, kk;
        if (TB) { jx = e >> 4; kk = e & 15; } else { kk = e >> 6; jx = e & 63; }
        const bool ok = (j0 + jx < g.n) && (k0 + kk < kend);
        const size_t idx = TB ? (size_t)(j0 + jx) * g.ldb + (k0 + kk) : (size_t)(k0 + kk) * g.ldb + (j0 + jx);
        const double v = B[ok ? idx : 0];
        vb[u] = ok ? v : 0.0;
      }
    }
  };
  if (kbeg < kend) load_tiles(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GB_K) {
    __syncthreads();                                  // the previous step's operand reads are done
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      if (TA) As[(e >> 6) * GB_LD + (e & 63)] = va[u]; else As[(e & 15) * GB_LD + (e >> 4)] = va[u];
      if (TB) Bs[(e & 15) * GB_LD + (e >> 4)] = vb[u]; else Bs[(e >> 6) * GB_LD + (e & 63)] = vb[u];
    }
    __syncthreads();
    if (k0 + GB_K < kend) load_tiles(k0 + GB_K);
#pragma unroll
    for (int k4 = 0; k4 < GB_K / 4; ++k4) {
      const int kk = 4 * k4 + (lane >> 4);
      const double a0 = As[kk * GB_LD + wr * 32 + (lane & 15)], a1 = As[kk * GB_LD + wr * 32 + 16 + (lane & 15)];
      const double b0 = Bs[kk * GB_LD + wc * 32 + (lane & 15)], b1 = Bs[kk * GB_LD + wc * 32 + 16 + (lane & 15)];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  const double alpha = g.alpha * (g.a_dev ? g.a_dev[0] : 1.0);
  const bool split = g.splits > 1;
  const double gamma = (!split && g.Cin) ? g.gamma * (g.g_dev ? g.g_dev[0] : 1.0) : 0.0;
  double* C = g.C + (split ? (size_t)blockIdx.z * (size_t)g.split_stride : 0);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r;
        const int col = j0 + wc * 32 + b * 16 + (lane & 15);
        if (row < g.m && col < g.n) {
          double v = alpha * acc[a][b][r];
          if (!split && g.E) v *= g.E[(size_t)row * g.lde + col];
          if (!split && g.Cin) v += gamma * g.Cin[(size_t)row * g.ldcin + col];
          C[(size_t)row * g.ldc + col] = v;
        }
      }
}

// ------------------------------------------------------------------------------------------------------------------
// Large products (both output dimensions in the hundreds or more): block tile 128 x 128 over eight waves, every wave a
// 64 x 32 sub-tile = 4 x 2 MFMA tiles (8 accumulators, 6 operand reads per 8 MFMAs, 120 VGPRs, so that two workgroups =
// four waves per SIMD fit); the larger block tile halves the operand bytes a workgroup pulls through L2 per flop against
// the 64 x 64 kernel above.  Tiles are handed out so that the workgroups resident on one XCD (linear workgroup ids that are equal
// modulo 8) work on an 8 x 8 patch of neighbouring tiles and share its 16 operand strips in that XCD's L2.
// ------------------------------------------------------------------------------------------------------------------
#define GL_M 128
#define GL_N 128
#define GL_LD 132
#define GL_GROUP 8
#ifndef GL_K
#define GL_K 16           // K step of the large kernel
#endif
#define GL_T 512          // threads of the large kernel: 2 x 4 waves
#ifndef GL_OCC
#define GL_OCC 2          // workgroups per CU the register budget is set for (2: four waves per SIMD, 128 VGPRs)
#endif
#define GL_PE (GL_M * GL_K / GL_T)     // operand elements per thread and K step

template <bool TA, bool TB>
__global__ void __launch_bounds__(GL_T, GL_OCC) gemm_f64_large_kernel(const GemmArgs g, const int nti, const int ntj) {
  __shared__ double As[GL_K * GL_LD];
  __shared__ double Bs[GL_K * GL_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 2, wc = wave & 3;             // 2 x 4 waves: 64 rows x 32 columns each
  // tile of this workgroup: XCD-contiguous renumbering, then 8-row panels walked column by column
  const int total = nti * ntj;
  const int lin = (int)blockIdx.x;
  const int xcd = lin & 7, slot = lin >> 3;
  const int qd = total >> 3, rm = total & 7;
  const int lp = xcd * qd + (xcd < rm ? xcd : rm) + slot;
  const int width = GL_GROUP * ntj;
  const int grp = lp / width, rem = lp - grp * width;
  const int first = grp * GL_GROUP;
  const int gsz = (nti - first < GL_GROUP) ? nti - first : GL_GROUP;
  const int ti = first + rem % gsz, tj = rem / gsz;
  const int i0 = ti * GL_M, j0 = tj * GL_N;
  int kbeg = 0, kend = g.k;
  if (g.splits > 1) {
    const int kper = ((g.k + g.splits - 1) / g.splits + GL_K - 1) / GL_K * GL_K;
    kbeg = (int)blockIdx.y * kper;
    kend = (kbeg + kper < g.k) ? kbeg + kper : g.k;
  }
  const double* __restrict__ A = g.A;
  const double* __restrict__ B = g.B;
  d4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  double va[GL_PE], vb[GL_PE];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int u = 0; u < GL_PE; ++u) {
      const int e = t + GL_T * u;
      {
        int i, kk;
        if (TA) { kk = e >> 7; i = e & 127; } else { i = e / GL_K; kk = e % GL_K; }
        const bool ok = (i0 + i < g.m) && (k0 + kk < kend);
        const size_t idx = TA ? (size_t)(k0 + kk) * g.lda + (i0 + i) : (size_t)(i0 + i) * g.lda + (k0 + kk);
        const double v = A[ok ? idx : 0];
        va[u] = ok ? v : 0.0;
      }
      {
        int jx, kk;
        if (TB) { jx = e / GL_K; kk = e % GL_K; } else { kk = e >> 7; jx = e & 127; }
        const bool ok = (j0 + jx < g.n) && (k0 + kk < kend);
        const size_t idx = TB ? (size_t)(j0 + jx) * g.ldb + (k0 + kk) : (size_t)(k0 + kk) * g.ldb + (j0 + jx);
        const double v = B[ok ? idx : 0];
        vb[u] = ok ? v : 0.0;
      }
    }
  };
  if (kbeg < kend) load_tiles(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GL_K) {
    __syncthreads();                                  // the previous step's operand reads are done
#pragma unroll
    for (int u = 0; u < GL_PE; ++u) {
      const int e = t + GL_T * u;
      if (TA) As[(e >> 7) * GL_LD + (e & 127)] = va[u]; else As[(e % GL_K) * GL_LD + (e / GL_K)] = va[u];
      if (TB) Bs[(e % GL_K) * GL_LD + (e / GL_K)] = vb[u]; else Bs[(e >> 7) * GL_LD + (e & 127)] = vb[u];
    }
    __syncthreads();
    if (k0 + GL_K < kend) load_tiles(k0 + GL_K);
#pragma unroll
    for (int k4 = 0; k4 < GL_K / 4; ++k4) {
      const int kk = 4 * k4 + (lane >> 4);
      double a[4], b[2];
#pragma unroll
      for (int x = 0; x < 4; ++x) a[x] = As[kk * GL_LD + wr * 64 + x * 16 + (lane & 15)];
#pragma unroll
      for (int y = 0; y < 2; ++y) b[y] = Bs[kk * GL_LD + wc * 32 + y * 16 + (lane & 15)];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
    }
  }
  const double alpha = g.alpha * (g.a_dev ? g.a_dev[0] : 1.0);
  const bool split = g.splits > 1;
  const double gamma = (!split && g.Cin) ? g.gamma * (g.g_dev ? g.g_dev[0] : 1.0) : 0.0;
  double* C = g.C + (split ? (size_t)blockIdx.y * (size_t)g.split_stride : 0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + wr * 64 + a * 16 + (lane >> 4) + 4 * r;
        const int col = j0 + wc * 32 + b * 16 + (lane & 15);
        if (row < g.m && col < g.n) {
          double v = alpha * acc[a][b][r];
          if (!split && g.E) v *= g.E[(size_t)row * g.lde + col];
          if (!split && g.Cin) v += gamma * g.Cin[(size_t)row * g.ldcin + col];
          C[(size_t)row * g.ldc + col] = v;
        }
      }
}

void launch_gemm(hipStream_t s, const GemmArgs& g) {
  if (g.m <= 0 || g.n <= 0) return;
  const int nti = (g.m + GL_M - 1) / GL_M, ntj = (g.n + GL_N - 1) / GL_N, ns = g.splits > 1 ? g.splits : 1;
  if (g.m >= 512 && g.n >= 256 && (int64_t)nti * ntj * ns >= 512) {       // enough 128 x 128 tiles to fill the chip twice
    const dim3 grid((unsigned)(nti * ntj), (unsigned)ns);
    if (g.ta && g.tb) hipLaunchKernelGGL((gemm_f64_large_kernel<true, true>), grid, dim3(GL_T), 0, s, g, nti, ntj);
    else if (g.ta) hipLaunchKernelGGL((gemm_f64_large_kernel<true, false>), grid, dim3(GL_T), 0, s, g, nti, ntj);
    else if (g.tb) hipLaunchKernelGGL((gemm_f64_large_kernel<false, true>), grid, dim3(GL_T), 0, s, g, nti, ntj);
    else hipLaunchKernelGGL((gemm_f64_large_kernel<false, false>), grid, dim3(GL_T), 0, s, g, nti, ntj);
    return;
  }
  const dim3 grid((unsigned)((g.m + GB_M - 1) / GB_M), (unsigned)((g.n + GB_N - 1) / GB_N), (unsigned)(g.splits > 1 ? g.splits : 1));
  if (g.ta && g.tb) hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, dim3(256), 0, s, g);
  else if (g.ta) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, dim3(256), 0, s, g);
  else if (g.tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, dim3(256), 0, s, g);
}
