// rgp_gemm.h -- workgroup-tiled fp64 GEMM on v_mfma_f64_16x16x4_f64 for the parts of the path that ARE plain matrix
// products once M exceeds what one LDS-resident workgroup can hold: the M x M x M products of the bound's algebra
// (autoreg/inference/svi_vardtc.py:88-169,197-215) and the tall-skinny products of the psi1 side (N x M x D, N x M x 2Q).
// Internal to librgp_hip.so.
#pragma once
#include "rgp_common.h"

// C[m x n] = (alpha * a_dev) * op(A)[m x k] op(B)[k x n] (o E) + (gamma * g_dev) * Cin        (row-major everywhere)
//   op(A)(i, kk) = ta ? A[kk * lda + i] : A[i * lda + kk];   op(B)(kk, j) = tb ? B[j * ldb + kk] : B[kk * ldb + j]
//   a_dev / g_dev: optional DEVICE scalars multiplied into alpha / gamma (1/noise and friends live on the device: no host sync)
//   E: optional m x n element-wise factor of the product term;  Cin: optional, may be C itself (each element is read and
//   written by the same thread) but C must not overlap A or B.
//   splits > 1: split z of the K range writes alpha * partial product to C + z * split_stride (no E, no Cin); the caller sums
//   the partials in fixed order (launch_reduce).
struct GemmArgs {
  int m = 0, n = 0, k = 0;
  const double* A = nullptr; int lda = 0; int ta = 0;
  const double* B = nullptr; int ldb = 0; int tb = 0;
  double* C = nullptr; int ldc = 0;
  double alpha = 1.0; const double* a_dev = nullptr;
  double gamma = 0.0; const double* g_dev = nullptr; const double* Cin = nullptr; int ldcin = 0;
  const double* E = nullptr; int lde = 0;
  int splits = 1; int64_t split_stride = 0;
};
void launch_gemm(hipStream_t s, const GemmArgs& g);

// plain call: C = alpha op(A) op(B) + gamma C
static inline void gemm(hipStream_t s, int m, int n, int k, double alpha, const double* A, int lda, bool ta, const double* B,
                        int ldb, bool tb, double gamma, double* C, int ldc, const double* a_dev = nullptr) {
  GemmArgs g;
  g.m = m; g.n = n; g.k = k; g.A = A; g.lda = lda; g.ta = ta; g.B = B; g.ldb = ldb; g.tb = tb; g.C = C; g.ldc = ldc;
  g.alpha = alpha; g.a_dev = a_dev; g.gamma = gamma;
  if (gamma != 0.0) { g.Cin = C; g.ldcin = ldc; }
  launch_gemm(s, g);
}
