// rgp_chain.hip -- the M x M algebra of the SVI bound on the device, one 1024-thread workgroup per layer.
//
// Restates SVI_VarDTC.inference + comp_KL_qU (autoreg/inference/svi_vardtc.py:70-215) and the q(U) gradient routing
// of SparseGP_MPI._inference_vardtc (autoreg/layers.py:71-79,173-176) in terms of the explicit inverse factor
// Linv = chol(Kuu + jitter I)^-1 (cond(L) = sqrt(cond(K)), so the explicit inverse costs no accuracy that matters at
// the 1e-6 jitter the reference uses) -- every triangular solve of the reference becomes a small GEMM.
// The factorisations run in LDS (one padded M x M matrix, M <= 140); GEMMs run on L2-resident workspace matrices.
#include "rgp_chain.h"

#include <math.h>

#define CH_THREADS 1024
#define LOG_2PI 1.8378770664093453

// ---- LDS-resident factorisations (ld = M+1 keeps column accesses conflict free) ---------------------------------
// left-looking Cholesky, lower, in place; thread i owns row i.
static __device__ void lds_chol(double* A, int M, int ld, int* status) {
  const int i = threadIdx.x;
  for (int j = 0; j < M; ++j) {
    double s = 0.0;
    if (i >= j && i < M) {
      s = A[i * ld + j];
      for (int k = 0; k < j; ++k) s -= A[i * ld + k] * A[j * ld + k];
    }
    __syncthreads();
    if (i == j) {
      if (!(s > 0.0)) {
        if (atomicCAS(&status[0], 0, RGP_ERR_NOT_PD) == 0) status[1] = j;
      }
      A[j * ld + j] = sqrt(s);
    }
    __syncthreads();
    if (i > j && i < M) A[i * ld + j] = s / A[j * ld + j];
    else if (i < j) A[i * ld + j] = 0.0;          // zero the strict upper triangle
    __syncthreads();
  }
}

// in-place inverse of a lower-triangular matrix (LAPACK dtrti2 order, columns last to first); thread i owns row i.
static __device__ void lds_trtri(double* L, int M, int ld) {
  const int i = threadIdx.x;
  for (int j = M - 1; j >= 0; --j) {
    const double ajj = 1.0 / L[j * ld + j];
    double tsum = 0.0;
    if (i > j && i < M) {
      for (int k = j + 1; k <= i; ++k) tsum += L[i * ld + k] * L[k * ld + j];
    }
    __syncthreads();
    if (i > j && i < M) L[i * ld + j] = -ajj * tsum;
    if (i == j) L[j * ld + j] = ajj;
    __syncthreads();
  }
}

static __device__ void copy_in(double* lds, int ld, const double* g, int M) {
  for (int i = threadIdx.x; i < M * M; i += blockDim.x) lds[(i / M) * ld + (i % M)] = g[i];
  __syncthreads();
}
static __device__ void copy_out(double* g, const double* lds, int ld, int M) {
  for (int i = threadIdx.x; i < M * M; i += blockDim.x) g[i] = lds[(i / M) * ld + (i % M)];
  __syncthreads();
}

// ---- workgroup GEMM on global (L2-resident) matrices, 4x4 register tiles -----------------------------------------
// C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C      (row-major, C must not alias A or B)
template <bool TA, bool TB>
static __device__ void wg_gemm(int m, int n, int k, double alpha, const double* __restrict__ A, int lda,
                               const double* __restrict__ B, int ldb, double beta, double* __restrict__ C, int ldc) {
  const int tm = (m + 3) / 4, tn = (n + 3) / 4;
  for (int tile = threadIdx.x; tile < tm * tn; tile += blockDim.x) {
    const int i0 = (tile / tn) * 4, j0 = (tile % tn) * 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int kk = 0; kk < k; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int i = i0 + a < m ? i0 + a : m - 1;
        av[a] = TA ? A[(size_t)kk * lda + i] : A[(size_t)i * lda + kk];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = j0 + b < n ? j0 + b : n - 1;
        bv[b] = TB ? B[(size_t)j * ldb + kk] : B[(size_t)kk * ldb + j];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = i0 + a, j = j0 + b;
        if (i < m && j < n) {
          const double c = beta == 0.0 ? 0.0 : beta * C[(size_t)i * ldc + j];
          C[(size_t)i * ldc + j] = alpha * acc[a][b] + c;
        }
      }
  }
  __syncthreads();
}

// ---- stage 1: everything that does not depend on the data statistics ---------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) svi_chain_pre_kernel(SviChain c) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, ld = M + 1, t = threadIdx.x;
  const double beta = 1.0 / c.noise[0];                        // svi_vardtc.py:80
  // Lm = chol(K), Linv
  copy_in(lds, ld, c.Kj, M);
  lds_chol(lds, M, ld, c.status);
  double ldm = block_sum((t < M) ? log(lds[t * ld + t]) : 0.0, scratch);
  lds_trtri(lds, M, ld);
  copy_out(c.Linv, lds, ld, M);
  // S = W W^T + a I ; Ls ; LsInv                                  (layers.py:71, svi_vardtc.py:95)
  wg_gemm<false, true>(M, M, M, 1.0, c.qU_W, M, c.qU_W, M, 0.0, c.S, M);
  if (t < M) c.S[t * M + t] += c.qU_a[0];
  __syncthreads();
  copy_in(lds, ld, c.S, M);
  lds_chol(lds, M, ld, c.status);
  double lds_ = block_sum((t < M) ? log(lds[t * ld + t]) : 0.0, scratch);
  copy_out(c.Ls, lds, ld, M);
  lds_trtri(lds, M, ld);
  copy_out(c.LsInv, lds, ld, M);
  // E = Linv Ls ; gmu = Linv mu ; Kinv_mu = Linv^T gmu            (svi_vardtc.py:96-97,153)
  wg_gemm<false, false>(M, M, M, 1.0, c.Linv, M, c.Ls, M, 0.0, c.E, M);
  wg_gemm<false, false>(M, D, M, 1.0, c.Linv, M, c.qU_m, D, 0.0, c.gmu, D);
  wg_gemm<true, false>(M, D, M, 1.0, c.Linv, M, c.gmu, D, 0.0, c.Kinv_mu, D);
  // B = D E E^T + gmu gmu^T                                       (:111)
  wg_gemm<false, true>(M, M, M, 1.0, c.E, M, c.E, M, 0.0, c.EEt, M);
  for (int i = t; i < M * M; i += blockDim.x) c.B[i] = (double)D * c.EEt[i];
  __syncthreads();
  wg_gemm<false, true>(M, M, D, 1.0, c.gmu, D, c.gmu, D, 1.0, c.B, M);
  // dL_dpsi2 = beta/2 Linv^T (D I - B) Linv                       (:169), symmetric by construction
  for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = ((i / M == i % M) ? (double)D : 0.0) - c.B[i];
  __syncthreads();
  wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.Linv, M, 0.0, c.T2, M);
  wg_gemm<true, false>(M, M, M, 0.5 * beta, c.Linv, M, c.T2, M, 0.0, c.dLdpsi2, M);
  // dL_dpsi1 = beta Ymean Kinv_mu^T -> coefficient matrix          (:164-167)
  for (int i = t; i < M * D; i += blockDim.x) c.Cpsi1[i] = beta * c.Kinv_mu[i];
  // woodbury_inv = Linv^T (I - E E^T) Linv                        (:154)
  if (c.winv) {
    for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = ((i / M == i % M) ? 1.0 : 0.0) - c.EEt[i];
    __syncthreads();
    wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.Linv, M, 0.0, c.T2, M);
    wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.T2, M, 0.0, c.winv, M);
  }
  if (t == 0) {
    c.scal[0] = ldm;      // sum log diag Lm
    c.scal[1] = lds_;     // sum log diag Ls
  }
}

// ---- stage 2: after the statistics are known --------------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) svi_chain_post_kernel(SviChain c) {
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, t = threadIdx.x;
  const double Dd = (double)D, Nd = (double)c.N;
  const double beta = 1.0 / c.noise[0];
  const double ratio = c.qU_ratio;
  // A = beta Linv psi2 Linv^T                                     (:107)
  wg_gemm<false, false>(M, M, M, 1.0, c.Linv, M, c.psi2, M, 0.0, c.T1, M);
  wg_gemm<false, true>(M, M, M, beta, c.T1, M, c.Linv, M, 0.0, c.A, M);
  // Ybt (M x D) = beta Linv psi1Y^T                               (:98)
  wg_gemm<false, true>(M, D, M, beta, c.Linv, M, c.psi1Y, M, 0.0, c.Ybt, D);
  double sBA = 0.0, trA = 0.0, gY = 0.0, e2 = 0.0, g2 = 0.0;
  for (int i = t; i < M * M; i += blockDim.x) {
    sBA += c.B[i] * c.A[i];
    if (i / M == i % M) trA += c.A[i];
    e2 += c.E[i] * c.E[i];
  }
  for (int i = t; i < M * D; i += blockDim.x) {
    gY += c.gmu[i] * c.Ybt[i];
    g2 += c.gmu[i] * c.gmu[i];
  }
  sBA = block_sum(sBA, scratch); trA = block_sum(trA, scratch); gY = block_sum(gY, scratch);
  e2 = block_sum(e2, scratch); g2 = block_sum(g2, scratch);
  const double data_fit = -0.5 * Dd * beta * c.psi0sum[0] - 0.5 * beta * c.yry[0] - 0.5 * sBA + 0.5 * Dd * trA + gY;
  const double logL = -0.5 * Nd * Dd * LOG_2PI + 0.5 * Dd * Nd * log(beta) + data_fit;            // :121-123
  const double dthetaL = -0.5 * Dd * Nd * beta - data_fit * beta;                                   // :139
  const double KL = -0.5 * Dd * M - Dd * c.scal[1] + Dd * c.scal[0] + 0.5 * Dd * e2 + 0.5 * g2;      // :209
  // dL_dKmm = 1/2 (t1 + t1^T) + t3,  t1 = Linv^T B A Linv,  t3 = 1/2 Linv^T (-D A - t2 - t2^T) Linv   (:129-133)
  wg_gemm<false, false>(M, M, M, 1.0, c.B, M, c.A, M, 0.0, c.T1, M);          // B A
  wg_gemm<false, true>(M, M, D, 1.0, c.gmu, D, c.Ybt, D, 0.0, c.T2, M);       // t2 = gmu Ybar
  for (int i = t; i < M * M; i += blockDim.x) {
    const int r = i / M, cc = i % M;
    // U = 1/2 (BA + (BA)^T) + 1/2 (-D A - t2 - t2^T): everything that is sandwiched by Linv^T . Linv
    c.T3[i] = 0.5 * (c.T1[i] + c.T1[cc * M + r]) + 0.5 * (-Dd * c.A[i] - c.T2[i] - c.T2[cc * M + r]);
  }
  __syncthreads();
  wg_gemm<false, false>(M, M, M, 1.0, c.T3, M, c.Linv, M, 0.0, c.T1, M);
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.T1, M, 0.0, c.dKmm, M);
  // mA = -Linv^T A Linv ; dL_dqU_mean = mA mu + Linv^T Ybt ; dL_dqU_var = D/2 mA      (:145-147)
  wg_gemm<false, false>(M, M, M, 1.0, c.A, M, c.Linv, M, 0.0, c.T1, M);
  wg_gemm<true, false>(M, M, M, -1.0, c.Linv, M, c.T1, M, 0.0, c.mA, M);
  wg_gemm<false, false>(M, D, M, 1.0, c.mA, M, c.qU_m, D, 0.0, c.gqm, D);
  wg_gemm<true, false>(M, D, M, 1.0, c.Linv, M, c.Ybt, D, 1.0, c.gqm, D);
  // KL gradients (:211-213): Kinv = Linv^T Linv, Sinv = LsInv^T LsInv
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.Linv, M, 0.0, c.Kinv, M);
  wg_gemm<true, false>(M, M, M, 1.0, c.LsInv, M, c.LsInv, M, 0.0, c.T1, M);   // Sinv
  // Gv = dL_dqU_var - ratio dKL_dvar = D/2 mA - ratio D/2 (Kinv - Sinv)
  for (int i = t; i < M * M; i += blockDim.x) c.Gv[i] = 0.5 * Dd * c.mA[i] - ratio * 0.5 * Dd * (c.Kinv[i] - c.T1[i]);
  for (int i = t; i < M * D; i += blockDim.x) c.gqm[i] -= ratio * c.Kinv_mu[i];
  __syncthreads();
  // V = mu mu^T + D S ; dKL_dK = D/2 Kinv - 1/2 Kinv V Kinv
  for (int i = t; i < M * M; i += blockDim.x) c.T2[i] = Dd * c.S[i];
  __syncthreads();
  wg_gemm<false, true>(M, M, D, 1.0, c.qU_m, D, c.qU_m, D, 1.0, c.T2, M);
  wg_gemm<false, false>(M, M, M, 1.0, c.T2, M, c.Kinv, M, 0.0, c.T1, M);
  wg_gemm<false, false>(M, M, M, 1.0, c.Kinv, M, c.T1, M, 0.0, c.T3, M);
  for (int i = t; i < M * M; i += blockDim.x) c.dKmm[i] -= ratio * (0.5 * Dd * c.Kinv[i] - 0.5 * c.T3[i]);
  __syncthreads();
  // q(U) parameter gradients (layers.py:173-176)
  if (c.gqU_m) for (int i = t; i < M * D; i += blockDim.x) c.gqU_m[i] = c.gqm[i];
  if (c.gqU_W) {
    for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = c.Gv[i] + c.Gv[(i % M) * M + i / M];
    __syncthreads();
    wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.qU_W, M, 0.0, c.gqU_W, M);
  }
  double tr = block_sum((t < M) ? c.Gv[t * M + t] : 0.0, scratch);
  if (t == 0) {
    if (c.gqU_a) c.gqU_a[0] = tr;
    c.bound[0] = logL - ratio * KL;                              // layers.py:76
    c.dthetaL[0] = dthetaL;
  }
}

size_t svi_chain_ws_doubles(int M, int D) {
  return (size_t)16 * M * M + 6 * (size_t)M * D + 64;
}

void svi_chain_carve(SviChain& c, WsAlloc& ws) {
  const size_t mm = (size_t)c.M * c.M, md = (size_t)c.M * c.D;
  c.Linv = ws.take<double>(mm); c.S = ws.take<double>(mm); c.Ls = ws.take<double>(mm); c.LsInv = ws.take<double>(mm);
  c.E = ws.take<double>(mm); c.EEt = ws.take<double>(mm); c.B = ws.take<double>(mm); c.A = ws.take<double>(mm);
  c.T1 = ws.take<double>(mm); c.T2 = ws.take<double>(mm); c.T3 = ws.take<double>(mm); c.mA = ws.take<double>(mm);
  c.Kinv = ws.take<double>(mm); c.Gv = ws.take<double>(mm); c.dLdpsi2 = ws.take<double>(mm); c.dKmm = ws.take<double>(mm);
  c.gmu = ws.take<double>(md); c.Kinv_mu = ws.take<double>(md); c.Cpsi1 = ws.take<double>(md); c.Ybt = ws.take<double>(md);
  c.gqm = ws.take<double>(md);
  c.scal = ws.take<double>(8); c.bound = ws.take<double>(1); c.dthetaL = ws.take<double>(1);
}

rgp_status launch_svi_chain_pre(rgp_handle h, const SviChain& c) {
  if (c.M > 140) return rgp_fail(h, RGP_ERR_BAD_ARG, "svi chain: M > 140 needs the blocked factorisation (not in this build)");
  const size_t lds = (size_t)c.M * (c.M + 1) * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)svi_chain_pre_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(svi_chain_pre_kernel, dim3(1), dim3(CH_THREADS), lds, h->stream, c);
  return RGP_OK;
}

rgp_status launch_svi_chain_post(rgp_handle h, const SviChain& c) {
  hipLaunchKernelGGL(svi_chain_post_kernel, dim3(1), dim3(CH_THREADS), 0, h->stream, c);
  return RGP_OK;
}

// ==================================================================================================================
// collapsed bound (VarDTC.inference, autoreg/inference/vardtc.py:88-208), given the statistics.
// With Li = L^-1 = LL^-1 Lm^-1 (L = Lm LL, vardtc.py:126):
//   b^T = beta Li psi1Y^T, v^T = Li^T b^T;   C = b^T b (+ beta^2 Li PsiR Li^T for uncertain outputs, :136-138)
//   R = 1/2 Lm^-T [D I - LL^-T (C + D I) LL^-1] Lm^-1 (:140-141);  dL_dpsi2 = beta R;  dL_dKmm = R - D/2 Lm^-T A Lm^-1
// Outputs consumed by the row kernels: vT (M x D), G2 = Li^T Li (M x M), beta.
// ==================================================================================================================
__global__ void __launch_bounds__(CH_THREADS) vardtc_chain_kernel(VarDtcChain c) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, ld = M + 1, t = threadIdx.x;
  const double Dd = (double)D, Nd = (double)c.N;
  const double beta = 1.0 / fmax(c.noise[0], 1e-6);                 // vardtc.py:102
  copy_in(lds, ld, c.Kj, M);
  lds_chol(lds, M, ld, c.status);
  lds_trtri(lds, M, ld);
  copy_out(c.Linv, lds, ld, M);
  // A = beta Linv psi2 Linv^T ; Lambda = I + A ; LL = chol(Lambda)          (:120-125)
  wg_gemm<false, false>(M, M, M, 1.0, c.Linv, M, c.psi2, M, 0.0, c.T1, M);
  wg_gemm<false, true>(M, M, M, beta, c.T1, M, c.Linv, M, 0.0, c.A, M);
  for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = c.A[i] + ((i / M == i % M) ? 1.0 : 0.0);
  __syncthreads();
  copy_in(lds, ld, c.T1, M);
  lds_chol(lds, M, ld, c.status);
  const double ldLL = block_sum((t < M) ? log(lds[t * ld + t]) : 0.0, scratch);
  lds_trtri(lds, M, ld);
  copy_out(c.LLinv, lds, ld, M);
  wg_gemm<false, false>(M, M, M, 1.0, c.LLinv, M, c.Linv, M, 0.0, c.Li, M);          // Li = L^-1
  wg_gemm<true, false>(M, M, M, 1.0, c.Li, M, c.Li, M, 0.0, c.G2, M);                // G2 = Li^T Li
  wg_gemm<false, true>(M, D, M, beta, c.Li, M, c.psi1Y, M, 0.0, c.bT, D);            // b^T (M x D)   (:130)
  wg_gemm<true, false>(M, D, M, 1.0, c.Li, M, c.bT, D, 0.0, c.vT, D);                // v^T           (:132)
  wg_gemm<false, true>(M, M, D, 1.0, c.bT, D, c.bT, D, 0.0, c.C, M);                 // C = b^T b     (:133)
  if (c.y_uncertain) {
    wg_gemm<false, false>(M, M, M, 1.0, c.Li, M, c.PsiR, M, 0.0, c.T1, M);
    wg_gemm<false, true>(M, M, M, beta * beta, c.T1, M, c.Li, M, 1.0, c.C, M);       // + c^T c       (:137-138)
  }
  double trC = 0.0, trA = 0.0, sR = 0.0;
  for (int i = t; i < M; i += blockDim.x) { trC += c.C[i * M + i]; trA += c.A[i * M + i]; }
  trC = block_sum(trC, scratch); trA = block_sum(trA, scratch);
  // R = 1/2 Linv^T [D I - LLinv^T (C + D I) LLinv] Linv
  for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = c.C[i] + ((i / M == i % M) ? Dd : 0.0);
  __syncthreads();
  wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.LLinv, M, 0.0, c.T2, M);
  wg_gemm<true, false>(M, M, M, -1.0, c.LLinv, M, c.T2, M, 0.0, c.T3, M);
  for (int i = t; i < M * M; i += blockDim.x) c.T3[i] += ((i / M == i % M) ? Dd : 0.0);
  __syncthreads();
  wg_gemm<false, false>(M, M, M, 1.0, c.T3, M, c.Linv, M, 0.0, c.T2, M);
  wg_gemm<true, false>(M, M, M, 0.5, c.Linv, M, c.T2, M, 0.0, c.R, M);
  // dL_dKmm = R - D/2 Linv^T A Linv                                                    (:155)
  wg_gemm<false, false>(M, M, M, 1.0, c.A, M, c.Linv, M, 0.0, c.T2, M);
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.T2, M, 0.0, c.T3, M);
  for (int i = t; i < M * M; i += blockDim.x) {
    c.dKmm[i] = c.R[i] - 0.5 * Dd * c.T3[i];
    c.dLdpsi2[i] = beta * c.R[i];
    sR += c.R[i] * beta * c.psi2[i];
  }
  sR = block_sum(sR, scratch);
  if (c.winv) {                                                                         // (:161)
    wg_gemm<true, false>(M, M, M, -1.0, c.LLinv, M, c.LLinv, M, 0.0, c.T1, M);
    for (int i = t; i < M; i += blockDim.x) c.T1[i * M + i] += 1.0;
    __syncthreads();
    wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.Linv, M, 0.0, c.T2, M);
    wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.T2, M, 0.0, c.winv, M);
  }
  if (t == 0) {
    const double psi0b = beta * c.psi0sum[0], YRY = beta * c.yry[0];
    const double bbt = trC;                                       // ||b||^2 (+ ||c||^2) = tr C
    c.bound[0] = -0.5 * (Dd * (Nd * LOG_2PI - Nd * log(beta) + psi0b - trA) + YRY - bbt) - Dd * ldLL;     // (:148-149)
    c.dthetaL[0] = 0.5 * (YRY * beta + beta * Dd * psi0b - Nd * Dd * beta) - beta * sR - beta * trC;   // (:168)
    c.beta[0] = beta;
  }
}

size_t vardtc_chain_ws_doubles(int M, int D) { return (size_t)14 * M * M + 3 * (size_t)M * D + 64; }

void vardtc_chain_carve(VarDtcChain& c, WsAlloc& ws) {
  const size_t mm = (size_t)c.M * c.M, md = (size_t)c.M * c.D;
  c.Linv = ws.take<double>(mm); c.A = ws.take<double>(mm); c.LLinv = ws.take<double>(mm); c.Li = ws.take<double>(mm);
  c.G2 = ws.take<double>(mm); c.T1 = ws.take<double>(mm); c.T2 = ws.take<double>(mm); c.T3 = ws.take<double>(mm);
  c.C = ws.take<double>(mm); c.R = ws.take<double>(mm); c.dKmm = ws.take<double>(mm); c.dLdpsi2 = ws.take<double>(mm);
  c.bT = ws.take<double>(md); c.vT = ws.take<double>(md); c.Vb = ws.take<double>(md);
  c.scal = ws.take<double>(8); c.bound = ws.take<double>(1); c.dthetaL = ws.take<double>(1); c.beta = ws.take<double>(1);
}

rgp_status launch_vardtc_chain(rgp_handle h, const VarDtcChain& c) {
  if (c.M > 140) return rgp_fail(h, RGP_ERR_BAD_ARG, "vardtc chain: M > 140 needs the blocked factorisation (not in this build)");
  const size_t lds = (size_t)c.M * (c.M + 1) * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)vardtc_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("vardtc chain LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(vardtc_chain_kernel, dim3(1), dim3(CH_THREADS), lds, h->stream, c);
  return RGP_OK;
}
