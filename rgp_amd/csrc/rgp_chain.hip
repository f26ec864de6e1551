// rgp_chain.hip -- the M x M algebra of the SVI bound on the device, one 1024-thread workgroup per layer.
//
// Restates SVI_VarDTC.inference + comp_KL_qU (autoreg/inference/svi_vardtc.py:70-215) and the q(U) gradient routing
// of SparseGP_MPI._inference_vardtc (autoreg/layers.py:71-79,173-176) in terms of the explicit inverse factor
// Linv = chol(Kuu + jitter I)^-1 (cond(L) = sqrt(cond(K))) -- every triangular solve of the reference becomes a small GEMM.
// The gradient with respect to K_uu is kept in the reference's congruence form Linv^T (...) Linv: its entries are large and
// cancel when dL/dZ is summed from them, and only an error of that form cancels with them.  (Through the explicit
// K_uu^-1 = Linv^T Linv, as Kinv X Kinv, every entry is as accurate -- but dL/dZ loses a factor cond(K_uu): measured
// 1e-7 against 1e-10 at cond 8e5, 2e-4 against 6e-9 at cond 6e7.)
// The factorisations run in LDS (one padded M x M matrix, M <= 140); GEMMs run on L2-resident workspace matrices.
#include "rgp_chain.h"
#include "rgp_gemm.h"

#include <math.h>

#define CH_THREADS 1024
#define LOG_2PI 1.8378770664093453

// ---- LDS-resident factorisations (ld = M+1 keeps column accesses conflict free) ---------------------------------
// Both run on ONE wave (lanes own rows i = j+lane, j+lane+64): the column recurrences need an ordering point per
// column, and a 1024-thread s_barrier per column (300 of them per factorisation) cost more than the arithmetic.
// Inside one wave LDS operations execute in order; the wave-scope fences only stop the compiler from reordering.
static __device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// left-looking Cholesky, lower, in place (M <= 128).
static __device__ void lds_chol(double* A, int M, int ld, int* status) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    for (int j = 0; j < M; ++j) {
      double sv[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = j + lane + 64 * r;
        double s0 = 0.0, s1 = 0.0;
        if (i < M) {
          s0 = A[i * ld + j];
          double s2 = 0.0, s3 = 0.0;
          int k = 0;
          for (; k + 7 < j; k += 8) {                          // 16 LDS loads in flight per trip: latency-bound otherwise
            const double a0 = A[i * ld + k], a1 = A[i * ld + k + 1], a2 = A[i * ld + k + 2], a3 = A[i * ld + k + 3];
            const double a4 = A[i * ld + k + 4], a5 = A[i * ld + k + 5], a6 = A[i * ld + k + 6], a7 = A[i * ld + k + 7];
            const double b0 = A[j * ld + k], b1 = A[j * ld + k + 1], b2 = A[j * ld + k + 2], b3 = A[j * ld + k + 3];
            const double b4 = A[j * ld + k + 4], b5 = A[j * ld + k + 5], b6 = A[j * ld + k + 6], b7 = A[j * ld + k + 7];
            s0 -= a0 * b0; s1 -= a1 * b1; s2 -= a2 * b2; s3 -= a3 * b3;
            s0 -= a4 * b4; s1 -= a5 * b5; s2 -= a6 * b6; s3 -= a7 * b7;
          }
          for (; k < j; ++k) s0 -= A[i * ld + k] * A[j * ld + k];
          s0 += s2; s1 += s3;
        }
        sv[r] = s0 + s1;
      }
      const double piv = __shfl(sv[0], 0);                  // lane 0, r = 0 holds row i = j
      if (!(piv > 0.0) && lane == 0) {
        if (atomicCAS(&status[0], 0, RGP_ERR_NOT_PD) == 0) status[1] = j;
      }
      const double d = sqrt(piv);
      wave_lds_sync();
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = j + lane + 64 * r;
        if (i < M) A[i * ld + j] = (i == j) ? d : sv[r] / d;
      }
      for (int i = lane; i < j; i += 64) A[i * ld + j] = 0.0;        // zero the strict upper triangle
      wave_lds_sync();
    }
  }
  __syncthreads();
}

// in-place inverse of a lower-triangular matrix (LAPACK dtrti2 order, columns last to first).
static __device__ void lds_trtri(double* L, int M, int ld) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    for (int j = M - 1; j >= 0; --j) {
      const double ajj = 1.0 / L[j * ld + j];
      double tv[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = j + 1 + lane + 64 * r;
        double t0 = 0.0, t1 = 0.0;
        if (i < M) {
          double t2 = 0.0, t3 = 0.0;
          int k = j + 1;
          for (; k + 7 <= i; k += 8) {
            const double a0 = L[i * ld + k], a1 = L[i * ld + k + 1], a2 = L[i * ld + k + 2], a3 = L[i * ld + k + 3];
            const double a4 = L[i * ld + k + 4], a5 = L[i * ld + k + 5], a6 = L[i * ld + k + 6], a7 = L[i * ld + k + 7];
            const double b0 = L[k * ld + j], b1 = L[(k + 1) * ld + j], b2 = L[(k + 2) * ld + j], b3 = L[(k + 3) * ld + j];
            const double b4 = L[(k + 4) * ld + j], b5 = L[(k + 5) * ld + j], b6 = L[(k + 6) * ld + j], b7 = L[(k + 7) * ld + j];
            t0 += a0 * b0; t1 += a1 * b1; t2 += a2 * b2; t3 += a3 * b3;
            t0 += a4 * b4; t1 += a5 * b5; t2 += a6 * b6; t3 += a7 * b7;
          }
          for (; k <= i; ++k) t0 += L[i * ld + k] * L[k * ld + j];
          t0 += t2; t1 += t3;
        }
        tv[r] = t0 + t1;
      }
      wave_lds_sync();                                       // every lane has read column j before it is overwritten
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = j + 1 + lane + 64 * r;
        if (i < M) L[i * ld + j] = -ajj * tv[r];
      }
      if (lane == 0) L[j * ld + j] = ajj;
      wave_lds_sync();
    }
  }
  __syncthreads();
}

// ---- blocked factorisations: 16 x 16 diagonal blocks in registers of one wave, panels and trailing updates on all waves ----
// The column-by-column versions above spend ~1 us per column (dependent LDS round trips); with the diagonal block held one row
// per lane and exchanged with v_readlane a block costs ~3 us, and the O(M^3) part runs on 1024 threads.
#define FB 16
#define CH_MAX_M 128
#define LDS_FACT_EXTRA (CH_MAX_M * FB + 2 * FB)        // doubles of LDS behind the matrix: panel buffer + 1/diag

// Cholesky of the nb x nb block at A (lower, in place), wave 0 only; writes 1/diag to rdiag[0..nb).
static __device__ void chol16_wave(double* A, int ld, int nb, int* status, int piv_off, double* rdiag) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  double a[FB];
#pragma unroll
  for (int c = 0; c < FB; ++c) a[c] = (i < nb && c <= i) ? A[i * ld + c] : ((i == c) ? 1.0 : 0.0);     // identity padding
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    const double piv = readlane_f64(a[j], j);
    if (j < nb && !(piv > 0.0) && lane == 0) {
      if (atomicCAS(&status[0], 0, RGP_ERR_NOT_PD) == 0) status[1] = piv_off + j;
    }
    const double d = sqrt(piv), rd = 1.0 / d;
    if (lane == 0 && j < nb) rdiag[j] = rd;
    if (i == j) a[j] = d;
    else if (i > j) a[j] *= rd;
#pragma unroll
    for (int c = j + 1; c < FB; ++c) {
      const double lcj = readlane_f64(a[j], c);          // L[c][j]
      if (i >= c) a[c] -= a[j] * lcj;
    }
  }
  if (lane < FB && i < nb) {
#pragma unroll
    for (int c = 0; c < FB; ++c)
      if (c < nb) A[i * ld + c] = (c <= i) ? a[c] : 0.0;
  }
}

// Inverse of the nb x nb lower-triangular block at L (in place), wave 0 only: lane c solves L x = e_c.
static __device__ void trtri16_wave(double* L, int ld, int nb) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  double a[FB], x[FB];
#pragma unroll
  for (int c = 0; c < FB; ++c) a[c] = (i < nb && c <= i && c < nb) ? L[i * ld + c] : ((i == c) ? 1.0 : 0.0);
  double dsel = 1.0;                                        // this lane's diagonal element
#pragma unroll
  for (int c = 0; c < FB; ++c) dsel = (i == c) ? a[c] : dsel;
  const double rdv = 1.0 / dsel;
#pragma unroll
  for (int r = 0; r < FB; ++r) {
    double sacc = (r == i) ? 1.0 : 0.0;                    // lane i holds column i of the inverse
#pragma unroll
    for (int k = 0; k < r; ++k) sacc -= readlane_f64(a[k], r) * x[k];         // L[r][k] x[k]
    x[r] = sacc * readlane_f64(rdv, r);
  }
  if (lane < FB && i < nb) {
#pragma unroll
    for (int r = 0; r < FB; ++r)
      if (r < nb && r >= i) L[r * ld + i] = x[r];
  }
}

// blocked right-looking Cholesky, lower, in place.  `ext` = LDS_FACT_EXTRA doubles of LDS behind the matrix.
static __device__ void lds_chol_blocked(double* A, int M, int ld, int* status, double* ext, int piv_base = 0) {
  double* rdiag = ext + (size_t)CH_MAX_M * FB;
  const int t = threadIdx.x, T = blockDim.x;
  for (int kb = 0; kb < M; kb += FB) {
    const int nb = (M - kb < FB) ? (M - kb) : FB;
    if (t < 64) chol16_wave(A + kb * ld + kb, ld, nb, status, piv_base + kb, rdiag);
    __syncthreads();
    const int r0 = kb + nb, nrem = M - r0;
    if (nrem > 0) {
      if (t < nrem) {                                        // panel: X L11^T = A21, one row per thread
        const int row = r0 + t;
        double x[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j) {
          if (j < nb) {
            double sacc = A[row * ld + kb + j];
#pragma unroll
            for (int c = 0; c < j; ++c) sacc -= x[c] * A[(kb + j) * ld + kb + c];
            x[j] = sacc * rdiag[j];
          } else {
            x[j] = 0.0;
          }
        }
#pragma unroll
        for (int j = 0; j < FB; ++j)
          if (j < nb) A[row * ld + kb + j] = x[j];
      }
      __syncthreads();
      for (int e = t; e < nrem * nrem; e += T) {            // trailing update of the lower triangle
        const int i = e / nrem, c = e - i * nrem;
        if (c <= i) {
          double sacc = 0.0;
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if (j < nb) sacc += A[(r0 + i) * ld + kb + j] * A[(r0 + c) * ld + kb + j];
          A[(r0 + i) * ld + r0 + c] -= sacc;
        }
      }
      __syncthreads();
    }
  }
  for (int e = t; e < M * M; e += T) {
    const int i = e / M, c = e - i * M;
    if (c > i) A[i * ld + c] = 0.0;
  }
  __syncthreads();
}

// blocked in-place inverse of a lower-triangular matrix (LAPACK dtrtri order: block columns last to first).
static __device__ void lds_trtri_blocked(double* L, int M, int ld, double* ext) {
  double* tmp = ext;                                        // [nrem x FB]
  double* rdiag = ext + (size_t)CH_MAX_M * FB;
  const int t = threadIdx.x, T = blockDim.x;
  const int nblk = (M + FB - 1) / FB;
  for (int jb = (nblk - 1) * FB; jb >= 0; jb -= FB) {
    const int nb = (M - jb < FB) ? (M - jb) : FB;
    const int r0 = jb + nb, nrem = M - r0;
    if (nrem > 0) {
      if (t < nb) rdiag[t] = 1.0 / L[(jb + t) * ld + jb + t];
      // tmp = Linv22 * L21   (Linv22: the already inverted trailing triangle)
      for (int e = t; e < nrem * FB; e += T) {
        const int i = e / FB, c = e - i * FB;
        double sacc = 0.0;
        if (c < nb)
          for (int k = 0; k <= i; ++k) sacc += L[(r0 + i) * ld + r0 + k] * L[(r0 + k) * ld + jb + c];
        tmp[e] = sacc;
      }
      __syncthreads();
      if (t < nrem) {                                        // X L11 = -tmp, one row per thread, columns last to first
        double x[FB];
#pragma unroll
        for (int cc = 0; cc < FB; ++cc) {
          const int c = FB - 1 - cc;
          if (c < nb) {
            double sacc = -tmp[t * FB + c];
#pragma unroll
            for (int k = c + 1; k < FB; ++k)
              if (k < nb) sacc -= x[k] * L[(jb + k) * ld + jb + c];
            x[c] = sacc * rdiag[c];
          } else {
            x[c] = 0.0;
          }
        }
#pragma unroll
        for (int c = 0; c < FB; ++c)
          if (c < nb) L[(r0 + t) * ld + jb + c] = x[c];
      }
      __syncthreads();
    }
    if (t < 64) trtri16_wave(L + jb * ld + jb, ld, nb);
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

// ---- workgroup GEMM, k-panels of 32 staged through LDS, 4x4 register tiles ----------------------------------------
// C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C      (row-major, C must not alias A or B; m, n <= 128)
// `pan` points at >= 2*GP_K*GP_LD doubles of LDS (the factorisation buffer is reused between factorisations).
#define GP_K 32
#define GP_LD 132
#define GP_PE (GP_K * 128 / CH_THREADS)          // panel elements per thread

// branch-free panel loads: all loads of a panel are issued before the first is consumed (a guarded load per element
// compiled to a load -> wait -> store chain, one global-memory latency per element)
template <bool TA>
static __device__ __forceinline__ void gp_load_a(double (&va)[GP_PE], const double* __restrict__ A, int lda, int m, int k0, int kb) {
#pragma unroll
  for (int u = 0; u < GP_PE; ++u) {
    const int e = threadIdx.x + u * CH_THREADS;
    int kk, i;
    if (TA) { kk = e / 128; i = e % 128; } else { i = e / GP_K; kk = e % GP_K; }      // coalesced along the stored rows
    const bool ok = kk < kb && i < m;
    const int kc = ok ? k0 + kk : 0, ic = ok ? i : 0;
    const double v = TA ? A[(size_t)kc * lda + ic] : A[(size_t)ic * lda + kc];
    va[u] = ok ? v : 0.0;
  }
}
template <bool TB>
static __device__ __forceinline__ void gp_load_b(double (&vb)[GP_PE], const double* __restrict__ B, int ldb, int n, int k0, int kb) {
#pragma unroll
  for (int u = 0; u < GP_PE; ++u) {
    const int e = threadIdx.x + u * CH_THREADS;
    int kk, jx;
    if (TB) { jx = e / GP_K; kk = e % GP_K; } else { kk = e / 128; jx = e % 128; }
    const bool ok = kk < kb && jx < n;
    const int kc = ok ? k0 + kk : 0, jc = ok ? jx : 0;
    const double v = TB ? B[(size_t)jc * ldb + kc] : B[(size_t)kc * ldb + jc];
    vb[u] = ok ? v : 0.0;
  }
}

template <bool TA, bool TB>
static __device__ void wg_gemm(int m, int n, int k, double alpha, const double* __restrict__ A, int lda,
                               const double* __restrict__ B, int ldb, double beta, double* __restrict__ C, int ldc,
                               double* pan) {
  double* Ap = pan;                    // [GP_K][GP_LD]  Ap[kk][i] = op(A)(i, k0+kk)
  double* Bp = pan + GP_K * GP_LD;     // [GP_K][GP_LD]  Bp[kk][j] = op(B)(k0+kk, j)
  // thread (ig, jl): rows 4 ig .. 4 ig + 3, columns jl, jl+32, jl+64, jl+96.  The column interleave makes the Bp reads of a
  // 32-lane group hit consecutive banks (a 4-contiguous-column tile gives a 4-way ds_read_b64 conflict), the Ap reads are
  // broadcasts.
  const int ig = threadIdx.x >> 5, jl = threadIdx.x & 31;
  const int i0 = 4 * ig;
  const bool has = i0 < m && jl < n;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  double va[GP_PE], vb[GP_PE];
  gp_load_a<TA>(va, A, lda, m, 0, k < GP_K ? k : GP_K);
  gp_load_b<TB>(vb, B, ldb, n, 0, k < GP_K ? k : GP_K);
  for (int k0 = 0; k0 < k; k0 += GP_K) {
    const int kb = (k - k0 < GP_K) ? k - k0 : GP_K;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GP_PE; ++u) {
      const int e = threadIdx.x + u * CH_THREADS;
      int kk, i;
      if (TA) { kk = e / 128; i = e % 128; } else { i = e / GP_K; kk = e % GP_K; }
      Ap[kk * GP_LD + i] = va[u];
    }
#pragma unroll
    for (int u = 0; u < GP_PE; ++u) {
      const int e = threadIdx.x + u * CH_THREADS;
      int kk, jx;
      if (TB) { jx = e / GP_K; kk = e % GP_K; } else { kk = e / 128; jx = e % 128; }
      Bp[kk * GP_LD + jx] = vb[u];
    }
    __syncthreads();
    if (k0 + GP_K < k) {               // the next panel's loads stay in flight under this panel's arithmetic
      const int kn = (k - k0 - GP_K < GP_K) ? k - k0 - GP_K : GP_K;
      gp_load_a<TA>(va, A, lda, m, k0 + GP_K, kn);
      gp_load_b<TB>(vb, B, ldb, n, k0 + GP_K, kn);
    }
    if (has) {
#pragma unroll 4
      for (int kk = 0; kk < kb; ++kk) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = Ap[kk * GP_LD + i0 + a];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = Bp[kk * GP_LD + jl + 32 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
      }
    }
  }
  if (has) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = i0 + a, jx = jl + 32 * b;
        if (i < m && jx < n) {
          const double c = beta == 0.0 ? 0.0 : beta * C[(size_t)i * ldc + jx];
          C[(size_t)i * ldc + jx] = alpha * acc[a][b] + c;
        }
      }
  }
  __syncthreads();
}

// ==================================================================================================================
// SVI chain.  The reference's expressions (svi_vardtc.py:95-169, 205-213) are evaluated in an algebraically equal form
// that needs 11 instead of 20 M x M x M products; with Kinv = Linv^T Linv, V = mu mu^T + D S, B = Linv V Linv^T:
//   dL_dpsi2 = beta/2 Linv^T (D I - B) Linv                         (:169, kept in the reference's sandwich form: the
//                                                                    difference D I - B is formed before the congruence)
//   -1/2 sum(B o A) + D/2 tr A  = sum(dL_dpsi2 o psi2)              (:113-123, A = beta Linv psi2 Linv^T never formed)
//   sum(gmu o Ybt)              = sum(Cpsi1 o psi1Y^T)
//   A = beta Linv psi2 Linv^T;  mA = -Linv^T A Linv                 (:107,145)
//   dL_dKmm - ratio dKL_dK = Linv^T [ 1/2 (B A + A B) - D/2 A - 1/2 (t2 + t2^T) - ratio (D/2 I - B/2) ] Linv,
//                            t2 = gmu Ybar, Ybar^T = beta Linv psi1Y^T          (:129-133, 213: ONE congruence)
//   dL_dqU_mean = mA mu + Linv^T Ybar^T                             (:146)
//   D/2 ||E||_F^2 + 1/2 ||gmu||_F^2 = 1/2 tr B                      (:209)
//   woodbury_inv = Linv^T (I - E E^T) Linv = (2 / (beta D)) dL_dpsi2 + Kinv_mu Kinv_mu^T / D     (:154)
// ==================================================================================================================

// ---- stage 1a: K_uu factor and dL/dpsi1 coefficients (what the psi1 passes wait for) -------------------------------
__global__ void __launch_bounds__(CH_THREADS) svi_chain_pre_a_kernel(SviChain c) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, ld = M + 1, t = threadIdx.x;
  const double beta = 1.0 / c.noise[0];                        // svi_vardtc.py:80
  // Lm = chol(K), Linv
  copy_in(lds, ld, c.Kj, M);
  lds_chol_blocked(lds, M, ld, c.status, lds + (size_t)M * ld);
  double ldm = block_sum((t < M) ? log(lds[t * ld + t]) : 0.0, scratch);
  lds_trtri_blocked(lds, M, ld, lds + (size_t)M * ld);
  copy_out(c.Linv, lds, ld, M);
  // gmu = Linv mu ; Kinv_mu = Linv^T gmu                          (svi_vardtc.py:97,153)
  wg_gemm<false, false>(M, D, M, 1.0, c.Linv, M, c.qU_m, D, 0.0, c.gmu, D, lds);
  wg_gemm<true, false>(M, D, M, 1.0, c.Linv, M, c.gmu, D, 0.0, c.Kinv_mu, D, lds);
  // dL_dpsi1 = beta Ymean Kinv_mu^T -> coefficient matrix          (:164-167)
  for (int i = t; i < M * D; i += blockDim.x) c.Cpsi1[i] = beta * c.Kinv_mu[i];
  if (t == 0) c.scal[0] = ldm;      // sum log diag Lm
}

// ---- stage 1b: dL/dpsi2 (what the psi2 pass waits for) -------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) svi_chain_pre_b_kernel(SviChain c) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, t = threadIdx.x;
  const double Dd = (double)D;
  const double beta = 1.0 / c.noise[0];
  // S = W W^T + a I                                               (layers.py:71)
  wg_gemm<false, true>(M, M, M, 1.0, c.qU_W, M, c.qU_W, M, 0.0, c.S, M, lds);
  const double qa = c.qU_a[0];
  // V = D S + mu mu^T
  for (int i = t; i < M * M; i += blockDim.x) {
    const double sv = c.S[i] + ((i / M == i % M) ? qa : 0.0);
    c.S[i] = sv;
    c.T1[i] = Dd * sv;
  }
  __syncthreads();
  wg_gemm<false, true>(M, M, D, 1.0, c.qU_m, D, c.qU_m, D, 1.0, c.T1, M, lds);
  // B = Linv V Linv^T  ( = D E E^T + gmu gmu^T, :111)
  wg_gemm<false, false>(M, M, M, 1.0, c.Linv, M, c.T1, M, 0.0, c.T2, M, lds);
  wg_gemm<false, true>(M, M, M, 1.0, c.T2, M, c.Linv, M, 0.0, c.B, M, lds);
  // dL_dpsi2 = beta/2 Linv^T (D I - B) Linv                       (:169), symmetric by construction
  for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = ((i / M == i % M) ? Dd : 0.0) - c.B[i];
  __syncthreads();
  wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.Linv, M, 0.0, c.T2, M, lds);
  wg_gemm<true, false>(M, M, M, 0.5 * beta, c.Linv, M, c.T2, M, 0.0, c.dLdpsi2, M, lds);
  if (c.dLs_sym)
    for (int i = t; i < M * M; i += blockDim.x) c.dLs_sym[i] = 0.5 * (c.dLdpsi2[i] + c.dLdpsi2[(i % M) * M + i / M]);
  const double trB = block_sum((t < M) ? c.B[t * M + t] : 0.0, scratch);
  if (t == 0) c.scal[2] = trB;      // D ||E||^2 + ||gmu||^2
}

// ---- stage 1c: what only stage 2 needs: log det S, Kinv, Sinv (:205-213) and the posterior's woodbury_inv (:154) ----
__global__ void __launch_bounds__(CH_THREADS) svi_chain_pre_c_kernel(SviChain c) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, ld = M + 1, t = threadIdx.x;
  const double Dd = (double)D;
  const double beta = 1.0 / c.noise[0];
  copy_in(lds, ld, c.S, M);
  lds_chol_blocked(lds, M, ld, c.status, lds + (size_t)M * ld);                    // Ls (svi_vardtc.py:95)
  double lds_ = block_sum((t < M) ? log(lds[t * ld + t]) : 0.0, scratch);
  lds_trtri_blocked(lds, M, ld, lds + (size_t)M * ld);
  copy_out(c.LsInv, lds, ld, M);
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.Linv, M, 0.0, c.Kinv, M, lds);
  wg_gemm<true, false>(M, M, M, 1.0, c.LsInv, M, c.LsInv, M, 0.0, c.Sinv, M, lds);
  // woodbury_inv (see the identities above)
  for (int i = t; i < M * M; i += blockDim.x) c.Wi[i] = 2.0 / (beta * Dd) * c.dLdpsi2[i];
  __syncthreads();
  wg_gemm<false, true>(M, M, D, 1.0 / Dd, c.Kinv_mu, D, c.Kinv_mu, D, 1.0, c.Wi, M, lds);
  if (t == 0) c.scal[1] = lds_;     // sum log diag Ls
}

// ---- stage 2: after the statistics are known --------------------------------------------------------------------
__global__ void __launch_bounds__(CH_THREADS) svi_chain_post_kernel(SviChain c) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, t = threadIdx.x;
  const double Dd = (double)D, Nd = (double)c.N;
  const double beta = 1.0 / c.noise[0];
  const double ratio = c.qU_ratio;
  // A = beta Linv psi2 Linv^T -> T2 ;  B A -> T3 ;  Ybar^T = beta Linv psi1Y^T -> Ybt (M x D) ;  t2 = gmu Ybar -> dKmm (scratch)
  wg_gemm<false, true>(M, M, M, 1.0, c.psi2, M, c.Linv, M, 0.0, c.T1, M, lds);
  wg_gemm<false, false>(M, M, M, beta, c.Linv, M, c.T1, M, 0.0, c.T2, M, lds);
  wg_gemm<false, false>(M, M, M, 1.0, c.B, M, c.T2, M, 0.0, c.T3, M, lds);
  wg_gemm<false, true>(M, D, M, beta, c.Linv, M, c.psi1Y, M, 0.0, c.Ybt, D, lds);
  wg_gemm<false, true>(M, M, D, 1.0, c.gmu, D, c.Ybt, D, 0.0, c.dKmm, M, lds);
  double sP = 0.0, gY = 0.0;
  for (int i = t; i < M * M; i += blockDim.x) sP += c.dLdpsi2[i] * c.psi2[i];
  for (int i = t; i < M * D; i += blockDim.x) gY += c.Cpsi1[i] * c.psi1Y[(size_t)(i % D) * M + i / D];
  sP = block_sum(sP, scratch); gY = block_sum(gY, scratch);
  const double data_fit = -0.5 * Dd * beta * c.psi0sum[0] - 0.5 * beta * c.yry[0] + sP + gY;
  const double logL = -0.5 * Nd * Dd * LOG_2PI + 0.5 * Dd * Nd * log(beta) + data_fit;            // :121-123
  const double dthetaL = -0.5 * Dd * Nd * beta - data_fit * beta;                                   // :139
  const double KL = -0.5 * Dd * M - Dd * c.scal[1] + Dd * c.scal[0] + 0.5 * c.scal[2];               // :209
  // dL_dKmm (:129-133) minus ratio dKL_dK (:213) as one congruence;  mA (:145);  dL_dqU_mean = mA mu + Linv^T Ybar^T (:146);
  // dL_dqU_var = D/2 mA (:147)
  for (int i = t; i < M * M; i += blockDim.x) {
    const int r = i / M, cc = i % M, it = cc * M + r;
    c.T1[i] = 0.5 * (c.T3[i] + c.T3[it]) - 0.5 * Dd * c.T2[i] - 0.5 * (c.dKmm[i] + c.dKmm[it])
              - ratio * ((r == cc ? 0.5 * Dd : 0.0) - 0.5 * c.B[i]);
  }
  __syncthreads();
  wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.Linv, M, 0.0, c.T3, M, lds);
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.T3, M, 0.0, c.dKmm, M, lds);
  wg_gemm<false, false>(M, M, M, 1.0, c.T2, M, c.Linv, M, 0.0, c.T3, M, lds);
  wg_gemm<true, false>(M, M, M, -1.0, c.Linv, M, c.T3, M, 0.0, c.mA, M, lds);
  wg_gemm<true, false>(M, D, M, 1.0, c.Linv, M, c.Ybt, D, 0.0, c.gqm, D, lds);
  for (int i = t; i < M * M; i += blockDim.x)
    c.Gv[i] = 0.5 * Dd * c.mA[i] - ratio * 0.5 * Dd * (c.Kinv[i] - c.Sinv[i]);      // dL_dqU_var - ratio dKL_dvar
  for (int i = t; i < M * D; i += blockDim.x) c.gqm[i] -= ratio * c.Kinv_mu[i];
  __syncthreads();
  wg_gemm<false, false>(M, D, M, 1.0, c.mA, M, c.qU_m, D, 1.0, c.gqm, D, lds);
  // q(U) parameter gradients (layers.py:173-176)
  if (c.gqU_m) for (int i = t; i < M * D; i += blockDim.x) c.gqU_m[i] = c.gqm[i];
  if (c.gqU_W) {
    for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = c.Gv[i] + c.Gv[(i % M) * M + i / M];
    __syncthreads();
    wg_gemm<false, false>(M, M, M, 1.0, c.T1, M, c.qU_W, M, 0.0, c.gqU_W, M, lds);
  }
  if (c.winv) for (int i = t; i < M * M; i += blockDim.x) c.winv[i] = c.Wi[i];
  double tr = block_sum((t < M) ? c.Gv[t * M + t] : 0.0, scratch);
  if (t == 0) {
    if (c.gqU_a) c.gqU_a[0] = tr;
    c.bound[0] = logL - ratio * KL;                              // layers.py:76
    c.dthetaL[0] = dthetaL;
  }
}


// ==================================================================================================================
// M > 128: the same algebra with the factorisations blocked over workgroups and every M x M x M product on the MFMA
// GEMM (rgp_gemm.hip).  One stream, ~90 launches per layer: at these sizes a layer step spends >100 ms in the psi2 passes,
// so the chain is not hidden behind side streams.
//   Cholesky (right-looking, block NB_BIG = 128): diagonal block factored AND inverted in LDS by one workgroup
//   (lds_chol_blocked / lds_trtri_blocked above); panel A21 <- A21 L11^-T and trailing update A22 -= P P^T as GEMMs.
//   Inverse factor: Linv[b, 0:b] = -L_bb^-1 (L[b, 0:b] Linv[0:b, 0:b]), block row after block row (two GEMMs each).
// ==================================================================================================================
#define NB_BIG 128

static size_t big_fact_doubles(int M) {
  const size_t nblk = (size_t)(M + NB_BIG - 1) / NB_BIG;
  return nblk * NB_BIG * NB_BIG + (size_t)M * NB_BIG;
}
static void big_fact_carve(int M, double*& bigD, double*& bigP, WsAlloc& ws) {
  const size_t nblk = (size_t)(M + NB_BIG - 1) / NB_BIG;
  bigD = ws.take<double>(nblk * NB_BIG * NB_BIG);
  bigP = ws.take<double>((size_t)M * NB_BIG);
}

// A11 = nb x nb block at (kb, kb) of the M x M matrix A: A11 <- chol(A11) (strict upper part of the block zeroed),
// Dinv <- A11^-1 (nb x nb, row stride NB_BIG); logdet[0] (+)= sum log diag(chol)   (first block assigns)
__global__ void __launch_bounds__(CH_THREADS) chol_diag_kernel(double* __restrict__ A, int ld, int kb, int nb,
                                                                double* __restrict__ Dinv, double* logdet, int* status) {
  extern __shared__ double lds[];
  __shared__ double scratch[16];
  const int t = threadIdx.x, ldl = nb + 1;
  for (int i = t; i < nb * nb; i += blockDim.x) lds[(i / nb) * ldl + (i % nb)] = A[(size_t)(kb + i / nb) * ld + kb + (i % nb)];
  __syncthreads();
  lds_chol_blocked(lds, nb, ldl, status, lds + (size_t)nb * ldl, kb);
  const double ldsum = block_sum((t < nb) ? log(lds[t * ldl + t]) : 0.0, scratch);
  for (int i = t; i < nb * nb; i += blockDim.x) A[(size_t)(kb + i / nb) * ld + kb + (i % nb)] = lds[(i / nb) * ldl + (i % nb)];
  __syncthreads();
  lds_trtri_blocked(lds, nb, ldl, lds + (size_t)nb * ldl);
  for (int i = t; i < nb * nb; i += blockDim.x) {
    const int r = i / nb, cc = i % nb;
    Dinv[(size_t)r * NB_BIG + cc] = (cc <= r) ? lds[r * ldl + cc] : 0.0;
  }
  if (t == 0 && logdet) logdet[0] = (kb == 0 ? 0.0 : logdet[0]) + ldsum;
}

// dst[(r0 + i) * ldd + c0 + j] = src[i * lds_ + j], i < rows, j < cols
__global__ void __launch_bounds__(256) copy_block_kernel(int rows, int cols, const double* __restrict__ src, int lds_,
                                                          double* __restrict__ dst, int ldd) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)rows * cols) dst[(size_t)(e / cols) * ldd + (e % cols)] = src[(size_t)(e / cols) * lds_ + (e % cols)];
}
// mode 0: zero the strict upper triangle of the M x M matrix; mode 1: zero everything
__global__ void __launch_bounds__(256) zero_kernel(int M, double* __restrict__ A, int mode) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)M * M && (mode == 1 || (e % M) > (e / M))) A[e] = 0.0;
}
static inline unsigned nblocks256(int64_t n) { return (unsigned)((n + 255) / 256); }

// A (M x M, symmetric positive definite, full storage) <- its lower Cholesky factor; Dinv <- inverses of the diagonal blocks
static rgp_status chol_big(rgp_handle h, hipStream_t s, double* A, int M, double* Dinv, double* Ptmp, double* logdet, int* status) {
  const size_t lds = ((size_t)NB_BIG * (NB_BIG + 1) + LDS_FACT_EXTRA) * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)chol_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("blocked Cholesky LDS attr: ") + hipGetErrorString(e));
  for (int kb = 0, b = 0; kb < M; kb += NB_BIG, ++b) {
    const int nb = (M - kb < NB_BIG) ? M - kb : NB_BIG;
    double* Db = Dinv + (size_t)b * NB_BIG * NB_BIG;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(CH_THREADS), lds, s, A, M, kb, nb, Db, logdet, status);
    const int r0 = kb + nb, rem = M - r0;
    if (rem > 0) {
      gemm(s, rem, nb, nb, 1.0, A + (size_t)r0 * M + kb, M, false, Db, NB_BIG, true, 0.0, Ptmp, NB_BIG);       // P = A21 L11^-T
      hipLaunchKernelGGL(copy_block_kernel, dim3(nblocks256((int64_t)rem * nb)), dim3(256), 0, s, rem, nb, Ptmp, NB_BIG,
                         A + (size_t)r0 * M + kb, M);
      gemm(s, rem, rem, nb, -1.0, Ptmp, NB_BIG, false, Ptmp, NB_BIG, true, 1.0, A + (size_t)r0 * M + r0, M);  // A22 -= P P^T
    }
  }
  hipLaunchKernelGGL(zero_kernel, dim3(nblocks256((int64_t)M * M)), dim3(256), 0, s, M, A, 0);
  return RGP_OK;
}

// Linv <- L^-1 for the lower-triangular L whose diagonal-block inverses are in Dinv; Tbuf holds NB_BIG x M doubles
static void trtri_big(hipStream_t s, const double* L, int M, const double* Dinv, double* Linv, double* Tbuf) {
  hipLaunchKernelGGL(zero_kernel, dim3(nblocks256((int64_t)M * M)), dim3(256), 0, s, M, Linv, 1);
  for (int kb = 0, b = 0; kb < M; kb += NB_BIG, ++b) {
    const int nb = (M - kb < NB_BIG) ? M - kb : NB_BIG;
    const double* Db = Dinv + (size_t)b * NB_BIG * NB_BIG;
    hipLaunchKernelGGL(copy_block_kernel, dim3(nblocks256((int64_t)nb * nb)), dim3(256), 0, s, nb, nb, Db, NB_BIG,
                       Linv + (size_t)kb * M + kb, M);
    if (kb > 0) {
      gemm(s, nb, kb, kb, 1.0, L + (size_t)kb * M, M, false, Linv, M, false, 0.0, Tbuf, M);          // T = L[b, 0:b] Linv[0:b, 0:b]
      gemm(s, nb, kb, nb, -1.0, Db, NB_BIG, false, Tbuf, M, false, 0.0, Linv + (size_t)kb * M, M);   // Linv[b, 0:b] = -L_bb^-1 T
    }
  }
}

// ---- element-wise glue of the blocked SVI chain (same identities as the single-workgroup kernels above) -----------------
// scal: [0] sum log diag Lm, [1] sum log diag Ls, [2] tr B, [3] beta, [4] beta/2, [5] -beta, [6] 2/(beta D)
__global__ void big_scal_kernel(SviChain c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double beta = 1.0 / c.noise[0];
    c.scal[3] = beta; c.scal[4] = 0.5 * beta; c.scal[5] = -beta; c.scal[6] = 2.0 / (beta * (double)c.D);
  }
}
// what: 0  Cpsi1 = beta Kinv_mu                      (M x D)
//       1  S += a I ; T1 = D S                       (M x M)
//       2  T1 = D I - B
//       3  dLs_sym = 1/2 (dLdpsi2 + dLdpsi2^T)
//       4  Wi = 2/(beta D) dLdpsi2
//       5  Gv                                        (svi_vardtc.py:147,213 as in svi_chain_post_kernel)
//       6  gqm -= ratio Kinv_mu                      (M x D)
//       7  T1 = Gv + Gv^T
//       8  copy S to LsInv's storage (factored in place there)
//       9  T1 = inner matrix of the congruence that yields dL_dKmm - ratio dKL_dK   (:129-133,213; t2 in dKmm)
__global__ void __launch_bounds__(256) big_elem_kernel(SviChain c, int what) {
  const int M = c.M, D = c.D;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t mm = (int64_t)M * M, md = (int64_t)M * D;
  const double Dd = (double)D;
  const int r = (int)(i / M), cc = (int)(i % M);
  const int64_t it = (int64_t)cc * M + r;
  switch (what) {
    case 0: if (i < md) c.Cpsi1[i] = c.scal[3] * c.Kinv_mu[i]; break;
    case 1: if (i < mm) { const double sv = c.S[i] + (r == cc ? c.qU_a[0] : 0.0); c.S[i] = sv; c.T1[i] = Dd * sv; } break;
    case 2: if (i < mm) c.T1[i] = (r == cc ? Dd : 0.0) - c.B[i]; break;
    case 3: if (i < mm && c.dLs_sym) c.dLs_sym[i] = 0.5 * (c.dLdpsi2[i] + c.dLdpsi2[it]); break;
    case 4: if (i < mm) c.Wi[i] = c.scal[6] * c.dLdpsi2[i]; break;
    case 5: if (i < mm) c.Gv[i] = 0.5 * Dd * c.mA[i] - c.qU_ratio * 0.5 * Dd * (c.Kinv[i] - c.Sinv[i]); break;
    case 6: if (i < md) c.gqm[i] -= c.qU_ratio * c.Kinv_mu[i]; break;
    case 7: if (i < mm) c.T1[i] = c.Gv[i] + c.Gv[it]; break;
    case 8: if (i < mm) c.LsInv[i] = c.S[i]; break;
    case 9: if (i < mm)
      c.T1[i] = 0.5 * (c.T3[i] + c.T3[it]) - 0.5 * Dd * c.T2[i] - 0.5 * (c.dKmm[i] + c.dKmm[it])
                - c.qU_ratio * ((r == cc ? 0.5 * Dd : 0.0) - 0.5 * c.B[i]);
      break;
    default: break;
  }
}
static void big_elem(hipStream_t s, const SviChain& c, int what, bool md) {
  const int64_t n = md ? (int64_t)c.M * c.D : (int64_t)c.M * c.M;
  hipLaunchKernelGGL(big_elem_kernel, dim3(nblocks256(n)), dim3(256), 0, s, c, what);
}
__global__ void __launch_bounds__(CH_THREADS) big_trace_kernel(int M, const double* __restrict__ A, double* out) {
  __shared__ double scratch[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < M; i += blockDim.x) s += A[(size_t)i * M + i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) out[0] = s;
}
// the scalars of stage 2 (svi_vardtc.py:121-123,139,209; layers.py:76,176) and the plain copies into the caller's outputs
__global__ void __launch_bounds__(CH_THREADS) big_final_kernel(SviChain c) {
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, t = threadIdx.x;
  const double Dd = (double)D, Nd = (double)c.N;
  const double beta = c.scal[3], ratio = c.qU_ratio;
  double sP = 0.0, gY = 0.0, tr = 0.0;
  for (int64_t i = t; i < (int64_t)M * M; i += blockDim.x) sP += c.dLdpsi2[i] * c.psi2[i];
  for (int64_t i = t; i < (int64_t)M * D; i += blockDim.x) gY += c.Cpsi1[i] * c.psi1Y[(size_t)(i % D) * M + i / D];
  for (int i = t; i < M; i += blockDim.x) tr += c.Gv[(size_t)i * M + i];
  sP = block_sum(sP, scratch); gY = block_sum(gY, scratch); tr = block_sum(tr, scratch);
  if (c.gqU_m) for (int64_t i = t; i < (int64_t)M * D; i += blockDim.x) c.gqU_m[i] = c.gqm[i];
  if (c.winv) for (int64_t i = t; i < (int64_t)M * M; i += blockDim.x) c.winv[i] = c.Wi[i];
  if (t == 0) {
    const double data_fit = -0.5 * Dd * beta * c.psi0sum[0] - 0.5 * beta * c.yry[0] + sP + gY;
    const double logL = -0.5 * Nd * Dd * LOG_2PI + 0.5 * Dd * Nd * log(beta) + data_fit;
    const double KL = -0.5 * Dd * M - Dd * c.scal[1] + Dd * c.scal[0] + 0.5 * c.scal[2];
    if (c.gqU_a) c.gqU_a[0] = tr;
    c.bound[0] = logL - ratio * KL;
    c.dthetaL[0] = -0.5 * Dd * Nd * beta - data_fit * beta;
  }
}

static rgp_status svi_chain_pre_big(rgp_handle h, const SviChain& c, hipStream_t s, hipEvent_t ev_after_a, hipEvent_t ev_after_b) {
  const int M = c.M, D = c.D;
  rgp_status st;
  hipError_t e;
  hipLaunchKernelGGL(big_scal_kernel, dim3(1), dim3(64), 0, s, c);
  // ---- (a) Lm = chol(K), Linv; gmu = Linv mu; Kinv_mu = Linv^T gmu; Cpsi1 = beta Kinv_mu          (svi_vardtc.py:93-97,153,164-167)
  RGP_HIP_CHECK(h, hipMemcpyAsync(c.T3, c.Kj, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, s));
  if ((st = chol_big(h, s, c.T3, M, c.bigD, c.bigP, c.scal + 0, c.status))) return st;
  trtri_big(s, c.T3, M, c.bigD, c.Linv, c.bigP);
  gemm(s, M, D, M, 1.0, c.Linv, M, false, c.qU_m, D, false, 0.0, c.gmu, D);
  gemm(s, M, D, M, 1.0, c.Linv, M, true, c.gmu, D, false, 0.0, c.Kinv_mu, D);
  big_elem(s, c, 0, true);
  if (ev_after_a && (e = hipEventRecord(ev_after_a, s)) != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain event: ") + hipGetErrorString(e));
  // ---- (b) S = W W^T + a I; V = D S + mu mu^T; B = Linv V Linv^T; dL_dpsi2 = beta/2 Linv^T (D I - B) Linv      (layers.py:71, :111,169)
  gemm(s, M, M, M, 1.0, c.qU_W, M, false, c.qU_W, M, true, 0.0, c.S, M);
  big_elem(s, c, 1, false);
  gemm(s, M, M, D, 1.0, c.qU_m, D, false, c.qU_m, D, true, 1.0, c.T1, M);
  gemm(s, M, M, M, 1.0, c.Linv, M, false, c.T1, M, false, 0.0, c.T2, M);
  gemm(s, M, M, M, 1.0, c.T2, M, false, c.Linv, M, true, 0.0, c.B, M);
  big_elem(s, c, 2, false);
  gemm(s, M, M, M, 1.0, c.T1, M, false, c.Linv, M, false, 0.0, c.T2, M);
  gemm(s, M, M, M, 1.0, c.Linv, M, true, c.T2, M, false, 0.0, c.dLdpsi2, M, c.scal + 4);
  big_elem(s, c, 3, false);
  hipLaunchKernelGGL(big_trace_kernel, dim3(1), dim3(CH_THREADS), 0, s, M, c.B, c.scal + 2);
  if (ev_after_b && (e = hipEventRecord(ev_after_b, s)) != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain event: ") + hipGetErrorString(e));
  // ---- (c) Ls = chol(S), LsInv; Kinv; Sinv; woodbury_inv                                            (:95,205-213,154)
  big_elem(s, c, 8, false);
  if ((st = chol_big(h, s, c.LsInv, M, c.bigD, c.bigP, c.scal + 1, c.status))) return st;
  trtri_big(s, c.LsInv, M, c.bigD, c.T3, c.bigP);
  gemm(s, M, M, M, 1.0, c.T3, M, true, c.T3, M, false, 0.0, c.Sinv, M);
  gemm(s, M, M, M, 1.0, c.Linv, M, true, c.Linv, M, false, 0.0, c.Kinv, M);
  big_elem(s, c, 4, false);
  gemm(s, M, M, D, 1.0 / (double)D, c.Kinv_mu, D, false, c.Kinv_mu, D, true, 1.0, c.Wi, M);
  e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain (blocked): ") + hipGetErrorString(e));
  return RGP_OK;
}

static rgp_status svi_chain_post_big(rgp_handle h, const SviChain& c, hipStream_t s) {
  const int M = c.M, D = c.D;
  // A = beta Linv psi2 Linv^T -> T2 ; B A -> T3 ; Ybar^T = beta Linv psi1Y^T -> Ybt ; t2 = gmu Ybar -> dKmm (scratch) ;
  // inner matrix -> T1 ; dKmm = Linv^T T1 Linv ; mA = -Linv^T A Linv ; gqm = Linv^T Ybar^T - ratio Kinv_mu (+ mA mu below)
  gemm(s, M, M, M, 1.0, c.psi2, M, false, c.Linv, M, true, 0.0, c.T1, M);
  gemm(s, M, M, M, 1.0, c.Linv, M, false, c.T1, M, false, 0.0, c.T2, M, c.scal + 3);
  gemm(s, M, M, M, 1.0, c.B, M, false, c.T2, M, false, 0.0, c.T3, M);
  gemm(s, M, D, M, 1.0, c.Linv, M, false, c.psi1Y, M, true, 0.0, c.Ybt, D, c.scal + 3);
  gemm(s, M, M, D, 1.0, c.gmu, D, false, c.Ybt, D, true, 0.0, c.dKmm, M);
  big_elem(s, c, 9, false);
  gemm(s, M, M, M, 1.0, c.T1, M, false, c.Linv, M, false, 0.0, c.T3, M);
  gemm(s, M, M, M, 1.0, c.Linv, M, true, c.T3, M, false, 0.0, c.dKmm, M);
  gemm(s, M, M, M, 1.0, c.T2, M, false, c.Linv, M, false, 0.0, c.T3, M);
  gemm(s, M, M, M, -1.0, c.Linv, M, true, c.T3, M, false, 0.0, c.mA, M);
  gemm(s, M, D, M, 1.0, c.Linv, M, true, c.Ybt, D, false, 0.0, c.gqm, D);
  big_elem(s, c, 5, false);
  big_elem(s, c, 6, true);
  gemm(s, M, D, M, 1.0, c.mA, M, false, c.qU_m, D, false, 1.0, c.gqm, D);
  if (c.gqU_W) {
    big_elem(s, c, 7, false);
    gemm(s, M, M, M, 1.0, c.T1, M, false, c.qU_W, M, false, 0.0, c.gqU_W, M);
  }
  hipLaunchKernelGGL(big_final_kernel, dim3(1), dim3(CH_THREADS), 0, s, c);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain post (blocked): ") + hipGetErrorString(e));
  return RGP_OK;
}

size_t svi_chain_ws_doubles(int M, int D) {
  return (size_t)18 * M * M + 6 * (size_t)M * D + 64 + (M > CH_MAX_M ? big_fact_doubles(M) + 64 : 0);
}

void svi_chain_carve(SviChain& c, WsAlloc& ws) {
  const size_t mm = (size_t)c.M * c.M, md = (size_t)c.M * c.D;
  c.Linv = ws.take<double>(mm); c.S = ws.take<double>(mm); c.LsInv = ws.take<double>(mm); c.B = ws.take<double>(mm);
  c.T1 = ws.take<double>(mm); c.T2 = ws.take<double>(mm); c.T3 = ws.take<double>(mm); c.mA = ws.take<double>(mm);
  c.Kinv = ws.take<double>(mm); c.Gv = ws.take<double>(mm); c.dLdpsi2 = ws.take<double>(mm); c.dKmm = ws.take<double>(mm);
  c.Sinv = ws.take<double>(mm); c.Wi = ws.take<double>(mm);
  c.gmu = ws.take<double>(md); c.Kinv_mu = ws.take<double>(md); c.Cpsi1 = ws.take<double>(md); c.Ybt = ws.take<double>(md);
  c.gqm = ws.take<double>(md);
  c.scal = ws.take<double>(8); c.bound = ws.take<double>(1); c.dthetaL = ws.take<double>(1);
  if (c.M > CH_MAX_M) big_fact_carve(c.M, c.bigD, c.bigP, ws);
}

rgp_status launch_svi_chain_pre(rgp_handle h, const SviChain& c, hipStream_t stream, hipEvent_t ev_after_a, hipEvent_t ev_after_b) {
  if (c.M > CH_MAX_M) return svi_chain_pre_big(h, c, stream, ev_after_a, ev_after_b);
  size_t lds = ((size_t)c.M * (c.M + 1) + LDS_FACT_EXTRA) * sizeof(double);
  if (lds < 2 * GP_K * GP_LD * sizeof(double)) lds = 2 * GP_K * GP_LD * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)svi_chain_pre_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)svi_chain_pre_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)svi_chain_pre_c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(svi_chain_pre_a_kernel, dim3(1), dim3(CH_THREADS), lds, stream, c);
  if (ev_after_a && (e = hipEventRecord(ev_after_a, stream)) != hipSuccess)
    return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain event: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(svi_chain_pre_b_kernel, dim3(1), dim3(CH_THREADS), lds, stream, c);
  if (ev_after_b && (e = hipEventRecord(ev_after_b, stream)) != hipSuccess)
    return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain event: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(svi_chain_pre_c_kernel, dim3(1), dim3(CH_THREADS), lds, stream, c);
  return RGP_OK;
}

rgp_status launch_svi_chain_post(rgp_handle h, const SviChain& c, hipStream_t stream) {
  if (c.M > CH_MAX_M) return svi_chain_post_big(h, c, stream);
  const size_t lds = 2 * GP_K * GP_LD * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)svi_chain_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("svi chain post LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(svi_chain_post_kernel, dim3(1), dim3(CH_THREADS), lds, stream, c);
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
  lds_chol_blocked(lds, M, ld, c.status, lds + (size_t)M * ld);
  lds_trtri_blocked(lds, M, ld, lds + (size_t)M * ld);
  copy_out(c.Linv, lds, ld, M);
  // A = beta Linv psi2 Linv^T ; Lambda = I + A ; LL = chol(Lambda)          (:120-125)
  wg_gemm<false, false>(M, M, M, 1.0, c.Linv, M, c.psi2, M, 0.0, c.T1, M, lds);
  wg_gemm<false, true>(M, M, M, beta, c.T1, M, c.Linv, M, 0.0, c.A, M, lds);
  for (int i = t; i < M * M; i += blockDim.x) c.T1[i] = c.A[i] + ((i / M == i % M) ? 1.0 : 0.0);
  __syncthreads();
  copy_in(lds, ld, c.T1, M);
  lds_chol_blocked(lds, M, ld, c.status, lds + (size_t)M * ld);
  const double ldLL = block_sum((t < M) ? log(lds[t * ld + t]) : 0.0, scratch);
  lds_trtri_blocked(lds, M, ld, lds + (size_t)M * ld);
  copy_out(c.LLinv, lds, ld, M);
  wg_gemm<false, false>(M, M, M, 1.0, c.LLinv, M, c.Linv, M, 0.0, c.Li, M, lds);          // Li = L^-1
  wg_gemm<true, false>(M, M, M, 1.0, c.Li, M, c.Li, M, 0.0, c.G2, M, lds);                // G2 = Li^T Li
  wg_gemm<false, true>(M, D, M, beta, c.Li, M, c.psi1Y, M, 0.0, c.bT, D, lds);            // b^T (M x D)   (:130)
  wg_gemm<true, false>(M, D, M, 1.0, c.Li, M, c.bT, D, 0.0, c.vT, D, lds);                // v^T           (:132)
  // From here the reference's expressions are evaluated through Kinv = Linv^T Linv and G2 = Li^T Li (7 instead of 13
  // M x M x M products; the differences below carry the same absolute error eps ||Kinv|| as the reference's sandwiches of
  // D I - LL^-T (C + D I) LL^-1 and I - LL^-T LL^-1, which are differences themselves):
  //   tr C = ||b||_F^2 (+ beta^2 sum(G2 o PsiR));   Li^T C Li = v^T v (+ beta^2 G2 PsiR G2)
  //   R = 1/2 [D Kinv - D G2 - Li^T C Li]                                                 (:140-141)
  //   Linv^T A Linv stays a congruence (two products with Linv)                           (:155)
  //   woodbury_inv = Kinv - G2                                                            (:161)
  double* Kinv = c.C;
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.Linv, M, 0.0, Kinv, M, lds);
  double trC = 0.0, trA = 0.0, sR = 0.0;
  for (int i = t; i < M * D; i += blockDim.x) trC += c.bT[i] * c.bT[i];
  for (int i = t; i < M; i += blockDim.x) trA += c.A[i * M + i];
  for (int i = t; i < M * M; i += blockDim.x) {
    c.T1[i] = Dd * (Kinv[i] - c.G2[i]);
    if (c.y_uncertain) trC += beta * beta * c.G2[i] * c.PsiR[i];
  }
  trC = block_sum(trC, scratch); trA = block_sum(trA, scratch);
  wg_gemm<false, true>(M, M, D, -1.0, c.vT, D, c.vT, D, 1.0, c.T1, M, lds);
  if (c.y_uncertain) {
    wg_gemm<false, false>(M, M, M, 1.0, c.G2, M, c.PsiR, M, 0.0, c.T2, M, lds);
    wg_gemm<false, false>(M, M, M, -beta * beta, c.T2, M, c.G2, M, 1.0, c.T1, M, lds);
  }
  wg_gemm<false, false>(M, M, M, 1.0, c.A, M, c.Linv, M, 0.0, c.T2, M, lds);             // Linv^T A Linv as the congruence it is:
  wg_gemm<true, false>(M, M, M, 1.0, c.Linv, M, c.T2, M, 0.0, c.T3, M, lds);              // see the header on dL/dZ
  for (int i = t; i < M * M; i += blockDim.x) {
    const double R = 0.5 * c.T1[i];
    c.R[i] = R;
    c.dKmm[i] = R - 0.5 * Dd * c.T3[i];
    c.dLdpsi2[i] = beta * R;
    sR += R * beta * c.psi2[i];
    if (c.winv) c.winv[i] = Kinv[i] - c.G2[i];
  }
  sR = block_sum(sR, scratch);
  if (t == 0) {
    const double psi0b = beta * c.psi0sum[0], YRY = beta * c.yry[0];
    const double bbt = trC;                                       // ||b||^2 (+ ||c||^2) = tr C
    c.bound[0] = -0.5 * (Dd * (Nd * LOG_2PI - Nd * log(beta) + psi0b - trA) + YRY - bbt) - Dd * ldLL;     // (:148-149)
    c.dthetaL[0] = 0.5 * (YRY * beta + beta * Dd * psi0b - Nd * Dd * beta) - beta * sR - beta * trC;   // (:168)
    c.beta[0] = beta;
  }
}

// ---- the same chain for M > 128: blocked factorisations, every product an MFMA GEMM launch, element-wise glue --------------
// scal: [0] beta, [1] beta^2, [4] sum log diag LL (chol_big)
__global__ void vd_scal_kernel(VarDtcChain c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double beta = 1.0 / fmax(c.noise[0], 1e-6);                 // vardtc.py:102
    c.beta[0] = beta; c.scal[0] = beta; c.scal[1] = beta * beta;
  }
}
// what 0: T1 = A + I;  1: T1 = D (Kinv - G2)  (Kinv in c.C);  2: R, dKmm, dL_dpsi2, woodbury_inv from T1, T3
__global__ void __launch_bounds__(256) vd_elem_kernel(VarDtcChain c, int what) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int M = c.M;
  if (i >= (int64_t)M * M) return;
  const double Dd = (double)c.D;
  if (what == 0) c.T1[i] = c.A[i] + ((i / M == i % M) ? 1.0 : 0.0);
  else if (what == 1) c.T1[i] = Dd * (c.C[i] - c.G2[i]);
  else {
    const double beta = c.beta[0];
    const double R = 0.5 * c.T1[i];
    c.R[i] = R;
    c.dKmm[i] = R - 0.5 * Dd * c.T3[i];
    c.dLdpsi2[i] = beta * R;
    if (c.winv) c.winv[i] = c.C[i] - c.G2[i];
  }
}
// bound and dL/dnoise from the traces (same expressions as the end of vardtc_chain_kernel)
__global__ void __launch_bounds__(CH_THREADS) vd_final_kernel(VarDtcChain c) {
  __shared__ double scratch[16];
  const int M = c.M, D = c.D, t = threadIdx.x;
  const double Dd = (double)D, Nd = (double)c.N, beta = c.beta[0];
  double trC = 0.0, trA = 0.0, sR = 0.0;
  for (int i = t; i < M * D; i += blockDim.x) trC += c.bT[i] * c.bT[i];
  for (int i = t; i < M; i += blockDim.x) trA += c.A[(size_t)i * M + i];
  for (int64_t i = t; i < (int64_t)M * M; i += blockDim.x) {
    if (c.y_uncertain) trC += beta * beta * c.G2[i] * c.PsiR[i];
    sR += c.R[i] * beta * c.psi2[i];
  }
  trC = block_sum(trC, scratch); trA = block_sum(trA, scratch); sR = block_sum(sR, scratch);
  if (t == 0) {
    const double psi0b = beta * c.psi0sum[0], YRY = beta * c.yry[0], ldLL = c.scal[4];
    c.bound[0] = -0.5 * (Dd * (Nd * LOG_2PI - Nd * log(beta) + psi0b - trA) + YRY - trC) - Dd * ldLL;     // (:148-149)
    c.dthetaL[0] = 0.5 * (YRY * beta + beta * Dd * psi0b - Nd * Dd * beta) - beta * sR - beta * trC;       // (:168)
  }
}

static rgp_status vardtc_chain_big(rgp_handle h, const VarDtcChain& c, hipStream_t s) {
  const int M = c.M, D = c.D;
  const unsigned eb = nblocks256((int64_t)M * M);
  rgp_status st;
  hipLaunchKernelGGL(vd_scal_kernel, dim3(1), dim3(64), 0, s, c);
  RGP_HIP_CHECK(h, hipMemcpyAsync(c.T3, c.Kj, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, s));
  if ((st = chol_big(h, s, c.T3, M, c.bigD, c.bigP, nullptr, c.status))) return st;
  trtri_big(s, c.T3, M, c.bigD, c.Linv, c.bigP);
  // A = beta Linv psi2 Linv^T ; LL = chol(I + A)                                             (:120-125)
  gemm(s, M, M, M, 1.0, c.Linv, M, false, c.psi2, M, false, 0.0, c.T1, M);
  gemm(s, M, M, M, 1.0, c.T1, M, false, c.Linv, M, true, 0.0, c.A, M, c.scal + 0);
  hipLaunchKernelGGL(vd_elem_kernel, dim3(eb), dim3(256), 0, s, c, 0);
  if ((st = chol_big(h, s, c.T1, M, c.bigD, c.bigP, c.scal + 4, c.status))) return st;
  trtri_big(s, c.T1, M, c.bigD, c.LLinv, c.bigP);
  gemm(s, M, M, M, 1.0, c.LLinv, M, false, c.Linv, M, false, 0.0, c.Li, M);                  // Li = L^-1
  gemm(s, M, M, M, 1.0, c.Li, M, true, c.Li, M, false, 0.0, c.G2, M);                        // G2 = Li^T Li
  gemm(s, M, D, M, 1.0, c.Li, M, false, c.psi1Y, M, true, 0.0, c.bT, D, c.scal + 0);         // b^T (M x D)   (:130)
  gemm(s, M, D, M, 1.0, c.Li, M, true, c.bT, D, false, 0.0, c.vT, D);                        // v^T           (:132)
  gemm(s, M, M, M, 1.0, c.Linv, M, true, c.Linv, M, false, 0.0, c.C, M);                     // Kinv
  hipLaunchKernelGGL(vd_elem_kernel, dim3(eb), dim3(256), 0, s, c, 1);
  gemm(s, M, M, D, -1.0, c.vT, D, false, c.vT, D, true, 1.0, c.T1, M);
  if (c.y_uncertain) {
    gemm(s, M, M, M, 1.0, c.G2, M, false, c.PsiR, M, false, 0.0, c.T2, M);
    gemm(s, M, M, M, -1.0, c.T2, M, false, c.G2, M, false, 1.0, c.T1, M, c.scal + 1);
  }
  gemm(s, M, M, M, 1.0, c.A, M, false, c.Linv, M, false, 0.0, c.T2, M);            // Linv^T A Linv as a congruence
  gemm(s, M, M, M, 1.0, c.Linv, M, true, c.T2, M, false, 0.0, c.T3, M);
  hipLaunchKernelGGL(vd_elem_kernel, dim3(eb), dim3(256), 0, s, c, 2);
  hipLaunchKernelGGL(vd_final_kernel, dim3(1), dim3(CH_THREADS), 0, s, c);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("vardtc chain (blocked): ") + hipGetErrorString(e));
  return RGP_OK;
}

size_t vardtc_chain_ws_doubles(int M, int D) {
  return (size_t)14 * M * M + 3 * (size_t)M * D + 64 + (M > CH_MAX_M ? big_fact_doubles(M) + 64 : 0);
}

void vardtc_chain_carve(VarDtcChain& c, WsAlloc& ws) {
  const size_t mm = (size_t)c.M * c.M, md = (size_t)c.M * c.D;
  c.Linv = ws.take<double>(mm); c.A = ws.take<double>(mm); c.LLinv = ws.take<double>(mm); c.Li = ws.take<double>(mm);
  c.G2 = ws.take<double>(mm); c.T1 = ws.take<double>(mm); c.T2 = ws.take<double>(mm); c.T3 = ws.take<double>(mm);
  c.C = ws.take<double>(mm); c.R = ws.take<double>(mm); c.dKmm = ws.take<double>(mm); c.dLdpsi2 = ws.take<double>(mm);
  c.bT = ws.take<double>(md); c.vT = ws.take<double>(md); c.Vb = ws.take<double>(md);
  c.scal = ws.take<double>(8); c.bound = ws.take<double>(1); c.dthetaL = ws.take<double>(1); c.beta = ws.take<double>(1);
  if (c.M > CH_MAX_M) big_fact_carve(c.M, c.bigD, c.bigP, ws);
}

rgp_status launch_vardtc_chain(rgp_handle h, const VarDtcChain& c) {
  if (c.M > CH_MAX_M) return vardtc_chain_big(h, c, h->stream);
  size_t lds = ((size_t)c.M * (c.M + 1) + LDS_FACT_EXTRA) * sizeof(double);
  if (lds < 2 * GP_K * GP_LD * sizeof(double)) lds = 2 * GP_K * GP_LD * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)vardtc_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("vardtc chain LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(vardtc_chain_kernel, dim3(1), dim3(CH_THREADS), lds, h->stream, c);
  return RGP_OK;
}
