// rgp_psi.hip -- psi-statistic kernels for gfx950 (fp64).
//
// What the reference computes through GPy (kern.K / psi1 / psi2 and their gradients; call sites
// autoreg/inference/vardtc.py:59-61,110, svi_vardtc.py:48-50,88, autoreg/layers.py:98-134,1452-1458) is done here
// in a factorised form:
//
//   psi2n[n,m,m'] = exp( zd[m,m'] + g0[n] + h[n,m] + h[n,m'] - 1/2 sum_q d2[n,q] z[m,q] z[m',q] )
//   h[n,m]        = sum_q d2[n,q] z[m,q] (mu[n,q] - z[m,q]/4),   z = Z - c,  mu = window mean - c
//
// so the only O(N M^2 Q) work is the per-row Gram  G_n = (z diag(d2_n)) z^T  and, for the backward pass, the product
// y_n = W_n z with W_n = dLs * psi2n.  One thread owns one inducing row m; z[m',:] is wave-uniform and travels
// through SGPRs (scalar loads), so the inner products issue as v_fma_f64 with a scalar operand and no LDS traffic.
#include "rgp_psi.h"

#include <math.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------------------
// Z-derived terms
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prep_z_kernel(int M, int Q, int QP, const double* __restrict__ Z,
                                                      const double* __restrict__ variance,
                                                      const double* __restrict__ inv_l, double jitter, double* c,
                                                      double* lam, double* Zc, double* ZcT, double* zd, double* Kuu) {
  const int t = threadIdx.x;
  for (int q = t; q < QP; q += blockDim.x) {
    double s = 0.0;
    if (q < Q) {
      for (int m = 0; m < M; ++m) s += Z[(size_t)m * Q + q];
      s /= (double)M;
    }
    c[q] = s;
    lam[q] = q < Q ? inv_l[q] : 0.0;
  }
  __syncthreads();
  for (int i = t; i < M * QP; i += blockDim.x) {
    const int m = i / QP, q = i % QP;
    Zc[i] = q < Q ? Z[(size_t)m * Q + q] - c[q] : 0.0;
  }
  for (int i = t; i < QP * RGP_PSI2_TEAM; i += blockDim.x) {     // transposed, row-padded copy: coalesced per-lane rows
    const int q = i / RGP_PSI2_TEAM, m = i % RGP_PSI2_TEAM;
    ZcT[i] = (q < Q && m < M) ? Z[(size_t)m * Q + q] - c[q] : 0.0;
  }
}

// K_uu + jitter I and the Z-distance table, one workgroup per inducing row.
__global__ void __launch_bounds__(128) kuu_rows_kernel(int M, int Q, int QP, const double* __restrict__ variance,
                                                        const double* __restrict__ Zc, const double* __restrict__ lam,
                                                        double jitter, double* __restrict__ zd, double* __restrict__ Kuu) {
  const int a = blockIdx.x, b = threadIdx.x;
  if (b >= M) return;
  double r = 0.0;
  for (int q = 0; q < Q; ++q) {
    const double d = Zc[a * QP + q] - Zc[b * QP + q];
    r += lam[q] * d * d;
  }
  zd[a * M + b] = -0.25 * r;
  Kuu[a * M + b] = variance[0] * exp(-0.5 * r) + (a == b ? jitter : 0.0);
}

void launch_prep_z(hipStream_t s, int M, int Q, int QP, const double* Z, const double* variance, const double* inv_l,
                   double jitter, const ZTerms& zt) {
  hipLaunchKernelGGL(prep_z_kernel, dim3(1), dim3(256), 0, s, M, Q, QP, Z, variance, inv_l, jitter, zt.c, zt.lam, zt.Zc,
                     zt.ZcT, zt.zd, zt.Kuu);
  hipLaunchKernelGGL(kuu_rows_kernel, dim3(M), dim3(128), 0, s, M, Q, QP, variance, zt.Zc, zt.lam, jitter, zt.zd, zt.Kuu);
}

__global__ void row_offsets_kernel(int n_seq, int X_win, const int64_t* __restrict__ own_len, int64_t* row_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t acc = 0;
    for (int s = 0; s < n_seq; ++s) {
      row_off[s] = acc;
      acc += own_len[s] - X_win;
    }
    row_off[n_seq] = acc;
  }
}

void launch_row_offsets(hipStream_t s, int n_seq, int X_win, const int64_t* own_len, int64_t* row_off) {
  hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(64), 0, s, n_seq, X_win, own_len, row_off);
}

// ------------------------------------------------------------------------------------------------------------------
// row preparation: one wave per row, lanes over the Q window columns
// ------------------------------------------------------------------------------------------------------------------
struct RowPrepArgs {
  int64_t N; int Q, QP;
  const double *mu, *S;            // dense mode (mu != nullptr)
  WindowSpec w;                    // window mode
  const double* variance;
  const double *c, *lam;
  double *Xmu, *d2, *e1, *em, *g0, *g1;
  double *Ym, *Yv, *yry_part;
};

__global__ void __launch_bounds__(256) rowprep_kernel(RowPrepArgs a) {
  __shared__ double scratch[16];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wpb = blockDim.x >> 6;
  const double logvar = log(a.variance[0]);
  const bool dense = a.mu != nullptr;
  double yry = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < a.N; r += (int64_t)gridDim.x * wpb) {
    int s = 0;
    int64_t nloc = r, own_base = 0, up_base = 0;
    if (!dense) {
      int lo = 0, hi = a.w.n_seq;                    // largest s with row_off[s] <= r
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.w.row_off[mid] <= r) lo = mid; else hi = mid;
      }
      s = lo;
      nloc = r - a.w.row_off[s];
      const int64_t Ns = a.w.own_len[s] - a.w.X_win;
      own_base = a.w.own_off[s] + nloc * a.w.X_dim;                           // element where the own window starts
      if (a.w.U_win > 0) up_base = a.w.up_off[s] + ((a.w.up_len[s] - Ns - a.w.U_win + 1) + nloc) * a.w.U_dim;
    }
    const int Qx = dense ? 0 : a.w.X_win * a.w.X_dim;
    double sl2 = 0.0, sl1 = 0.0, sdm = 0.0, sem = 0.0;
    for (int q = lane; q < a.QP; q += 64) {
      double mu = 0.0, d2 = 0.0, e1 = 0.0;
      if (q < a.Q) {
        double m, S;
        if (dense) {
          m = a.mu[r * a.Q + q];
          S = a.S[r * a.Q + q];
        } else if (q < Qx) {                           // own history: contiguous slice of the flattened own signal
          m = a.w.own_mean[own_base + q];
          S = a.w.own_var ? a.w.own_var[own_base + q] : 0.0;
        } else {
          const int64_t i = up_base + (q - Qx);
          m = a.w.up_mean[i];
          S = a.w.up_var ? a.w.up_var[i] : a.w.up_var_const;
        }
        const double lam = a.lam[q];
        mu = m - a.c[q];
        const double t2 = 1.0 + 2.0 * S * lam, t1 = 1.0 + S * lam;
        d2 = lam / t2;
        e1 = lam / t1;
        sl2 += log(t2);
        sl1 += log(t1);
        sdm += d2 * mu * mu;
        sem += e1 * mu * mu;
      }
      a.Xmu[r * a.QP + q] = mu;
      a.d2[r * a.QP + q] = d2;
      a.e1[r * a.QP + q] = e1;
      a.em[r * a.QP + q] = e1 * mu;
    }
    sl2 = wave_sum(sl2); sl1 = wave_sum(sl1); sdm = wave_sum(sdm); sem = wave_sum(sem);
    if (lane == 0) {
      a.g0[r] = 2.0 * logvar - 0.5 * sl2 - sdm;
      a.g1[r] = logvar - 0.5 * sl1 - 0.5 * sem;          // includes -1/2 sum e mu^2 (the psi1 exponent is expanded)
    }
    if (!dense && a.Ym) {                              // outputs of this row: own[X_win + n]
      const int64_t ybase = a.w.own_off[s] + (a.w.X_win + nloc) * a.w.D;
      for (int d = lane; d < a.w.D; d += 64) {
        const double m = a.w.own_mean[ybase + d];
        a.Ym[r * a.w.D + d] = m;
        yry += m * m;
        if (a.w.y_uncertain) {
          const double v = a.w.own_var[ybase + d];
          a.Yv[r * a.w.D + d] = v;
          yry += v;
        }
      }
    }
  }
  if (a.yry_part) {
    const double tot = block_sum(yry, scratch);
    if (threadIdx.x == 0) a.yry_part[blockIdx.x] = tot;
  }
}

static int rowprep_blocks(int64_t N) {
  int64_t b = (N + 3) / 4;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

void launch_rowprep_dense(hipStream_t s, int64_t N, int Q, const double* mu, const double* S, const double* variance,
                          const ZTerms& zt, const RowTerms& rt) {
  RowPrepArgs a{};
  a.N = N; a.Q = Q; a.QP = zt.QP; a.mu = mu; a.S = S; a.variance = variance; a.c = zt.c; a.lam = zt.lam;
  a.Xmu = rt.Xmu; a.d2 = rt.d2; a.e1 = rt.e1; a.em = rt.em; a.g0 = rt.g0; a.g1 = rt.g1;
  hipLaunchKernelGGL(rowprep_kernel, dim3(rowprep_blocks(N)), dim3(256), 0, s, a);
}

// yry_out must hold 2048 doubles of partials followed by nothing else; caller reduces with launch_reduce.
void launch_rowprep_window(hipStream_t s, int64_t N, int Q, const WindowSpec& w, const double* variance,
                           const ZTerms& zt, const RowTerms& rt, double* Ym, double* Yv, double* yry_part) {
  RowPrepArgs a{};
  a.N = N; a.Q = Q; a.QP = zt.QP; a.mu = nullptr; a.S = nullptr; a.w = w; a.variance = variance; a.c = zt.c;
  a.lam = zt.lam; a.Xmu = rt.Xmu; a.d2 = rt.d2; a.e1 = rt.e1; a.em = rt.em; a.g0 = rt.g0; a.g1 = rt.g1;
  a.Ym = Ym; a.Yv = Yv; a.yry_part = yry_part;
  hipLaunchKernelGGL(rowprep_kernel, dim3(2048), dim3(256), 0, s, a);   // fixed grid: yry_part has 2048 slots
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 forward.  psi1[n,m] = exp(g1'[n] - 1/2 sum_q z_mq (e_nq z_mq - 2 (e mu)_nq)),  g1' = g1 - 1/2 sum e mu^2.
// Thread (m, team): z[m,:] lives in registers; e[n,:] and (e mu)[n,:] are wave-uniform -> SGPR operands (as in psi2).
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ void __launch_bounds__(256) psi1_fwd_kernel(const int64_t N, const int M, const double* __restrict__ ZcT,
                                                        const double* __restrict__ e1, const double* __restrict__ em,
                                                        const double* __restrict__ g1, double* __restrict__ psi1) {
  const int t = threadIdx.x, tr = t & 127;
  const int team = __builtin_amdgcn_readfirstlane(t >> 7);
  const bool act = tr < M;
  double z[QP];
#pragma unroll
  for (int q = 0; q < QP; ++q) z[q] = ZcT[q * RGP_PSI2_TEAM + tr];
  for (int64_t r = (int64_t)blockIdx.x * 2 + team; r < N; r += (int64_t)gridDim.x * 2) {
    const double* __restrict__ e = e1 + r * QP;
    const double* __restrict__ m = em + r * QP;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < QP; q += 2) {
      a0 += z[q] * (e[q] * z[q] - 2.0 * m[q]);
      a1 += z[q + 1] * (e[q + 1] * z[q + 1] - 2.0 * m[q + 1]);
    }
    if (act) psi1[r * M + tr] = exp(g1[r] - 0.5 * (a0 + a1));
  }
}

void launch_psi1_fwd(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, double* psi1) {
  int64_t b = (N + 1) / 2;
  const int blocks = (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
  switch (zt.QP) {
#define P1F(qp) case qp: hipLaunchKernelGGL((psi1_fwd_kernel<qp>), dim3(blocks), dim3(256), 0, s, N, M, zt.ZcT, rt.e1, rt.em, rt.g1, psi1); break;
    P1F(8) P1F(16) P1F(24) P1F(32) P1F(40) P1F(48) P1F(56) P1F(64)
#undef P1F
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 backward (row part): one wave per row.
//   Lp[m] = dL_dpsi1[n,m] psi1[n,m];  s0 = sum Lp;  s1[q] = sum Lp z[m,q];  s2[q] = sum Lp z[m,q]^2
//   dmu = -e (mu s0 - s1);  dS = 1/2 e^2 S2 - 1/2 e s0 with S2 = mu^2 s0 - 2 mu s1 + s2
//   dinv_l[q] += -1/2 (e/lam)^2 S2 - 1/2 s0 (1 - e/lam)/lam
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) psi1_bwd_kernel(int64_t N, int M, int Q, int QP, const double* __restrict__ Zc,
                                                        const double* __restrict__ lam, const double* __restrict__ Xmu,
                                                        const double* __restrict__ e1, const double* dLdpsi1,
                                                        const double* __restrict__ psi1, double* Lp,
                                                        double* __restrict__ gmu, double* __restrict__ gS,
                                                        double* __restrict__ dlam_part, double* __restrict__ lsum_part) {
  extern __shared__ double lds[];
  double* zs = lds;                         // [M x QP]   (reads are lane-consecutive in q)
  double* lps = lds + (size_t)M * QP;       // [4 waves x M]
  double* scratch = lps + 4 * (size_t)M;    // [16]
  double* dl_s = scratch + 16;              // [4 x QP]
  for (int i = threadIdx.x; i < M * QP; i += blockDim.x) zs[i] = Zc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wpb = blockDim.x >> 6;
  double* lp = lps + (size_t)wave * M;
  double dl_acc = 0.0;                      // lane q (QP <= 64)
  double ls_acc = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < N; r += (int64_t)gridDim.x * wpb) {
    double s0 = 0.0;
    for (int m = lane; m < M; m += 64) {
      const double v = dLdpsi1[r * M + m] * psi1[r * M + m];
      lp[m] = v;
      Lp[r * M + m] = v;
      s0 += v;
    }
    s0 = wave_sum(s0);
    ls_acc += s0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (lane < QP) {
      const int q = lane;
      double s1 = 0.0, s2 = 0.0;
      for (int m = 0; m < M; ++m) {
        const double z = zs[m * QP + q], l = lp[m];
        s1 += l * z;
        s2 += l * z * z;
      }
      const double mu = Xmu[r * QP + q], e = e1[r * QP + q];
      const double S2 = mu * mu * s0 - 2.0 * mu * s1 + s2;
      gmu[r * QP + q] = -e * (mu * s0 - s1);
      gS[r * QP + q] = 0.5 * e * e * S2 - 0.5 * e * s0;
      if (q < Q) {
        const double il = 1.0 / lam[q];
        const double el = e * il;
        dl_acc += -0.5 * el * el * S2 - 0.5 * s0 * (1.0 - el) * il;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
  if (lane < QP) dl_s[wave * QP + lane] = dl_acc;
  const double ls_tot = block_sum(lane == 0 ? ls_acc : 0.0, scratch);
  __syncthreads();
  if (threadIdx.x < QP) {
    double s = 0.0;
    for (int w = 0; w < 4; ++w) s += dl_s[w * QP + threadIdx.x];
    dlam_part[(size_t)blockIdx.x * QP + threadIdx.x] = s;
  }
  if (threadIdx.x == 0) lsum_part[blockIdx.x] = ls_tot;
}

void launch_psi1_bwd(hipStream_t s, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt, const double* dLdpsi1,
                     const double* psi1, double* Lp, double* gmu, double* gS, double* dlam_part, double* lsum_part,
                     int nblk) {
  const size_t lds = ((size_t)M * zt.QP + 4 * (size_t)M + 16 + 4 * (size_t)zt.QP) * sizeof(double);
  hipLaunchKernelGGL(psi1_bwd_kernel, dim3(nblk), dim3(256), lds, s, N, M, Q, zt.QP, zt.Zc, zt.lam, rt.Xmu, rt.e1,
                     dLdpsi1, psi1, Lp, gmu, gS, dlam_part, lsum_part);
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 backward (Z part): dZ[m,q] = sum_n Lp[n,m] e[n,q] (mu[n,q] - z[m,q])  -> per-block partials.
// Thread (g, q): q = t % QP, g = t / QP owns rows m = g, g+G, ...; tiles of 32 rows are staged in LDS.
// ------------------------------------------------------------------------------------------------------------------
#define DZ_TILE 32
#define DZ_MAXK 32
__global__ void __launch_bounds__(256) psi1_dz_kernel(int64_t N, int M, int QP, const double* __restrict__ Zc,
                                                       const double* __restrict__ Xmu, const double* __restrict__ e1,
                                                       const double* __restrict__ Lp, double* __restrict__ dz_part) {
  extern __shared__ double lds[];
  double* lp_s = lds;                                 // [DZ_TILE x M]
  double* em_s = lp_s + (size_t)DZ_TILE * M;          // [DZ_TILE x QP]  e*mu
  double* e_s = em_s + (size_t)DZ_TILE * QP;          // [DZ_TILE x QP]
  const int G = blockDim.x / QP;
  const int q = threadIdx.x % QP, g = threadIdx.x / QP;
  const bool act = g < G;
  double a1[DZ_MAXK], a2[DZ_MAXK];
#pragma unroll
  for (int k = 0; k < DZ_MAXK; ++k) { a1[k] = 0.0; a2[k] = 0.0; }
  const int64_t per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = (r0 + per < N) ? r0 + per : N;
  for (int64_t rb = r0; rb < r1; rb += DZ_TILE) {
    const int nr = (int)((r1 - rb) < DZ_TILE ? (r1 - rb) : DZ_TILE);
    __syncthreads();
    for (int i = threadIdx.x; i < nr * M; i += blockDim.x) lp_s[i] = Lp[rb * M + i];
    for (int i = threadIdx.x; i < nr * QP; i += blockDim.x) {
      const double e = e1[rb * QP + i];
      e_s[i] = e;
      em_s[i] = e * Xmu[rb * QP + i];
    }
    __syncthreads();
    if (act) {
      for (int r = 0; r < nr; ++r) {
        const double b1 = em_s[r * QP + q], b2 = e_s[r * QP + q];
#pragma unroll
        for (int k = 0; k < DZ_MAXK; ++k) {
          const int m = g + k * G;
          if (m < M) {
            const double l = lp_s[r * M + m];
            a1[k] += l * b1;
            a2[k] += l * b2;
          }
        }
      }
    }
  }
  if (act) {
#pragma unroll
    for (int k = 0; k < DZ_MAXK; ++k) {
      const int m = g + k * G;
      if (m < M) dz_part[((size_t)blockIdx.x * M + m) * QP + q] = a1[k] - Zc[m * QP + q] * a2[k];
    }
  }
}

void launch_psi1_dz(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, const double* Lp,
                    double* dz_part, int nblk) {
  const size_t lds = ((size_t)DZ_TILE * M + 2 * (size_t)DZ_TILE * zt.QP) * sizeof(double);
  hipLaunchKernelGGL(psi1_dz_kernel, dim3(nblk), dim3(256), lds, s, N, M, zt.QP, zt.Zc, rt.Xmu, rt.e1, Lp, dz_part);
}

// ------------------------------------------------------------------------------------------------------------------
// psi2: fused forward (sum over rows) and backward (row gradients, dZ, dinv_l, dvar partials) in one pass.
// ------------------------------------------------------------------------------------------------------------------
struct Psi2Args {
  int64_t N; int M, Q;
  const double *Zc, *ZcT, *zd, *lam, *dLs;
  const double *Xmu, *d2, *g0;
  Psi2Out o;
};

#define RED_QC 4                 // q columns per reduction chunk: each WAVE transposes 2*RED_QC values x 64 lanes in its own
#define RED_RS 72                // LDS region (no workgroup barrier); row stride 72 doubles = 16 mod 64 dwords keeps the four
                                 // value rows a 32-lane group reads in disjoint banks
#define RED_WAVE (2 * RED_QC * RED_RS)
#define PSI2_THREADS 256         // 2 halves x 128 inducing rows: half h sweeps m' in [h*ceil(M/2), ...)

// One (row n, inducing pair m,m') step.  zs = z[m',:] is wave-uniform (SGPR operands).
//   E = base + hb[m'] - 1/2 sum_q dzl[q] zs[q],  dzl = z_m (d2_n - lam)  (the Zdist factor is folded into the Gram:
//   -1/4 sum lam (z_m - z_m')^2 = -1/4 l_m - 1/4 l_m' + 1/2 sum lam z_m z_m', with l_m carried inside base / hb)
template <int QP, bool FWD, bool BWD>
static __device__ __forceinline__ void psi2_pair(const double* __restrict__ zs, const double (&dzl)[QP], double (&y)[QP],
                                                 double& v, const double base, const double hbm, const double dlv,
                                                 double* __restrict__ p2slot) {
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
  for (int q = 0; q < QP; q += 4) {
    c0 += dzl[q] * zs[q];
    c1 += dzl[q + 1] * zs[q + 1];
    c2 += dzl[q + 2] * zs[q + 2];
    c3 += dzl[q + 3] * zs[q + 3];
  }
  const double E = (base + hbm) - 0.5 * ((c0 + c1) + (c2 + c3));
  const double p = exp(E);
  if (FWD) atomicAdd(p2slot, p);          // ds_add_f64, no return: the slot has exactly one owner thread -> deterministic
  if (BWD) {
    const double w = p * dlv;
    v += w;
#pragma unroll
    for (int q = 0; q < QP; ++q) y[q] += w * zs[q];
  }
}

// Sweep m' over [a, b) for the thread's row: Gram + exp (+ forward accumulation, + y update).
// dLs[m', m] stream: two loads in flight (prefetch distance 2), consumed in registers without rotation copies.
template <int QP, bool FWD, bool BWD>
static __device__ __forceinline__ void psi2_sweep(const int a, const int b, const double* __restrict__ Zc,
                                                  const double* __restrict__ dlrow, const int M, const double (&dzl)[QP],
                                                  double (&y)[QP], double& v, const double base,
                                                  const double* __restrict__ hb, double* __restrict__ p2acc, const int tri,
                                                  const int tr, const bool act) {
  if (a >= b) return;
  const int last = b - 1;
  double dlA = BWD ? dlrow[(size_t)a * M] : 0.0;
  double dlB = BWD ? dlrow[(size_t)((a + 1 < b) ? a + 1 : last) * M] : 0.0;
  for (int mp = a; mp < b; mp += 2) {
    const int mpa = mp, mpb = (mp + 1 < b) ? mp + 1 : last;
    const double dla = dlA, dlb = dlB;
    if (BWD) {
      dlA = dlrow[(size_t)((mp + 2 < b) ? mp + 2 : last) * M];
      dlB = dlrow[(size_t)((mp + 3 < b) ? mp + 3 : last) * M];
    }
    {
      const bool own = act && mpa >= tr;
      psi2_pair<QP, FWD, BWD>(Zc + (size_t)mpa * QP, dzl, y, v, base, hb[mpa], dla,
                              p2acc + (own ? mpa * (mpa + 1) / 2 + tr : tri + tr));
    }
    if (mp + 1 < b) {
      const bool own = act && mpb >= tr;
      psi2_pair<QP, FWD, BWD>(Zc + (size_t)mpb * QP, dzl, y, v, base, hb[mpb], dlb,
                              p2acc + (own ? mpb * (mpb + 1) / 2 + tr : tri + tr));
    }
  }
}

// Workgroup = two TEAMS of 128 threads; team k works on data row r0 + 2*it + k and sweeps ALL m' for its inducing row, so
// y_n[m,:] is complete in one thread and the dZ slab is read-modified-written once per (row, m) -- half the L2 traffic of
// splitting the m' sweep between the teams.  Both teams accumulate psi2n into the shared packed M x M LDS array; to keep
// that deterministic the sweep runs in two phases separated by a barrier: team 0 does m' in [0, M/2) while team 1 does
// [M/2, M), then they swap, so every LDS slot receives team 0's add (of its row) strictly before / after team 1's.
// Every pointer is a separate noalias kernel argument: that is what lets the compiler prove the wave-uniform loads of
// z[m',:], d2[n,:] and mu[n,:] are not clobbered by the kernel's own stores and turn them into scalar loads.
// Occupancy: 2 workgroups (8 waves) per CU = 2 waves per SIMD; measured v_fma_f64 throughput is 31 TF at 1 wave/SIMD
// and 63.5 TF at 2 (profiles/r1_f64_peak.txt), so the register budget is capped at 256 by __launch_bounds__.
template <int QP, bool FWD, bool BWD>
__global__ void __launch_bounds__(PSI2_THREADS, 2)
psi2_rows_kernel(const int64_t N, const int M, const int Q, const double* __restrict__ Zc, const double* __restrict__ ZcT,
                 const double* __restrict__ lam_g,
                 const double* __restrict__ dLs_g, const double* __restrict__ Xmu_g, const double* __restrict__ d2_g,
                 const double* __restrict__ g0_g, double* __restrict__ p2_part, double* __restrict__ dz_part,
                 double* __restrict__ dlam_part, double* __restrict__ wsum_part, double* __restrict__ gmu_g,
                 double* __restrict__ gS_g, const int accumulate) {
  extern __shared__ double lds[];
  const int tri = M * (M + 1) / 2;
  double* p2acc = lds;                                   // [tri + 128] (+128 dummy slots)   (FWD)
  double* hbuf = lds + (FWD ? tri + RGP_PSI2_TEAM : 0);  // [2 teams x 128]
  double* red = hbuf + 2 * RGP_PSI2_TEAM;                // [4 waves x RED_WAVE]          (BWD)
  double* res = red + (BWD ? 4 * RED_WAVE : 0);          // [4 waves x (2*QP + 8)]        (BWD)

  const int t = threadIdx.x;
  const int tr = t & (RGP_PSI2_TEAM - 1);                // inducing row owned by this thread
  const int team = __builtin_amdgcn_readfirstlane(t >> 7);   // wave-uniform: keeps z[m'] and the row operands in SGPRs
  const bool act = tr < M;
  const int tm = act ? tr : M - 1;
  const int Mh = (M + 1) >> 1;
  const double* __restrict__ zrowp = Zc + (size_t)tm * QP;
#define zrow(q) zrowp[q]
  double* hb = hbuf + team * RGP_PSI2_TEAM;

  if (FWD) {
    for (int i = t; i < tri + RGP_PSI2_TEAM; i += PSI2_THREADS) p2acc[i] = 0.0;
  }
  // dZ partial of this (block, team): layout [q][row]
  double* dzp = BWD ? dz_part + ((size_t)blockIdx.x * 2 + team) * QP * RGP_PSI2_TEAM : nullptr;
  double ellq = 0.0;                                     // 1/4 sum_q lam_q z_mq^2
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    if (BWD) dzp[q * RGP_PSI2_TEAM + tr] = 0.0;
    ellq += 0.25 * lam_g[q] * zrow(q) * zrow(q);
  }
  double dlam_acc = 0.0, wsum_acc = 0.0;
  __syncthreads();

  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + 1) / 2 * 2;       // even: the teams take alternate rows
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  const double* __restrict__ dlrow = dLs_g + tm;          // dLs[m'][m]: coalesced over the lanes' rows

  for (int64_t rr = r0; rr < r1; rr += 2) {
    asm volatile("" ::: "memory");                          // keep the per-row reloads of z[m,:] out of the registers
    const bool valid = rr + team < r1;
    const int64_t n = valid ? rr + team : r1 - 1;
    const double* __restrict__ dn = d2_g + n * QP;          // wave-uniform -> scalar loads
    const double* __restrict__ mun = Xmu_g + n * QP;
    double dzl[QP];
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int q = 0; q < QP; q += 2) {
      const double z0 = zrow(q), z1 = zrow(q + 1);
      dzl[q] = z0 * (dn[q] - lam_g[q]);
      dzl[q + 1] = z1 * (dn[q + 1] - lam_g[q + 1]);
      h0 += z0 * dn[q] * (mun[q] - 0.25 * z0);
      h1 += z1 * dn[q + 1] * (mun[q + 1] - 0.25 * z1);
    }
    const double h = (h0 + h1) - ellq;
    hb[tr] = h;
    const double base = (act && valid) ? (g0_g[n] + h) : -INFINITY;    // exp(-inf) = 0: padded lanes / rows add nothing
    __syncthreads();

    double y[QP];
    double v = 0.0;
    if (BWD) {
#pragma unroll
      for (int q = 0; q < QP; ++q) y[q] = 0.0;
    }
    // phase 1: team 0 sweeps [0, Mh), team 1 sweeps [Mh, M);  phase 2: swapped
    psi2_sweep<QP, FWD, BWD>(team ? Mh : 0, team ? M : Mh, Zc, dlrow, M, dzl, y, v, base, hb, p2acc, tri, tr, act);
    if (FWD) __syncthreads();
    psi2_sweep<QP, FWD, BWD>(team ? 0 : Mh, team ? Mh : M, Zc, dlrow, M, dzl, y, v, base, hb, p2acc, tri, tr, act);

    if (BWD) {
      asm volatile("" ::: "memory");
      // dZ partial:  d (v (2 mu - z) - y)
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const double z = zrow(q);
        dzp[q * RGP_PSI2_TEAM + tr] += dn[q] * (v * (2.0 * mun[q] - z) - y[q]);
      }
      // a1[q] = sum_m v z ; a23[q] = 1/2 sum_m z (v z + y) ; w = sum_m v   (sums over the team's 128 rows).
      // Each wave reduces its 64 lanes through its own LDS region (in-order LDS, no workgroup barrier), then one barrier
      // and a 2-term sum per value.
      const int wave = t >> 6, lane = t & 63;
      double* wres = res + wave * (2 * QP + 8);
      {
        double* wred = red + wave * RED_WAVE;
        const double ws = wave_sum(v);
        if (lane == 0) wres[2 * QP] = ws;
        for (int q0 = 0; q0 < QP; q0 += RED_QC) {
#pragma unroll
          for (int j = 0; j < RED_QC; ++j) {
            const double z = zrow(q0 + j);
            wred[j * RED_RS + lane] = v * z;
            wred[(RED_QC + j) * RED_RS + lane] = 0.5 * z * (v * z + y[q0 + j]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int jv = lane >> 3, part = lane & 7;       // 8 values x 8 partial sums of 8 entries
          double sacc = 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k) sacc += wred[jv * RED_RS + part + 8 * k];
          sacc += __shfl_xor(sacc, 1);
          sacc += __shfl_xor(sacc, 2);
          sacc += __shfl_xor(sacc, 4);
          if (part == 0) wres[(jv < RED_QC ? 0 : QP) + q0 + (jv % RED_QC)] = sacc;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
      __syncthreads();
      const int RW = 2 * QP + 8;
      const double* tres = res + (2 * team) * RW;            // the team's two waves
      const double w = tres[2 * QP] + tres[RW + 2 * QP];
      if (tr < Q && valid) {
        const int q = tr;
        const double mu = mun[q], dq = dn[q];
        const double a1 = tres[q] + tres[RW + q];
        const double a23 = tres[QP + q] + tres[RW + QP + q];
        const double dmu = 2.0 * dq * (a1 - mu * w);
        const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
        const double dS = -2.0 * dq * dq * dd;
        if (accumulate) {
          gmu_g[n * QP + q] += dmu;
          gS_g[n * QP + q] += dS;
        } else {
          gmu_g[n * QP + q] = dmu;
          gS_g[n * QP + q] = dS;
        }
        const double il = 1.0 / lam_g[q];
        dlam_acc += dd * (dq * il) * (dq * il) - 0.5 * w * il;
      } else if (tr >= Q && tr < QP && valid && !accumulate) {
        gmu_g[n * QP + tr] = 0.0;
        gS_g[n * QP + tr] = 0.0;
      }
      if (tr == 0 && valid) wsum_acc += w;
      __syncthreads();
    } else {
      __syncthreads();      // hbuf is rewritten at the top of the next iteration
    }
  }

  if (FWD) {
    __syncthreads();
    double* out = p2_part + (size_t)blockIdx.x * tri;
    for (int i = t; i < tri; i += PSI2_THREADS) out[i] = p2acc[i];
  }
  if (BWD) {
    if (tr < QP) dlam_part[((size_t)blockIdx.x * 2 + team) * QP + tr] = (tr < Q) ? dlam_acc : 0.0;
    if (tr == 0) wsum_part[(size_t)blockIdx.x * 2 + team] = wsum_acc;
  }
}
#undef zrow

// ------------------------------------------------------------------------------------------------------------------
// Fused forward + backward pass, FOUR teams (512 threads, one workgroup per CU), used when its LDS image fits 160 KB
// (M = 100, Q = 40: 129 KB).  Same arithmetic as psi2_rows_kernel<QP, true, true>, but the dZ accumulator ([QP][128])
// lives in LDS next to the packed M x M forward accumulator instead of an L2 slab: the slab version moved 8-11 GB of
// HBM/fabric traffic per launch (profiles/r1_psi2_traffic_raw.json) for 0.2 GB of algorithmic bytes.
// Team k works on data row rr + k.  Shared accumulators stay deterministic by rotation: in sweep phase ph team k handles
// the m' quarter (k + ph) % 4, and in dZ phase ph it adds the q quarter (k + ph) % 4 -- within a phase the teams touch
// disjoint LDS slots, and every slot is updated by teams in phase order.
// ------------------------------------------------------------------------------------------------------------------
#define PSI2_T4 512
template <int QP>
__global__ void __launch_bounds__(PSI2_T4, 2)
psi2_rows4_kernel(const int64_t N, const int M, const int Q, const double* __restrict__ Zc, const double* __restrict__ lam_g,
                  const double* __restrict__ dLs_g, const double* __restrict__ Xmu_g, const double* __restrict__ d2_g,
                  const double* __restrict__ g0_g, double* __restrict__ p2_part, double* __restrict__ dz_part,
                  double* __restrict__ dlam_part, double* __restrict__ wsum_part, double* __restrict__ gmu_g,
                  double* __restrict__ gS_g, const int accumulate) {
  extern __shared__ double lds[];
  const int tri = M * (M + 1) / 2;
  double* p2acc = lds;                                   // [tri + 128] (+128 dummy slots)
  double* dzacc = p2acc + tri + RGP_PSI2_TEAM;           // [QP x 128]
  double* hbuf = dzacc + QP * RGP_PSI2_TEAM;             // [4 teams x 128]
  double* red = hbuf + 4 * RGP_PSI2_TEAM;                // [8 waves x RED_WAVE]
  double* res = red + 8 * RED_WAVE;                      // [8 waves x (2*QP + 8)]

  const int t = threadIdx.x;
  const int tr = t & (RGP_PSI2_TEAM - 1);
  const int team = __builtin_amdgcn_readfirstlane(t >> 7);     // 0..3, wave-uniform
  const bool act = tr < M;
  const int tm = act ? tr : M - 1;
  const int Mq = (M + 3) >> 2;
  const double* __restrict__ zrowp = Zc + (size_t)tm * QP;
#define zrow(q) zrowp[q]
  double* hb = hbuf + team * RGP_PSI2_TEAM;

  for (int i = t; i < tri + RGP_PSI2_TEAM + QP * RGP_PSI2_TEAM; i += PSI2_T4) p2acc[i] = 0.0;     // p2acc and dzacc are adjacent
  double ellq = 0.0;
#pragma unroll
  for (int q = 0; q < QP; ++q) ellq += 0.25 * lam_g[q] * zrow(q) * zrow(q);
  double dlam_acc = 0.0, wsum_acc = 0.0;
  __syncthreads();

  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + 3) / 4 * 4;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  const double* __restrict__ dlrow = dLs_g + tm;

  for (int64_t rr = r0; rr < r1; rr += 4) {
    asm volatile("" ::: "memory");
    const bool valid = rr + team < r1;
    const int64_t n = valid ? rr + team : r1 - 1;
    const double* __restrict__ dn = d2_g + n * QP;
    const double* __restrict__ mun = Xmu_g + n * QP;
    double dzl[QP];
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int q = 0; q < QP; q += 2) {
      const double z0 = zrow(q), z1 = zrow(q + 1);
      dzl[q] = z0 * (dn[q] - lam_g[q]);
      dzl[q + 1] = z1 * (dn[q + 1] - lam_g[q + 1]);
      h0 += z0 * dn[q] * (mun[q] - 0.25 * z0);
      h1 += z1 * dn[q + 1] * (mun[q + 1] - 0.25 * z1);
    }
    const double h = (h0 + h1) - ellq;
    hb[tr] = h;
    const double base = (act && valid) ? (g0_g[n] + h) : -INFINITY;
    __syncthreads();

    double y[QP];
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < QP; ++q) y[q] = 0.0;
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
      const int rq = (team + ph) & 3;
      const int a = rq * Mq, b = (a + Mq < M) ? a + Mq : M;
      psi2_sweep<QP, true, true>(a, b, Zc, dlrow, M, dzl, y, v, base, hb, p2acc, tri, tr, act);
      if (ph < 3) __syncthreads();
    }

    asm volatile("" ::: "memory");
    // per-team reductions (two waves each) through per-wave LDS regions
    const int wave = t >> 6, lane = t & 63;
    double* wres = res + wave * (2 * QP + 8);
    {
      double* wred = red + wave * RED_WAVE;
      const double ws = wave_sum(v);
      if (lane == 0) wres[2 * QP] = ws;
      for (int q0 = 0; q0 < QP; q0 += RED_QC) {
#pragma unroll
        for (int j = 0; j < RED_QC; ++j) {
          const double z = zrow(q0 + j);
          wred[j * RED_RS + lane] = v * z;
          wred[(RED_QC + j) * RED_RS + lane] = 0.5 * z * (v * z + y[q0 + j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int jv = lane >> 3, part = lane & 7;
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += wred[jv * RED_RS + part + 8 * k];
        sacc += __shfl_xor(sacc, 1);
        sacc += __shfl_xor(sacc, 2);
        sacc += __shfl_xor(sacc, 4);
        if (part == 0) wres[(jv < RED_QC ? 0 : QP) + q0 + (jv % RED_QC)] = sacc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    // dZ accumulation  d (v (2 mu - z) - y)  into the LDS slab, q quarters rotated over the teams
    constexpr int QQ = QP / 4;                           // QP is a multiple of 8
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
      const int rq = (team + ph) & 3;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (rq == k) {
#pragma unroll
          for (int j = 0; j < QQ; ++j) {
            const int q = k * QQ + j;
            dzacc[q * RGP_PSI2_TEAM + tr] += dn[q] * (v * (2.0 * mun[q] - zrow(q)) - y[q]);
          }
        }
      }
      __syncthreads();
    }
    const int RW = 2 * QP + 8;
    const double* tres = res + (2 * team) * RW;
    const double w = tres[2 * QP] + tres[RW + 2 * QP];
    if (tr < Q && valid) {
      const int q = tr;
      const double mu = mun[q], dq = dn[q];
      const double a1 = tres[q] + tres[RW + q];
      const double a23 = tres[QP + q] + tres[RW + QP + q];
      const double dmu = 2.0 * dq * (a1 - mu * w);
      const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
      const double dS = -2.0 * dq * dq * dd;
      if (accumulate) {
        gmu_g[n * QP + q] += dmu;
        gS_g[n * QP + q] += dS;
      } else {
        gmu_g[n * QP + q] = dmu;
        gS_g[n * QP + q] = dS;
      }
      const double il = 1.0 / lam_g[q];
      dlam_acc += dd * (dq * il) * (dq * il) - 0.5 * w * il;
    } else if (tr >= Q && tr < QP && valid && !accumulate) {
      gmu_g[n * QP + tr] = 0.0;
      gS_g[n * QP + tr] = 0.0;
    }
    if (tr == 0 && valid) wsum_acc += w;
    __syncthreads();
  }

  {
    double* out = p2_part + (size_t)blockIdx.x * tri;
    for (int i = t; i < tri; i += PSI2_T4) out[i] = p2acc[i];
    double* outz = dz_part + (size_t)blockIdx.x * QP * RGP_PSI2_TEAM;
    for (int i = t; i < QP * RGP_PSI2_TEAM; i += PSI2_T4) outz[i] = dzacc[i];
  }
  if (tr < QP) dlam_part[((size_t)blockIdx.x * 4 + team) * QP + tr] = (tr < Q) ? dlam_acc : 0.0;
  if (tr == 0) wsum_part[(size_t)blockIdx.x * 4 + team] = wsum_acc;
}
#undef zrow

// LDS bytes of the four-team kernel
static size_t psi2_rows4_lds(int M, int QP) {
  const size_t tri = (size_t)M * (M + 1) / 2;
  return (tri + RGP_PSI2_TEAM + (size_t)QP * RGP_PSI2_TEAM + 4 * RGP_PSI2_TEAM + 8 * RED_WAVE + 8 * (2 * (size_t)QP + 8)) * sizeof(double);
}

bool psi2_use_rows4(int M, int QP) {
  static const bool off = getenv("RGP_PSI2_NO_ROWS4") != nullptr;
  return !off && psi2_rows4_lds(M, QP) <= 160 * 1024;
}

Psi2Plan psi2_plan(rgp_handle h, int64_t N, int M, int QP, bool fwd, bool bwd) {
  Psi2Plan pl;
  if (fwd && bwd && psi2_use_rows4(M, QP)) {
    int64_t g = h->num_cu;                         // one 512-thread workgroup per CU
    if (g > (N + 3) / 4) g = (N + 3) / 4;
    pl.grid = (int)(g < 1 ? 1 : g);
    pl.n_p2 = pl.grid; pl.n_dz = pl.grid; pl.n_dl = 4 * pl.grid;
  } else {
    pl.grid = (int)psi2_num_blocks(h, N);
    pl.n_p2 = pl.grid; pl.n_dz = 2 * pl.grid; pl.n_dl = 2 * pl.grid;
  }
  return pl;
}

size_t psi2_num_blocks(rgp_handle h, int64_t N) {
  // two 256-thread workgroups per CU (LDS holds one packed M x M accumulator each; 2 waves per SIMD)
  int64_t b = 2 * (int64_t)h->num_cu;
  if (b > (N + 1) / 2) b = (N + 1) / 2;
  return (size_t)(b < 1 ? 1 : b);
}

template <int QP>
static hipError_t psi2_dispatch(hipStream_t s, bool fwd, bool bwd, const Psi2Args& a, int nblk) {
  const int tri = a.M * (a.M + 1) / 2;
  size_t lds = (size_t)2 * RGP_PSI2_TEAM;
  if (fwd) lds += tri + RGP_PSI2_TEAM;
  if (bwd) lds += 4 * RED_WAVE + 4 * (2 * QP + 8);
  lds *= sizeof(double);
#define PSI2_LAUNCH(F, B)                                                                                             \
  do {                                                                                                                \
    hipError_t e = hipFuncSetAttribute((const void*)psi2_rows_kernel<QP, F, B>,                                       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
    if (e != hipSuccess) return e;                                                                                    \
    hipLaunchKernelGGL((psi2_rows_kernel<QP, F, B>), dim3(nblk), dim3(PSI2_THREADS), lds, s, a.N, a.M, a.Q, a.Zc, a.ZcT, \
                       a.lam, a.dLs, a.Xmu, a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part,        \
                       a.o.gmu, a.o.gS, a.o.accumulate);                                                              \
  } while (0)
  if (fwd && bwd && psi2_use_rows4(a.M, QP)) {
    const size_t l4 = psi2_rows4_lds(a.M, QP);
    hipError_t e = hipFuncSetAttribute((const void*)psi2_rows4_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((psi2_rows4_kernel<QP>), dim3(nblk), dim3(PSI2_T4), l4, s, a.N, a.M, a.Q, a.Zc, a.lam, a.dLs, a.Xmu, a.d2,
                       a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu, a.o.gS, a.o.accumulate);
    return hipSuccess;
  }
  if (fwd && bwd) PSI2_LAUNCH(true, true);
  else if (fwd) PSI2_LAUNCH(true, false);
  else PSI2_LAUNCH(false, true);
#undef PSI2_LAUNCH
  return hipSuccess;
}

rgp_status launch_psi2(rgp_handle h, bool fwd, bool bwd, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt,
                       const double* dLs, const Psi2Out& out, int nblk) {
  if (M > RGP_MAX_M) return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2: M > 128 is not supported by this build");
  if (zt.QP > RGP_MAX_QP) return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2: Q > 64 is not supported by this build");
  Psi2Args a{};
  a.N = N; a.M = M; a.Q = Q; a.Zc = zt.Zc; a.ZcT = zt.ZcT; a.zd = zt.zd; a.lam = zt.lam; a.dLs = dLs;
  a.Xmu = rt.Xmu; a.d2 = rt.d2; a.g0 = rt.g0; a.o = out;
  hipStream_t s = h->stream;
  hipError_t e = hipSuccess;
  switch (zt.QP) {
    case 8:  e = psi2_dispatch<8>(s, fwd, bwd, a, nblk); break;
    case 16: e = psi2_dispatch<16>(s, fwd, bwd, a, nblk); break;
    case 24: e = psi2_dispatch<24>(s, fwd, bwd, a, nblk); break;
    case 32: e = psi2_dispatch<32>(s, fwd, bwd, a, nblk); break;
    case 40: e = psi2_dispatch<40>(s, fwd, bwd, a, nblk); break;
    case 48: e = psi2_dispatch<48>(s, fwd, bwd, a, nblk); break;
    case 56: e = psi2_dispatch<56>(s, fwd, bwd, a, nblk); break;
    case 64: e = psi2_dispatch<64>(s, fwd, bwd, a, nblk); break;
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2: unsupported padded Q");
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 launch: ") + hipGetErrorString(e));
  return RGP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// deterministic reduction of per-block partials
// ------------------------------------------------------------------------------------------------------------------
// 32 elements x 8 partial groups per workgroup: group g sums partials g, g+8, ... in order, then the 8 group sums are
// added in fixed order -> deterministic and 8x more parallel than one thread per element.
__global__ void __launch_bounds__(256) reduce_kernel(int nblk, int64_t len, int64_t stride, const double* __restrict__ part,
                                                      double* __restrict__ out, double scale, int accumulate) {
  __shared__ double sm[8][33];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  double s = 0.0;
  if (i < len) {
    for (int b = g; b < nblk; b += 8) s += part[(size_t)b * stride + i];
  }
  sm[g][e] = s;
  __syncthreads();
  if (g == 0 && i < len) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += sm[k][e];
    tot *= scale;
    out[i] = accumulate ? out[i] + tot : tot;
  }
}

void launch_reduce(hipStream_t s, int nblk, int64_t len, int64_t stride, const double* part, double* out, double scale,
                   int accumulate) {
  const int blocks = (int)((len + 31) / 32);
  hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(256), 0, s, nblk, len, stride, part, out, scale, accumulate);
}

__global__ void __launch_bounds__(256) unpack_tri_kernel(int M, const double* __restrict__ tri, double* __restrict__ full) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * M) return;
  const int r = i / M, c = i % M;
  const int hi = r > c ? r : c, lo = r > c ? c : r;
  full[i] = tri[hi * (hi + 1) / 2 + lo];
}

void launch_unpack_tri(hipStream_t s, int M, const double* tri, double* full) {
  hipLaunchKernelGGL(unpack_tri_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, tri, full);
}
