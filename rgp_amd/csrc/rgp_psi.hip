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
                                                      double* lam, double* Zc, double* zd, double* Kuu) {
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
  __syncthreads();
  const double var = variance[0];
  for (int i = t; i < M * M; i += blockDim.x) {
    const int a = i / M, b = i % M;
    double r = 0.0;
    for (int q = 0; q < Q; ++q) {
      const double d = Zc[a * QP + q] - Zc[b * QP + q];
      r += lam[q] * d * d;
    }
    zd[i] = -0.25 * r;
    Kuu[i] = var * exp(-0.5 * r) + (a == b ? jitter : 0.0);
  }
}

void launch_prep_z(hipStream_t s, int M, int Q, int QP, const double* Z, const double* variance, const double* inv_l,
                   double jitter, const ZTerms& zt) {
  hipLaunchKernelGGL(prep_z_kernel, dim3(1), dim3(256), 0, s, M, Q, QP, Z, variance, inv_l, jitter, zt.c, zt.lam, zt.Zc,
                     zt.zd, zt.Kuu);
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
  double *Xmu, *d2, *e1, *g0, *g1;
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
    double sl2 = 0.0, sl1 = 0.0, sdm = 0.0;
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
      }
      a.Xmu[r * a.QP + q] = mu;
      a.d2[r * a.QP + q] = d2;
      a.e1[r * a.QP + q] = e1;
    }
    sl2 = wave_sum(sl2); sl1 = wave_sum(sl1); sdm = wave_sum(sdm);
    if (lane == 0) {
      a.g0[r] = 2.0 * logvar - 0.5 * sl2 - sdm;
      a.g1[r] = logvar - 0.5 * sl1;
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
  a.Xmu = rt.Xmu; a.d2 = rt.d2; a.e1 = rt.e1; a.g0 = rt.g0; a.g1 = rt.g1;
  hipLaunchKernelGGL(rowprep_kernel, dim3(rowprep_blocks(N)), dim3(256), 0, s, a);
}

// yry_out must hold 2048 doubles of partials followed by nothing else; caller reduces with launch_reduce.
void launch_rowprep_window(hipStream_t s, int64_t N, int Q, const WindowSpec& w, const double* variance,
                           const ZTerms& zt, const RowTerms& rt, double* Ym, double* Yv, double* yry_part) {
  RowPrepArgs a{};
  a.N = N; a.Q = Q; a.QP = zt.QP; a.mu = nullptr; a.S = nullptr; a.w = w; a.variance = variance; a.c = zt.c;
  a.lam = zt.lam; a.Xmu = rt.Xmu; a.d2 = rt.d2; a.e1 = rt.e1; a.g0 = rt.g0; a.g1 = rt.g1;
  a.Ym = Ym; a.Yv = Yv; a.yry_part = yry_part;
  hipLaunchKernelGGL(rowprep_kernel, dim3(2048), dim3(256), 0, s, a);   // fixed grid: yry_part has 2048 slots
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 forward: one wave per row, lanes over m.  Zc staged in LDS with an odd row stride (conflict-free b64 reads).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) psi1_fwd_kernel(int64_t N, int M, int QP, const double* __restrict__ Zc,
                                                        const double* __restrict__ Xmu, const double* __restrict__ e1,
                                                        const double* __restrict__ g1, double* __restrict__ psi1) {
  extern __shared__ double lds[];
  const int QS = QP + 1;
  for (int i = threadIdx.x; i < M * QP; i += blockDim.x) lds[(i / QP) * QS + (i % QP)] = Zc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < N; r += (int64_t)gridDim.x * wpb) {
    const double* mu = Xmu + r * QP;
    const double* e = e1 + r * QP;
    const double g = g1[r];
    for (int m = lane; m < M; m += 64) {
      double acc = 0.0;
      for (int q = 0; q < QP; ++q) {
        const double d = mu[q] - lds[m * QS + q];
        acc += e[q] * d * d;
      }
      psi1[r * M + m] = exp(g - 0.5 * acc);
    }
  }
}

void launch_psi1_fwd(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, double* psi1) {
  const size_t lds = (size_t)M * (zt.QP + 1) * sizeof(double);
  const int blocks = rowprep_blocks(N);
  hipLaunchKernelGGL(psi1_fwd_kernel, dim3(blocks), dim3(256), lds, s, N, M, zt.QP, zt.Zc, rt.Xmu, rt.e1, rt.g1, psi1);
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
  const double *Zc, *zd, *lam, *dLs;
  const double *Xmu, *d2, *g0;
  Psi2Out o;
};

#define RED_QC 8                 // q columns per reduction chunk (2*RED_QC values x 128 threads)
#define RED_RS 136               // padded row stride of the reduction buffer (conflict-free)

// Every pointer is a separate noalias kernel argument: that is what lets the compiler prove the wave-uniform loads
// of z[m',:], d2[n,:] and mu[n,:] are not clobbered by the kernel's own stores and turn them into scalar loads.
template <int QP, int NB, bool FWD, bool BWD>
__global__ void __launch_bounds__(RGP_PSI2_TEAM, 1)
psi2_rows_kernel(const int64_t N, const int M, const int Q, const double* __restrict__ Zc, const double* __restrict__ zd_g,
                 const double* __restrict__ lam_g, const double* __restrict__ dLs_g, const double* __restrict__ Xmu_g,
                 const double* __restrict__ d2_g, const double* __restrict__ g0_g, double* __restrict__ p2_part,
                 double* __restrict__ dz_part, double* __restrict__ dlam_part, double* __restrict__ wsum_part,
                 double* __restrict__ gmu_g, double* __restrict__ gS_g, const int accumulate) {
  extern __shared__ double lds[];
  struct { int64_t N; int M, Q; const double *zd, *lam, *dLs, *Xmu, *d2, *g0;
           struct { double *p2_part, *dz_part, *dlam_part, *wsum_part, *gmu, *gS; int accumulate; } o; } a;
  a.N = N; a.M = M; a.Q = Q; a.zd = zd_g; a.lam = lam_g; a.dLs = dLs_g; a.Xmu = Xmu_g; a.d2 = d2_g; a.g0 = g0_g;
  a.o.p2_part = p2_part; a.o.dz_part = dz_part; a.o.dlam_part = dlam_part; a.o.wsum_part = wsum_part;
  a.o.gmu = gmu_g; a.o.gS = gS_g; a.o.accumulate = accumulate;
  const int tri = M * (M + 1) / 2;
  double* p2acc = lds;                                   // [tri]           (FWD)
  double* hbuf = lds + (FWD ? tri : 0);                  // [NB x 128]
  double* red = hbuf + NB * RGP_PSI2_TEAM;               // [2*RED_QC x RED_RS]   (BWD)
  double* res = red + (BWD ? 2 * RED_QC * RED_RS : 0);   // [2*QP]                (BWD)
  double* scratch = res + (BWD ? 2 * QP : 0);            // [16]

  const int t = threadIdx.x;
  const bool act = t < M;
  const int tm = act ? t : M - 1;
  const double* __restrict__ zrow = Zc + (size_t)tm * QP;

  if (FWD) {
    for (int i = t; i < tri; i += RGP_PSI2_TEAM) p2acc[i] = 0.0;
  }
  double* dzp = BWD ? a.o.dz_part + (size_t)blockIdx.x * QP * RGP_PSI2_TEAM : nullptr;   // layout [q][t]
  if (BWD) {
#pragma unroll
    for (int q = 0; q < QP; ++q) dzp[q * RGP_PSI2_TEAM + t] = 0.0;
  }
  double dlam_acc = 0.0, wsum_acc = 0.0;
  __syncthreads();

  const int64_t per = ((a.N + gridDim.x - 1) / gridDim.x + NB - 1) / NB * NB;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < a.N) ? r0 + per : a.N;

  for (int64_t row = r0; row < r1; row += NB) {
    double dz[NB][QP];
    double hb[NB], g0b[NB];
    int64_t nrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool valid = row + b < r1;
      const int64_t n = valid ? row + b : r1 - 1;
      nrow[b] = n;
      const double* __restrict__ dn = a.d2 + n * QP;       // wave-uniform -> scalar loads
      const double* __restrict__ mun = a.Xmu + n * QP;
      double h0 = 0.0, h1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; q += 2) {
        const double z0 = zrow[q], z1 = zrow[q + 1];
        dz[b][q] = z0 * dn[q];
        dz[b][q + 1] = z1 * dn[q + 1];
        h0 += dz[b][q] * (mun[q] - 0.25 * z0);
        h1 += dz[b][q + 1] * (mun[q + 1] - 0.25 * z1);
      }
      const double h = h0 + h1;
      hbuf[b * RGP_PSI2_TEAM + t] = h;
      hb[b] = (act && valid) ? h : -INFINITY;              // exp(-inf) = 0: padded lanes / rows contribute nothing
      g0b[b] = a.g0[n];
    }
    __syncthreads();

    double y[NB][QP];
    double v[NB];
    if (BWD) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        v[b] = 0.0;
#pragma unroll
        for (int q = 0; q < QP; ++q) y[b][q] = 0.0;
      }
    }
    double zd_next = a.zd[tm];
    double dl_next = BWD ? a.dLs[tm] : 0.0;
    for (int mp = 0; mp < M; ++mp) {
      const double* __restrict__ zs = Zc + (size_t)mp * QP;    // wave-uniform row of z
      const double zdv = zd_next, dlv = dl_next;
      const int mn = (mp + 1 < M) ? mp + 1 : mp;
      zd_next = a.zd[(size_t)mn * M + tm];
      if (BWD) dl_next = a.dLs[(size_t)mn * M + tm];
      double psum = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
        for (int q = 0; q < QP; q += 4) {
          c0 += dz[b][q] * zs[q];
          c1 += dz[b][q + 1] * zs[q + 1];
          c2 += dz[b][q + 2] * zs[q + 2];
          c3 += dz[b][q + 3] * zs[q + 3];
        }
        const double E = (zdv + g0b[b] + hb[b]) + hbuf[b * RGP_PSI2_TEAM + mp] - 0.5 * ((c0 + c1) + (c2 + c3));
        const double p = exp(E);
        psum += p;
        if (BWD) {
          const double w = p * dlv;
          v[b] += w;
#pragma unroll
          for (int q = 0; q < QP; ++q) y[b][q] += w * zs[q];
        }
      }
      if (FWD) {
        if (act && mp >= t) p2acc[mp * (mp + 1) / 2 + t] += psum;
      }
    }

    if (BWD) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const bool valid = row + b < r1;
        const int64_t n = nrow[b];
        const double* __restrict__ dn = a.d2 + n * QP;
        const double* __restrict__ mun = a.Xmu + n * QP;
        const double vb = valid ? v[b] : 0.0;
        // dZ partial:  v d (2 mu - z) - d y
#pragma unroll
        for (int q = 0; q < QP; ++q) {
          const double z = zrow[q];
          dzp[q * RGP_PSI2_TEAM + t] += dn[q] * (vb * (2.0 * mun[q] - z) - (valid ? y[b][q] : 0.0));
        }
        const double w = block_sum(vb, scratch);
        // a1[q] = sum_m v z ; a23[q] = 1/2 sum_m z (v z + y)
        for (int q0 = 0; q0 < QP; q0 += RED_QC) {
#pragma unroll
          for (int j = 0; j < RED_QC; ++j) {
            const double z = zrow[q0 + j];
            red[j * RED_RS + t] = vb * z;
            red[(RED_QC + j) * RED_RS + t] = 0.5 * z * (vb * z + (valid ? y[b][q0 + j] : 0.0));
          }
          __syncthreads();
          {
            const int jv = t >> 3, part = t & 7;           // 16 values x 8 partial sums
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < RGP_PSI2_TEAM / 8; ++k) s += red[jv * RED_RS + part + 8 * k];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (part == 0) res[(jv < RED_QC ? 0 : QP) + q0 + (jv % RED_QC)] = s;
          }
          __syncthreads();
        }
        if (t < a.Q && valid) {
          const int q = t;
          const double mu = mun[q], dq = dn[q];
          const double a1 = res[q], a23 = res[QP + q];
          const double dmu = 2.0 * dq * (a1 - mu * w);
          const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
          const double dS = -2.0 * dq * dq * dd;
          if (a.o.accumulate) {
            a.o.gmu[n * QP + q] += dmu;
            a.o.gS[n * QP + q] += dS;
          } else {
            a.o.gmu[n * QP + q] = dmu;
            a.o.gS[n * QP + q] = dS;
          }
          const double il = 1.0 / a.lam[q];
          dlam_acc += dd * (dq * il) * (dq * il) - 0.5 * w * il;
        } else if (t >= a.Q && t < QP && valid && !a.o.accumulate) {
          a.o.gmu[n * QP + t] = 0.0;
          a.o.gS[n * QP + t] = 0.0;
        }
        if (t == 0) wsum_acc += w;
        __syncthreads();
      }
    } else {
      __syncthreads();      // hbuf is rewritten at the top of the next iteration
    }
  }

  if (FWD) {
    __syncthreads();
    double* out = a.o.p2_part + (size_t)blockIdx.x * tri;
    for (int i = t; i < tri; i += RGP_PSI2_TEAM) out[i] = p2acc[i];
  }
  if (BWD) {
    if (t < QP) a.o.dlam_part[(size_t)blockIdx.x * QP + t] = (t < a.Q) ? dlam_acc : 0.0;
    if (t == 0) a.o.wsum_part[blockIdx.x] = wsum_acc;
  }
}

size_t psi2_num_blocks(rgp_handle h, int64_t N) {
  // two 128-thread workgroups per CU (LDS holds one packed M x M accumulator each)
  int64_t b = 2 * (int64_t)h->num_cu;
  if (b > (N + 1) / 2) b = (N + 1) / 2;
  return (size_t)(b < 1 ? 1 : b);
}

template <int QP, int NB>
static void psi2_dispatch(hipStream_t s, bool fwd, bool bwd, const Psi2Args& a, int nblk) {
  const int tri = a.M * (a.M + 1) / 2;
  size_t lds = (size_t)NB * RGP_PSI2_TEAM + 16;
  if (fwd) lds += tri;
  if (bwd) lds += 2 * RED_QC * RED_RS + 2 * QP;
  lds *= sizeof(double);
  if (fwd && bwd) {
    hipLaunchKernelGGL((psi2_rows_kernel<QP, NB, true, true>), dim3(nblk), dim3(RGP_PSI2_TEAM), lds, s, a.N, a.M, a.Q, a.Zc, a.zd, a.lam,
                       a.dLs, a.Xmu, a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu, a.o.gS,
                       a.o.accumulate);
  } else if (fwd) {
    hipLaunchKernelGGL((psi2_rows_kernel<QP, NB, true, false>), dim3(nblk), dim3(RGP_PSI2_TEAM), lds, s, a.N, a.M, a.Q, a.Zc, a.zd, a.lam,
                       a.dLs, a.Xmu, a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu, a.o.gS,
                       a.o.accumulate);
  } else {
    hipLaunchKernelGGL((psi2_rows_kernel<QP, NB, false, true>), dim3(nblk), dim3(RGP_PSI2_TEAM), lds, s, a.N, a.M, a.Q, a.Zc, a.zd, a.lam,
                       a.dLs, a.Xmu, a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu, a.o.gS,
                       a.o.accumulate);
  }
}

rgp_status launch_psi2(rgp_handle h, bool fwd, bool bwd, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt,
                       const double* dLs, const Psi2Out& out, int nblk) {
  if (M > RGP_MAX_M) return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2: M > 128 is not supported by this build");
  if (zt.QP > RGP_MAX_QP) return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2: Q > 64 is not supported by this build");
  Psi2Args a{};
  a.N = N; a.M = M; a.Q = Q; a.Zc = zt.Zc; a.zd = zt.zd; a.lam = zt.lam; a.dLs = dLs;
  a.Xmu = rt.Xmu; a.d2 = rt.d2; a.g0 = rt.g0; a.o = out;
  hipStream_t s = h->stream;
  // rows per thread-pass: two rows share one sweep over z[m'] only while dz+y for both fit the 256 arch VGPRs
  static const int nb_env = getenv("RGP_PSI2_NB") ? atoi(getenv("RGP_PSI2_NB")) : 0;
  const int nb = nb_env ? nb_env : (zt.QP <= 16 ? 2 : 1);
#define PSI2_CASE(qp)                                                         \
  case qp:                                                                    \
    if (nb == 2 && qp <= 40) psi2_dispatch<qp, (qp <= 40 ? 2 : 1)>(s, fwd, bwd, a, nblk); \
    else psi2_dispatch<qp, 1>(s, fwd, bwd, a, nblk);                          \
    break;
  switch (zt.QP) {
    PSI2_CASE(8) PSI2_CASE(16) PSI2_CASE(24) PSI2_CASE(32) PSI2_CASE(40) PSI2_CASE(48) PSI2_CASE(56) PSI2_CASE(64)
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2: unsupported padded Q");
  }
#undef PSI2_CASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 launch: ") + hipGetErrorString(e));
  return RGP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// deterministic reduction of per-block partials
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_kernel(int nblk, int64_t len, int64_t stride, const double* __restrict__ part,
                                                      double* __restrict__ out, double scale, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  double s = 0.0;
  for (int g = 0; g < nblk; ++g) s += part[(size_t)g * stride + i];
  s *= scale;
  out[i] = accumulate ? out[i] + s : s;
}

void launch_reduce(hipStream_t s, int nblk, int64_t len, int64_t stride, const double* part, double* out, double scale,
                   int accumulate) {
  const int blocks = (int)((len + 255) / 256);
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
