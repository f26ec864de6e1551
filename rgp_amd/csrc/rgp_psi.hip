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
#include <stdio.h>
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
  const int a = blockIdx.x;
  for (int b = threadIdx.x; b < M; b += blockDim.x) {
    double r = 0.0;
    for (int q = 0; q < Q; ++q) {
      const double d = Zc[(size_t)a * QP + q] - Zc[(size_t)b * QP + q];
      r += lam[q] * d * d;
    }
    zd[(size_t)a * M + b] = -0.25 * r;
    Kuu[(size_t)a * M + b] = variance[0] * exp(-0.5 * r) + (a == b ? jitter : 0.0);
  }
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
  // each wave takes a contiguous range of rows: the sequence lookup is one binary search plus increments
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  const int64_t rpw = (a.N + nwaves - 1) / nwaves;
  const int64_t rbeg = ((int64_t)blockIdx.x * wpb + wave) * rpw;
  const int64_t rend = (rbeg + rpw < a.N) ? rbeg + rpw : a.N;
  int s = 0;
  int64_t seq_end = 0, seq_beg = 0, Ns = 0, own_off_s = 0, up_base_s = 0;
  if (!dense && rbeg < rend) {
    int lo = 0, hi = a.w.n_seq;                      // largest s with row_off[s] <= rbeg
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.w.row_off[mid] <= rbeg) lo = mid; else hi = mid;
    }
    s = lo - 1;                                      // the loop below advances into sequence lo
    seq_end = a.w.row_off[lo];
  }
  for (int64_t r = rbeg; r < rend; ++r) {
    int64_t nloc = r, own_base = 0, up_base = 0;
    if (!dense) {
      while (r >= seq_end) {                         // next non-empty sequence
        ++s;
        seq_beg = a.w.row_off[s];
        seq_end = a.w.row_off[s + 1];
        Ns = a.w.own_len[s] - a.w.X_win;
        own_off_s = a.w.own_off[s];
        up_base_s = (a.w.U_win > 0) ? a.w.up_off[s] + (a.w.up_len[s] - Ns - a.w.U_win + 1) * a.w.U_dim : 0;
      }
      nloc = r - seq_beg;
      own_base = own_off_s + nloc * a.w.X_dim;                                // element where the own window starts
      up_base = up_base_s + nloc * a.w.U_dim;
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
    sl2 = wave_sum_dpp(sl2); sl1 = wave_sum_dpp(sl1); sdm = wave_sum_dpp(sdm); sem = wave_sum_dpp(sem);
    if (lane == 0) {
      a.g0[r] = 2.0 * logvar - 0.5 * sl2 - sdm;
      a.g1[r] = logvar - 0.5 * sl1 - 0.5 * sem;          // includes -1/2 sum e mu^2 (the psi1 exponent is expanded)
    }
    if (!dense && a.Ym) {                              // outputs of this row: own[X_win + n]
      const int64_t ybase = own_off_s + (a.w.X_win + nloc) * a.w.D;
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
    P1F(8) P1F(16) P1F(20) P1F(24) P1F(32) P1F(40) P1F(48) P1F(56) P1F(64)
#undef P1F
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 backward (row part): one wave per group of 4 rows.
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
  double dl_acc = 0.0;                      // lane q (QP <= 64)
  double ls_acc = 0.0;
  // A wave takes PB rows at a time: the Lp values of a row live in two registers per lane (m = lane, lane + 64) and are
  // broadcast with v_readlane, so one LDS read of z[m, q] serves PB rows and the PB accumulation chains are independent.
  constexpr int PB = 4;
  const int64_t ngroups = (N + PB - 1) / PB;
  for (int64_t gidx = (int64_t)blockIdx.x * wpb + wave; gidx < ngroups; gidx += (int64_t)gridDim.x * wpb) {
    const int64_t rb = gidx * PB;
    double l0[PB], l1[PB], s0[PB], s1[PB], s2[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const int64_t r = rb + b;
      const bool ok = r < N;
      l0[b] = (ok && lane < M) ? dLdpsi1[r * M + lane] * psi1[r * M + lane] : 0.0;
      l1[b] = (ok && lane + 64 < M) ? dLdpsi1[r * M + lane + 64] * psi1[r * M + lane + 64] : 0.0;
      if (ok && lane < M) Lp[r * M + lane] = l0[b];
      if (ok && lane + 64 < M) Lp[r * M + lane + 64] = l1[b];
      s0[b] = wave_sum(l0[b] + l1[b]);
      s1[b] = 0.0; s2[b] = 0.0;
    }
    ls_acc += (s0[0] + s0[1]) + (s0[2] + s0[3]);
    const int q = lane < QP ? lane : QP - 1;
    const int Ma = M < 64 ? M : 64;
    for (int m = 0; m < Ma; ++m) {
      const double z = zs[m * QP + q], zz = z * z;
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const double l = readlane_f64(l0[b], m);
        s1[b] += l * z;
        s2[b] += l * zz;
      }
    }
    for (int m = 64; m < M; ++m) {
      const double z = zs[m * QP + q], zz = z * z;
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const double l = readlane_f64(l1[b], m - 64);
        s1[b] += l * z;
        s2[b] += l * zz;
      }
    }
    if (lane < QP) {
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const int64_t r = rb + b;
        if (r < N) {
          const double mu = Xmu[r * QP + q], e = e1[r * QP + q];
          const double S2 = mu * mu * s0[b] - 2.0 * mu * s1[b] + s2[b];
          gmu[r * QP + q] = -e * (mu * s0[b] - s1[b]);
          gS[r * QP + q] = 0.5 * e * e * S2 - 0.5 * e * s0[b];
          if (q < Q) {
            const double il = 1.0 / lam[q];
            const double el = e * il;
            dl_acc += -0.5 * el * el * S2 - 0.5 * s0[b] * (1.0 - el) * il;
          }
        }
      }
    }
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
// psi1 side of the SVI bound in three passes (svi_vardtc.py:150-200 + GPy psi_comp rbf_psi_comp.py psi1 terms).
// dL/dpsi1 = Ym Cpsi1^T does not depend on the statistics, so the forward value never has to be re-read:
//   pass F (thread = inducing row m):  P = psi1, Lp = P * (Ym Cpsi1^T) (stored), psi1Y = Ym^T P (per-workgroup partials)
//   pass R (thread = data row n):      s0, s1[q], s2[q] = sum_m Lp {1, z, z^2}  -> gmu, gS, dinv_l;  dYm = P Cpsi1 - beta Ym
//   pass Z (thread = inducing row m):  dZ[m,q] = sum_n Lp[n,m] (e mu - e z)[n,q]
// Each layout makes one of the two reductions thread-local.  Row operands that are uniform over a workgroup (e, e*mu, Ym)
// are staged in LDS and read as broadcasts: 2*QP doubles per row do not fit the SGPR file (the scalar-operand version of
// the forward spilled 105 SGPRs per row through v_writelane and ran 5 load-wait phases per row; a spill-free variant with
// two half-row scalar loads per row measured 475 us against 165 us for the LDS version: one row per wave-iteration leaves
// the scalar-load latency exposed).
// ------------------------------------------------------------------------------------------------------------------
#define P1_TILE 32                 // rows staged per tile, 16 per team of 128 threads

template <int QP, int DP>
__global__ void __launch_bounds__(256, 2)
psi1_svi_fwd_kernel(const int64_t N, const int M, const int D, const double* __restrict__ Zc, const double* __restrict__ e1,
                    const double* __restrict__ em, const double* __restrict__ g1, const double* __restrict__ Ym,
                    const double* __restrict__ Cpsi1, double* __restrict__ P, double* __restrict__ Lp,
                    double* __restrict__ psi1Y_part) {
  constexpr int EE = P1_TILE * QP;                                  // double2 entries
  constexpr int CB = DP * RGP_PSI2_TEAM;                            // doubles needed to combine the teams
  constexpr int NB = (2 * EE > CB) ? 2 * EE : CB;
  __shared__ __attribute__((aligned(16))) double buf[NB];
  __shared__ double ys[P1_TILE * DP];
  __shared__ double gs[P1_TILE];
  double2* ee = reinterpret_cast<double2*>(buf);
  const int t = threadIdx.x, tr = t & (RGP_PSI2_TEAM - 1), team = t >> 7;
  const bool act = tr < M;
  const int tm = act ? tr : M - 1;
  double z[QP], c[DP], py[DP];
#pragma unroll
  for (int q = 0; q < QP; ++q) z[q] = Zc[(size_t)tm * QP + q];
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    c[d] = d < D ? Cpsi1[(size_t)tm * D + d] : 0.0;
    py[d] = 0.0;
  }
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + P1_TILE - 1) / P1_TILE * P1_TILE;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  for (int64_t rb = r0; rb < r1; rb += P1_TILE) {
    const int nr = (int)((r1 - rb) < P1_TILE ? (r1 - rb) : P1_TILE);
    __syncthreads();
    {                                             // branch-free staging: all loads of the tile are in flight together
      const int64_t lim = (int64_t)nr * QP - 1;
      constexpr int NE = (P1_TILE * QP + 255) / 256;            // (e, e*mu) entries per thread
      double2 v[NE];
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int64_t i = t + 256 * k, ic = i < lim ? i : lim;
        v[k] = make_double2(e1[rb * QP + ic], em[rb * QP + ic]);
      }
      const int yr = (t / DP < nr) ? t / DP : nr - 1, yd = t % DP;
      const double yv = (t < P1_TILE * DP && yd < D) ? Ym[(rb + yr) * D + (yd < D ? yd : 0)] : 0.0;
      const double gv = g1[rb + (t < nr ? t : nr - 1)];
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if (t + 256 * k < P1_TILE * QP) ee[t + 256 * k] = v[k];
      if (t < P1_TILE * DP) ys[t] = yv;
      if (DP > 8 && t + 256 < P1_TILE * DP) {
        const int i2 = t + 256, r2 = (i2 / DP < nr) ? i2 / DP : nr - 1, d2 = i2 % DP;
        ys[i2] = d2 < D ? Ym[(rb + r2) * D + d2] : 0.0;
      }
      if (t < P1_TILE) gs[t] = gv;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = team; r < nr; r += 2) {
      const double2* er = ee + r * QP;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; q += 2) {
        const double2 x0 = er[q], x1 = er[q + 1];
        a0 += z[q] * (x0.x * z[q] - 2.0 * x0.y);
        a1 += z[q + 1] * (x1.x * z[q + 1] - 2.0 * x1.y);
      }
      const double p = exp(gs[r] - 0.5 * (a0 + a1));
      double dl = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double y = ys[r * DP + d];
        dl += y * c[d];
        py[d] += y * p;
      }
      if (act) {
        Lp[(rb + r) * M + tr] = dl * p;
        if (P) P[(rb + r) * M + tr] = p;
      }
    }
  }
  // psi1Y partial of this workgroup: team 0 + team 1 (fixed order)
  __syncthreads();
  if (team == 1) {
#pragma unroll
    for (int d = 0; d < DP; ++d) buf[d * RGP_PSI2_TEAM + tr] = py[d];
  }
  __syncthreads();
  if (team == 0 && act) {
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < D) psi1Y_part[((size_t)blockIdx.x * D + d) * M + tr] = py[d] + buf[d * RGP_PSI2_TEAM + tr];
  }
}

// pass F with two inducing rows per lane (m, m + 64): one wave covers all M <= 128 rows of a data row, the four waves of
// the workgroup take different data rows of the tile, and each broadcast ds_read_b128 of (e, e*mu) feeds four multiply-adds
// instead of two (the one-row version is bound by those reads).  Used when the D-sized register arrays are small (DP <= 4).
template <int QP, int DP>
__global__ void __launch_bounds__(256, 2)
psi1_svi_fwd2_kernel(const int64_t N, const int M, const int D, const double* __restrict__ Zc, const double* __restrict__ e1,
                     const double* __restrict__ em, const double* __restrict__ g1, const double* __restrict__ Ym,
                     const double* __restrict__ Cpsi1, double* __restrict__ P, double* __restrict__ Lp,
                     double* __restrict__ psi1Y_part) {
  constexpr int EE = P1_TILE * QP;                                  // double2 entries
  constexpr int CB = 3 * DP * RGP_PSI2_TEAM;                        // doubles needed to combine the four waves
  constexpr int NB = (2 * EE > CB) ? 2 * EE : CB;
  __shared__ __attribute__((aligned(16))) double buf[NB];
  __shared__ double ys[P1_TILE * DP];
  __shared__ double gs[P1_TILE];
  __shared__ double etab[PSI2_EXP_TAB];
  double2* ee = reinterpret_cast<double2*>(buf);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  const int m0 = lane, m1 = lane + 64;
  const bool act0 = m0 < M, act1 = m1 < M;
  const int c0 = act0 ? m0 : M - 1, c1 = act1 ? m1 : M - 1;
  double z0[QP], z1[QP], cA[DP], cB[DP], pyA[DP], pyB[DP];
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    z0[q] = Zc[(size_t)c0 * QP + q];
    z1[q] = Zc[(size_t)c1 * QP + q];
  }
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    cA[d] = d < D ? Cpsi1[(size_t)c0 * D + d] : 0.0;
    cB[d] = d < D ? Cpsi1[(size_t)c1 * D + d] : 0.0;
    pyA[d] = 0.0;
    pyB[d] = 0.0;
  }
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + P1_TILE - 1) / P1_TILE * P1_TILE;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  for (int64_t rb = r0; rb < r1; rb += P1_TILE) {
    const int nr = (int)((r1 - rb) < P1_TILE ? (r1 - rb) : P1_TILE);
    __syncthreads();
    {
      const int64_t lim = (int64_t)nr * QP - 1;
      constexpr int NE = (P1_TILE * QP + 255) / 256;            // (e, e*mu) entries per thread
      double2 v[NE];
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int64_t i = t + 256 * k, ic = i < lim ? i : lim;
        v[k] = make_double2(e1[rb * QP + ic], em[rb * QP + ic]);
      }
      const int yr = (t / DP < nr) ? t / DP : nr - 1, yd = t % DP;
      const double yv = (t < P1_TILE * DP && yd < D) ? Ym[(rb + yr) * D + (yd < D ? yd : 0)] : 0.0;
      const double gv = g1[rb + (t < nr ? t : nr - 1)];
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if (t + 256 * k < P1_TILE * QP) ee[t + 256 * k] = v[k];
      if (t < P1_TILE * DP) ys[t] = yv;
      if (t < P1_TILE) gs[t] = gv;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = wave; r < nr; r += 4) {
      const double2* er = ee + r * QP;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const double2 x = er[q];
        a0 += z0[q] * (x.x * z0[q] - 2.0 * x.y);
        a1 += z1[q] * (x.x * z1[q] - 2.0 * x.y);
      }
      const double g = gs[r];
      const double p0 = exp_tab(g - 0.5 * a0, etab), p1 = exp_tab(g - 0.5 * a1, etab);
      double dl0 = 0.0, dl1 = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double y = ys[r * DP + d];
        dl0 += y * cA[d];
        dl1 += y * cB[d];
        pyA[d] += y * p0;
        pyB[d] += y * p1;
      }
      if (act0) {
        Lp[(rb + r) * M + m0] = dl0 * p0;
        if (P) P[(rb + r) * M + m0] = p0;
      }
      if (act1) {
        Lp[(rb + r) * M + m1] = dl1 * p1;
        if (P) P[(rb + r) * M + m1] = p1;
      }
    }
  }
  // psi1Y partial of this workgroup: wave 0 + 1 + 2 + 3 (fixed order)
  __syncthreads();
  if (wave > 0) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      buf[((wave - 1) * DP + d) * RGP_PSI2_TEAM + m0] = pyA[d];
      buf[((wave - 1) * DP + d) * RGP_PSI2_TEAM + m1] = pyB[d];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      if (d < D) {
        double sA = pyA[d], sB = pyB[d];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          sA += buf[(w * DP + d) * RGP_PSI2_TEAM + m0];
          sB += buf[(w * DP + d) * RGP_PSI2_TEAM + m1];
        }
        if (act0) psi1Y_part[((size_t)blockIdx.x * D + d) * M + m0] = sA;
        if (act1) psi1Y_part[((size_t)blockIdx.x * D + d) * M + m1] = sB;
      }
    }
  }
}

// pass F for wide outputs (D > 16): the D-sized register arrays of the kernels above do not scale, so the two contractions
// over d run as GEMMs (T = Ym Cpsi1^T before this kernel, psi1Y = Ym^T P after it) and the kernel only evaluates
// P = psi1 and Lp = P o T in place (Lp holds T on entry).  Same staging of the row operands as psi1_svi_fwd_kernel.
template <int QP>
__global__ void __launch_bounds__(256, 2)
psi1_svi_fwdT_kernel(const int64_t N, const int M, const double* __restrict__ Zc, const double* __restrict__ e1,
                     const double* __restrict__ em, const double* __restrict__ g1, double* __restrict__ P, double* Lp) {
  __shared__ __attribute__((aligned(16))) double buf[2 * P1_TILE * QP];
  __shared__ double gs[P1_TILE];
  double2* ee = reinterpret_cast<double2*>(buf);
  const int t = threadIdx.x, tr = t & (RGP_PSI2_TEAM - 1), team = t >> 7;
  const bool act = tr < M;
  const int tm = act ? tr : M - 1;
  double z[QP];
#pragma unroll
  for (int q = 0; q < QP; ++q) z[q] = Zc[(size_t)tm * QP + q];
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + P1_TILE - 1) / P1_TILE * P1_TILE;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  for (int64_t rb = r0; rb < r1; rb += P1_TILE) {
    const int nr = (int)((r1 - rb) < P1_TILE ? (r1 - rb) : P1_TILE);
    __syncthreads();
    {
      const int64_t lim = (int64_t)nr * QP - 1;
      constexpr int NE = (P1_TILE * QP + 255) / 256;
      double2 v[NE];
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int64_t i = t + 256 * k, ic = i < lim ? i : lim;
        v[k] = make_double2(e1[rb * QP + ic], em[rb * QP + ic]);
      }
      const double gv = g1[rb + (t < nr ? t : nr - 1)];
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if (t + 256 * k < P1_TILE * QP) ee[t + 256 * k] = v[k];
      if (t < P1_TILE) gs[t] = gv;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = team; r < nr; r += 2) {
      const double2* er = ee + r * QP;
      const double tv = act ? Lp[(rb + r) * M + tr] : 0.0;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; q += 2) {
        const double2 x0 = er[q], x1 = er[q + 1];
        a0 += z[q] * (x0.x * z[q] - 2.0 * x0.y);
        a1 += z[q + 1] * (x1.x * z[q + 1] - 2.0 * x1.y);
      }
      const double p = exp(gs[r] - 0.5 * (a0 + a1));
      if (act) {
        Lp[(rb + r) * M + tr] = tv * p;
        if (P) P[(rb + r) * M + tr] = p;
      }
    }
  }
}

// pass Z.  Lane = two inducing rows (m, m + 64), wave = one quarter of the q columns; all four waves walk all rows of the
// tile.  One broadcast ds_read_b128 of (e, e*mu)[row, q] then feeds four multiply-adds instead of two, and a (m, q) slot has
// exactly one owner, so nothing is combined at the end.  The Lp tile is staged in LDS (coalesced 128-column rows).
#ifndef P1Z_TILE
#define P1Z_TILE 16                // rows per tile of pass Z: 26 KB of LDS per workgroup at QP = 40, six workgroups per CU
#endif
template <int QP>
__global__ void __launch_bounds__(256)
psi1_svi_dz_kernel(const int64_t N, const int M, const double* __restrict__ Zc, const double* __restrict__ e1,
                   const double* __restrict__ em, const double* __restrict__ Lp, double* __restrict__ dz_part) {
  constexpr int QW = QP / 4;                               // QP is a multiple of 4
  __shared__ __attribute__((aligned(16))) double buf[2 * P1Z_TILE * QP];
  __shared__ double lps[P1Z_TILE * RGP_PSI2_TEAM];
  double2* ee = reinterpret_cast<double2*>(buf);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q0 = wave * QW;
  const int m0 = lane, m1 = lane + 64;
  const int c0 = m0 < M ? m0 : M - 1, c1 = m1 < M ? m1 : M - 1;
  double z0[QW], z1[QW], a0[QW], a1[QW];
#pragma unroll
  for (int j = 0; j < QW; ++j) {
    z0[j] = Zc[(size_t)c0 * QP + q0 + j];
    z1[j] = Zc[(size_t)c1 * QP + q0 + j];
    a0[j] = 0.0;
    a1[j] = 0.0;
  }
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + P1Z_TILE - 1) / P1Z_TILE * P1Z_TILE;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  for (int64_t rb = r0; rb < r1; rb += P1Z_TILE) {
    const int nr = (int)((r1 - rb) < P1Z_TILE ? (r1 - rb) : P1Z_TILE);
    {
      const int64_t lim = (int64_t)nr * QP - 1;
      constexpr int NE = (P1Z_TILE * QP + 255) / 256;          // (e, e*mu) entries per thread
      double2 v[NE];
      double lv[P1Z_TILE * RGP_PSI2_TEAM / 256];
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int64_t i = t + 256 * k, ic = i < lim ? i : lim;
        v[k] = make_double2(e1[rb * QP + ic], em[rb * QP + ic]);
      }
#pragma unroll
      for (int k = 0; k < P1Z_TILE * RGP_PSI2_TEAM / 256; ++k) {
        const int i = t + 256 * k, row = i >> 7, col = i & 127;
        const bool ok = row < nr && col < M;
        const double x = Lp[(rb + (ok ? row : 0)) * M + (ok ? col : 0)];
        lv[k] = ok ? x : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if (t + 256 * k < P1Z_TILE * QP) ee[t + 256 * k] = v[k];
#pragma unroll
      for (int k = 0; k < P1Z_TILE * RGP_PSI2_TEAM / 256; ++k) lps[t + 256 * k] = lv[k];
    }
    __syncthreads();
#pragma unroll 2
    for (int r = 0; r < nr; ++r) {
      const double2* er = ee + r * QP + q0;
      const double l0 = lps[r * RGP_PSI2_TEAM + m0], l1 = lps[r * RGP_PSI2_TEAM + m1];
#pragma unroll
      for (int j = 0; j < QW; ++j) {
        const double2 x = er[j];
        a0[j] += l0 * (x.y - x.x * z0[j]);
        a1[j] += l1 * (x.y - x.x * z1[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < QW; ++j) {
    if (m0 < M) dz_part[((size_t)blockIdx.x * M + m0) * QP + q0 + j] = a0[j];
    if (m1 < M) dz_part[((size_t)blockIdx.x * M + m1) * QP + q0 + j] = a1[j];
  }
}

// pass R: lane = data row, the NS (2 or 4) waves of a workgroup take one NS-th of the q columns each for the same 64 rows
// (4 when there are fewer 64-row groups than SIMDs: the pass is then latency bound with one wave per SIMD, and the q split is
// its only parallelism left);
// z[m, half] and Cpsi1[m,:] are wave-uniform scalar operands.  Row-major [rows x cols] arrays are read and written
// through 64 x 16 LDS tiles (global side: 16 lanes per row = 128 contiguous bytes; LDS side: one row per lane), because a
// direct per-lane walk along a row touches 64 cache lines per load and moved 16x the bytes through L2 (586 us vs 50).
#define P1_RT 17                   // LDS tile row stride (doubles)
static __device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int QP, int DP, bool DY, int NS>
__global__ void __launch_bounds__(64 * NS) __attribute__((amdgpu_waves_per_eu(2)))
psi1_svi_rows_kernel(const int64_t N, const int M, const int Q, const int D, const double* __restrict__ Zc,
                     const double* __restrict__ lam, const double* __restrict__ Xmu, const double* __restrict__ e1,
                     const double* __restrict__ Lp, const double* __restrict__ P, const double* __restrict__ Cpsi1,
                     const double* __restrict__ Ym, const double* __restrict__ noise, double* __restrict__ gmu,
                     double* __restrict__ gS, double* __restrict__ dYm, double* __restrict__ dlam_part,
                     double* __restrict__ lsum_part) {
  constexpr int QH = QP / NS;
  constexpr int NTH = 64 * NS;
  // sweep: two 64 x 16 tiles (Lp chunk, P chunk); epilogue: per wave one 64 x (16 + 16) tile of s1 | s2 columns plus s0
  __shared__ double shbuf[NS * (64 * 2 * P1_RT) + NS * 64];
  double* const tl0 = shbuf;
  double* const tl1 = shbuf + 64 * P1_RT;
  __shared__ double dl_s[QP];
  __shared__ double scratch[16];
  const int t = threadIdx.x, lane = t & 63;
  const int half = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qo = half * QH;
  double ls_acc = 0.0;
  if (t < QP) dl_s[t] = 0.0;
  __syncthreads();
  const int64_t ngroups = (N + 63) / 64;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t n0 = g * 64, n = n0 + lane;
    const bool ok = n < N;
    double s0 = 0.0, s1[QH], s2[QH], dy[DP];
#pragma unroll
    for (int k = 0; k < QH; ++k) { s1[k] = 0.0; s2[k] = 0.0; }
#pragma unroll
    for (int d = 0; d < DP; ++d) dy[d] = 0.0;
    for (int m0 = 0; m0 < M; m0 += 16) {
      constexpr int NLD = 1024 / NTH;                 // tile elements per thread
      double lv[NLD], pv[NLD];
      bool vm[NLD];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = t + NTH * k, row = i >> 4, col = i & 15;
        const bool valid = (n0 + row < N) && (m0 + col < M);
        const int64_t src = (valid ? n0 + row : N - 1) * M + (valid ? m0 + col : 0);     // clamped: no branch around the load
        lv[k] = Lp[src];
        if (DY) pv[k] = P[src];
        vm[k] = valid;
      }
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = t + NTH * k, row = i >> 4, col = i & 15;
        tl0[row * P1_RT + col] = vm[k] ? lv[k] : 0.0;
        if (DY) tl1[row * P1_RT + col] = vm[k] ? pv[k] : 0.0;
      }
      __syncthreads();
      const int mc = (M - m0 < 16) ? (M - m0) : 16;
#pragma unroll 2
      for (int j = 0; j < mc; ++j) {
        const double* __restrict__ zr = Zc + (size_t)(m0 + j) * QP + qo;
        const double lp = tl0[lane * P1_RT + j];
        s0 += lp;
#pragma unroll
        for (int k = 0; k < QH; ++k) {
          const double lz = lp * zr[k];
          s1[k] += lz;
          s2[k] += lz * zr[k];
        }
        if (DY && half == 0) {
          const double p = tl1[lane * P1_RT + j];
#pragma unroll
          for (int d = 0; d < DP; ++d) dy[d] += p * (d < D ? Cpsi1[(size_t)(m0 + j) * D + d] : 0.0);
        }
      }
    }
    if (!ok) s0 = 0.0;
    if (half == 0) ls_acc += s0;
    // epilogue: transpose s0, s1, s2 through a wave-private LDS tile, then work element-wise in the coalesced layout
    // (16 lanes per row): mu, e are read and gmu, gS written as 128-byte row segments.
    __syncthreads();                                  // both waves are done with the Lp / P tiles: the buffer is reused
    double* st = shbuf + (size_t)half * (64 * 2 * P1_RT);
    double* s0t = shbuf + NS * (64 * 2 * P1_RT) + half * 64;
    s0t[lane] = s0;
#pragma unroll
    for (int c0 = 0; c0 < QH; c0 += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (c0 + k < QH) {
          st[lane * (2 * P1_RT) + k] = s1[c0 + k];
          st[lane * (2 * P1_RT) + P1_RT + k] = s2[c0 + k];
        }
      }
      wave_lds_sync();
      const int col = c0 + (lane & 15);
      const bool cok = col < QH;
      const int cc = cok ? col : QH - 1;
      const int q = qo + cc;
      const double il = (q < Q) ? 1.0 / lam[q] : 0.0;
      double mu[16], e[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = 4 * k + (lane >> 4);
        const int64_t rr = (n0 + row < N) ? n0 + row : N - 1;
        mu[k] = Xmu[rr * QP + q];
        e[k] = e1[rr * QP + q];
      }
      double dls = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = 4 * k + (lane >> 4);
        const bool rok = n0 + row < N;
        const double a1 = st[row * (2 * P1_RT) + (lane & 15)], a2 = st[row * (2 * P1_RT) + P1_RT + (lane & 15)], sr = s0t[row];
        const double S2 = mu[k] * mu[k] * sr - 2.0 * mu[k] * a1 + a2;
        if (rok && cok) {
          gmu[(n0 + row) * QP + q] = -e[k] * (mu[k] * sr - a1);
          gS[(n0 + row) * QP + q] = 0.5 * e[k] * e[k] * S2 - 0.5 * e[k] * sr;
        }
        const double el = e[k] * il;
        dls += (rok && cok && q < Q) ? (-0.5 * el * el * S2 - 0.5 * sr * (1.0 - el) * il) : 0.0;
      }
      dls += __shfl_xor(dls, 16);
      dls += __shfl_xor(dls, 32);
      if (lane < 16 && cok) dl_s[q] += dls;          // slot q belongs to this lane of this wave only
      wave_lds_sync();
    }
    if (DY && half == 0 && ok) {
      const double beta = 1.0 / noise[0];
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) dYm[n * D + d] = -beta * Ym[n * D + d] + dy[d];
    }
  }
  const double ls_tot = block_sum(ls_acc, scratch);
  __syncthreads();
  if (t < QP) dlam_part[(size_t)blockIdx.x * QP + t] = dl_s[t];
  if (t == 0) lsum_part[blockIdx.x] = ls_tot;
}

bool psi1_svi_supported(int D) { return D >= 1 && D <= 16; }
int psi1_svi_rows_blocks(int64_t N) {
  int64_t b = (N + 63) / 64;
  return (int)(b < 1 ? 1 : (b > PSI1_ROWS_MAXBLK ? PSI1_ROWS_MAXBLK : b));
}

template <int QP, int DP>
static void psi1_svi_dispatch(hipStream_t s, int64_t N, int M, int Q, int D, const ZTerms& zt, const RowTerms& rt,
                              const double* Ym, const double* Cpsi1, const double* noise, double* P, double* Lp,
                              double* psi1Y_part, double* gmu, double* gS, double* dYm, double* dlam_part,
                              double* lsum_part, double* dz_part, int nblk) {
  if (DP <= 4)
    hipLaunchKernelGGL((psi1_svi_fwd2_kernel<QP, (DP <= 4 ? DP : 1)>), dim3(nblk), dim3(256), 0, s, N, M, D, zt.Zc, rt.e1,
                       rt.em, rt.g1, Ym, Cpsi1, dYm ? P : nullptr, Lp, psi1Y_part);
  else
    hipLaunchKernelGGL((psi1_svi_fwd_kernel<QP, DP>), dim3(nblk), dim3(256), 0, s, N, M, D, zt.Zc, rt.e1, rt.em, rt.g1, Ym,
                       Cpsi1, dYm ? P : nullptr, Lp, psi1Y_part);
  const int rb = psi1_svi_rows_blocks(N);
  if ((N + 63) / 64 < 640) {                // far fewer 64-row groups than SIMDs (1024): four q quarters per group (+1.2 % per step at 255 groups)
    if (dYm)
      hipLaunchKernelGGL((psi1_svi_rows_kernel<QP, DP, true, 4>), dim3(rb), dim3(256), 0, s, N, M, Q, D, zt.Zc, zt.lam, rt.Xmu,
                         rt.e1, Lp, P, Cpsi1, Ym, noise, gmu, gS, dYm, dlam_part, lsum_part);
    else
      hipLaunchKernelGGL((psi1_svi_rows_kernel<QP, 1, false, 4>), dim3(rb), dim3(256), 0, s, N, M, Q, D, zt.Zc, zt.lam, rt.Xmu,
                         rt.e1, Lp, Lp, Cpsi1, Ym, noise, gmu, gS, dYm, dlam_part, lsum_part);
  } else if (dYm)
    hipLaunchKernelGGL((psi1_svi_rows_kernel<QP, DP, true, 2>), dim3(rb), dim3(128), 0, s, N, M, Q, D, zt.Zc, zt.lam, rt.Xmu,
                       rt.e1, Lp, P, Cpsi1, Ym, noise, gmu, gS, dYm, dlam_part, lsum_part);
  else
    hipLaunchKernelGGL((psi1_svi_rows_kernel<QP, 1, false, 2>), dim3(rb), dim3(128), 0, s, N, M, Q, D, zt.Zc, zt.lam, rt.Xmu,
                       rt.e1, Lp, Lp, Cpsi1, Ym, noise, gmu, gS, dYm, dlam_part, lsum_part);
  hipLaunchKernelGGL((psi1_svi_dz_kernel<QP>), dim3(nblk), dim3(256), 0, s, N, M, zt.Zc, rt.e1, rt.em, Lp, dz_part);
}

// Launches the three passes; the caller reduces psi1Y_part [nblk x D x M], dz_part [nblk x M x QP] and
// dlam_part / lsum_part [psi1_svi_rows_blocks(N) x QP / x 1].
void launch_psi1_svi(hipStream_t s, int64_t N, int M, int Q, int D, const ZTerms& zt, const RowTerms& rt, const double* Ym,
                     const double* Cpsi1, const double* noise, double* P, double* Lp, double* psi1Y_part, double* gmu,
                     double* gS, double* dYm, double* dlam_part, double* lsum_part, double* dz_part, int nblk) {
#define P1S(qp)                                                                                                       \
  case qp:                                                                                                            \
    if (D <= 1) psi1_svi_dispatch<qp, 1>(s, N, M, Q, D, zt, rt, Ym, Cpsi1, noise, P, Lp, psi1Y_part, gmu, gS, dYm, dlam_part, lsum_part, dz_part, nblk); \
    else if (D <= 4) psi1_svi_dispatch<qp, 4>(s, N, M, Q, D, zt, rt, Ym, Cpsi1, noise, P, Lp, psi1Y_part, gmu, gS, dYm, dlam_part, lsum_part, dz_part, nblk); \
    else psi1_svi_dispatch<qp, 16>(s, N, M, Q, D, zt, rt, Ym, Cpsi1, noise, P, Lp, psi1Y_part, gmu, gS, dYm, dlam_part, lsum_part, dz_part, nblk); \
    break;
  switch (zt.QP) {
    P1S(8) P1S(16) P1S(20) P1S(24) P1S(32) P1S(40) P1S(48) P1S(56) P1S(64)
    default: break;
  }
#undef P1S
}

// Wide outputs (D > 16): passes F (with T = Ym Cpsi1^T already in Lp), R (row gradients; dL/dY is a GEMM of the caller) and Z.
template <int QP>
static void psi1_svi_wide_dispatch(hipStream_t s, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt,
                                   const double* noise, double* P, double* Lp, double* gmu, double* gS, double* dlam_part,
                                   double* lsum_part, double* dz_part, int nblk) {
  hipLaunchKernelGGL((psi1_svi_fwdT_kernel<QP>), dim3(nblk), dim3(256), 0, s, N, M, zt.Zc, rt.e1, rt.em, rt.g1, P, Lp);
  const int rb = psi1_svi_rows_blocks(N);
  if ((N + 63) / 64 < 640)
    hipLaunchKernelGGL((psi1_svi_rows_kernel<QP, 1, false, 4>), dim3(rb), dim3(256), 0, s, N, M, Q, 1, zt.Zc, zt.lam, rt.Xmu,
                       rt.e1, Lp, Lp, Lp, Lp, noise, gmu, gS, nullptr, dlam_part, lsum_part);
  else
    hipLaunchKernelGGL((psi1_svi_rows_kernel<QP, 1, false, 2>), dim3(rb), dim3(128), 0, s, N, M, Q, 1, zt.Zc, zt.lam, rt.Xmu,
                       rt.e1, Lp, Lp, Lp, Lp, noise, gmu, gS, nullptr, dlam_part, lsum_part);
  hipLaunchKernelGGL((psi1_svi_dz_kernel<QP>), dim3(nblk), dim3(256), 0, s, N, M, zt.Zc, rt.e1, rt.em, Lp, dz_part);
}

void launch_psi1_svi_wide(hipStream_t s, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt, const double* noise,
                          double* P, double* Lp, double* gmu, double* gS, double* dlam_part, double* lsum_part, double* dz_part,
                          int nblk) {
  switch (zt.QP) {
#define P1W(qp) case qp: psi1_svi_wide_dispatch<qp>(s, N, M, Q, zt, rt, noise, P, Lp, gmu, gS, dlam_part, lsum_part, dz_part, nblk); break;
    P1W(8) P1W(16) P1W(20) P1W(24) P1W(32) P1W(40) P1W(48) P1W(56) P1W(64)
#undef P1W
    default: break;
  }
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
                                                 double* __restrict__ p2slot, const double* __restrict__ etab) {
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
  for (int q = 0; q < QP; q += 4) {
    c0 += dzl[q] * zs[q];
    c1 += dzl[q + 1] * zs[q + 1];
    c2 += dzl[q + 2] * zs[q + 2];
    c3 += dzl[q + 3] * zs[q + 3];
  }
  const double E = (base + hbm) - 0.5 * ((c0 + c1) + (c2 + c3));
  const double p = exp_tab(E, etab);
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
                                                  const int tr, const bool act, const double* __restrict__ etab) {
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
                              p2acc + (own ? mpa * (mpa + 1) / 2 + tr : tri + tr), etab);
    }
    if (mp + 1 < b) {
      const bool own = act && mpb >= tr;
      psi2_pair<QP, FWD, BWD>(Zc + (size_t)mpb * QP, dzl, y, v, base, hb[mpb], dlb,
                              p2acc + (own ? mpb * (mpb + 1) / 2 + tr : tri + tr), etab);
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
  __shared__ double etab[PSI2_EXP_TAB];
  if (threadIdx.x < PSI2_EXP_TAB) etab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / PSI2_EXP_TAB));

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
    psi2_sweep<QP, FWD, BWD>(team ? Mh : 0, team ? M : Mh, Zc, dlrow, M, dzl, y, v, base, hb, p2acc, tri, tr, act, etab);
    if (FWD) __syncthreads();
    psi2_sweep<QP, FWD, BWD>(team ? 0 : Mh, team ? Mh : M, Zc, dlrow, M, dzl, y, v, base, hb, p2acc, tri, tr, act, etab);

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
  __shared__ double etab[PSI2_EXP_TAB];
  if (threadIdx.x < PSI2_EXP_TAB) etab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / PSI2_EXP_TAB));

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
      psi2_sweep<QP, true, true>(a, b, Zc, dlrow, M, dzl, y, v, base, hb, p2acc, tri, tr, act, etab);
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
    constexpr int QQ = QP / 4;                           // QP is a multiple of 4
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

// ------------------------------------------------------------------------------------------------------------------
// Fused forward + backward pass with PACKED teams: the 512 threads of a workgroup are split into NT = 512 / TS teams with
// team stride TS = max(M, 64) (M = 100: 5 rows in flight, 500 of 512 lanes busy instead of 400 with 128-lane teams; M = 50:
// 8 rows; M = 128: 4), so a wave holds lanes of up to two data rows.  What keeps that cheap:
//   * all threads sweep m' in the same order, so z[m',:] stays a scalar operand for mixed waves too;
//   * the forward sums are LDS fp64 atomics (ds_add_f64) from all teams, no rotation and no barriers inside the sweep.
//     The NT adds a slot receives per iteration arrive in hardware order, so the last bits of psi2 can differ from run to
//     run; rgp_set_deterministic (or RGP_PSI2_NO_PACKED=1) selects the rotating four-team kernel instead, which is
//     bit-reproducible.  (A 64-bit fixed-point accumulator was order independent but kept only ~9 digits on entries far
//     below the row bound; it failed the 1e-10 gradient parity on strongly cancelling cases.)
//   * the dZ slab is updated by one team after the other (NT short phases);
//   * the per-row reductions over m split each wave's transposed sum at the team boundary.
// ------------------------------------------------------------------------------------------------------------------
#define PSI2_MAX_NT 8
#ifndef PSI2_UNR
#define PSI2_UNR 4            // pairs per trip of the sweep loop
#endif

template <int QP>
static __device__ __forceinline__ void psi2_pair_fix(const double* __restrict__ zs, const double (&dzl)[QP], double (&y)[QP],
                                                     double& v, const double base, const double hbm, const double dlv,
                                                     double* __restrict__ p2slot, const bool own, const double* __restrict__ etab) {
  // dzl carries the factor -1/2, and the first chain starts from the (row, m) + (row, m') constant: E is one add away.
  // Two chains are enough: a dependent v_fma_f64 issues every ~6 cycles from one wave (profiles/r1_f64_lat.txt).
  double c0 = base + hbm, c1 = 0.0;
#pragma unroll
  for (int q = 0; q < QP; q += 2) {
    c0 += dzl[q] * zs[q];
    c1 += dzl[q + 1] * zs[q + 1];
  }
  const double p = exp_tab(c0 + c1, etab);
  if (own) atomicAdd(p2slot, p);          // ds_add_f64, no return
  const double w = p * dlv;
  v += w;
#pragma unroll
  for (int q = 0; q < QP; ++q) y[q] += w * zs[q];
}

#ifdef PSI2_STAMPS
#define STAMP(acc) do { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define STAMP(acc) do { } while (0)
#endif

template <int QP>
__global__ void __launch_bounds__(PSI2_T4, 2)
psi2_rowsN_kernel(const int64_t N, const int M, const int Q, const int NT, const int TS, const double* __restrict__ Zc,
                  const double* __restrict__ lam_g, const double* __restrict__ dLs_g, const double* __restrict__ Xmu_g,
                  const double* __restrict__ d2_g, const double* __restrict__ g0_g,
                  double* __restrict__ p2_part, double* __restrict__ dz_part, double* __restrict__ dlam_part,
                  double* __restrict__ wsum_part, double* __restrict__ gmu_g, double* __restrict__ gS_g,
                  const int accumulate, double* dbgout) {
  extern __shared__ double lds[];
  const int tri = M * (M + 1) / 2;
  constexpr int RW = 2 * QP + 8;
  constexpr int RD = 2 * QP + 2;                          // staged row: d2[QP] | mu[QP] | g0 | pad
  double* p2acc = lds;                                    // [tri]
  double* dzacc = lds + tri;                              // [QP x 128]
  double* hbuf = dzacc + QP * RGP_PSI2_TEAM;              // [NT x 128]
  double* rowd = hbuf + NT * RGP_PSI2_TEAM;               // [NT x RD]
  double* red = rowd + NT * RD;                           // [8 waves x RED_WAVE]
  double* res = red + 8 * RED_WAVE;                       // [8 waves x 2 x RW]
  double* etab = res + 8 * 2 * RW;                        // [256] 2^(j/256)
#ifdef PSI2_STAMPS
  long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tF = 0, tG = 0, tH = 0, t0 = 0, t1;
#endif

  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  // team stride TS = max(M, 64): teams of exactly M threads for M >= 64, one (partly idle) wave per team below that
  const int team_raw = t / TS;
  const int tr_raw = t - team_raw * TS;
  const bool act = team_raw < NT && tr_raw < M;           // idle: the last 512 - NT*TS threads and the lanes >= M of a team
  const int team = team_raw < NT ? team_raw : NT - 1;
  const int tr = act ? tr_raw : M - 1;
  const double* __restrict__ zrowp = Zc + (size_t)tr * QP;
#define zrow(q) zrowp[q]
  double* hb = hbuf + team * RGP_PSI2_TEAM;
  const double* rd = rowd + team * RD;
  // team boundary inside this wave: lanes < bnd belong to team ta, the others to ta + 1
  const int ta = (64 * wave) / TS;
  const int bnd = __builtin_amdgcn_readfirstlane(((ta + 1) * TS - 64 * wave < 64) ? (ta + 1) * TS - 64 * wave : 64);
  const bool inA = lane < bnd;

  for (int i = t; i < tri; i += PSI2_T4) p2acc[i] = 0.0;
  for (int i = t; i < QP * RGP_PSI2_TEAM; i += PSI2_T4) dzacc[i] = 0.0;
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  double ellq = 0.0;
#pragma unroll
  for (int q = 0; q < QP; ++q) ellq += 0.25 * lam_g[q] * zrow(q) * zrow(q);
  double dlam_acc = 0.0, wsum_acc = 0.0;
  long long iters = 0;

  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + NT - 1) / NT * NT;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  const double* __restrict__ dlrow = dLs_g + tr;
  const int tr_own = act ? tr : 0x7fffffff;          // inactive threads own no slot

  // staged row elements of this thread (NT*RD <= 1024: at most two per thread), loaded one iteration ahead
  const bool stager = t < NT * RD, stager2 = t + PSI2_T4 < NT * RD;
  auto stage_load = [&](int64_t rr, int e) -> double {
    const int sk = e / RD, sj = e - sk * RD;
    const int64_t nk = (rr + sk < r1) ? rr + sk : r1 - 1;
    if (nk < 0) return 0.0;
    if (sj < QP) return d2_g[nk * QP + sj];
    if (sj < 2 * QP) return Xmu_g[nk * QP + (sj - QP)];
    if (sj == 2 * QP) return g0_g[nk];
    return 0.0;
  };
  double staged = (stager && r0 < r1) ? stage_load(r0, t) : 0.0;
  double staged2 = (stager2 && r0 < r1) ? stage_load(r0, t + PSI2_T4) : 0.0;
  if (stager) rowd[t] = staged;
  if (stager2) rowd[t + PSI2_T4] = staged2;
  __syncthreads();

  for (int64_t rr = r0; rr < r1; rr += NT) {
    asm volatile("" ::: "memory");
    ++iters;
#ifdef PSI2_STAMPS
    t0 = __builtin_readcyclecounter();
#endif
    const bool valid = rr + team < r1;
    const int64_t n = valid ? rr + team : r1 - 1;
    // loads whose latency hides under the sweep: next iteration's rows, and this row's gradient slots
    staged = (stager && rr + NT < r1) ? stage_load(rr + NT, t) : 0.0;
    staged2 = (stager2 && rr + NT < r1) ? stage_load(rr + NT, t + PSI2_T4) : 0.0;
    const bool grad_thread = team_raw < NT && tr_raw < QP && valid;      // lanes >= M of a short team do this job too (TS >= 64 >= QP)
    double g_old_mu = 0.0, g_old_S = 0.0;
    if (grad_thread && accumulate && tr_raw < Q) {
      g_old_mu = gmu_g[n * QP + tr_raw];
      g_old_S = gS_g[n * QP + tr_raw];
    }
    STAMP(tA);
    double dzl[QP];
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int q = 0; q < QP; q += 2) {
      const double z0 = zrow(q), z1 = zrow(q + 1);
      const double d0 = rd[q], d1 = rd[q + 1];
      dzl[q] = -0.5 * z0 * (d0 - lam_g[q]);
      dzl[q + 1] = -0.5 * z1 * (d1 - lam_g[q + 1]);
      h0 += z0 * d0 * (rd[QP + q] - 0.25 * z0);
      h1 += z1 * d1 * (rd[QP + q + 1] - 0.25 * z1);
    }
    const double h = (h0 + h1) - ellq;
    if (act) hb[tr] = h;
    const double base = (act && valid) ? (rd[2 * QP] + h) : -INFINITY;
    __syncthreads();
    STAMP(tB);

    double y[QP];
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < QP; ++q) y[q] = 0.0;
    {
      const int last = M - 1;
      double dlv[PSI2_UNR];
#pragma unroll
      for (int u = 0; u < PSI2_UNR; ++u) dlv[u] = dlrow[(size_t)((u < M) ? u : last) * M];
      for (int mp = 0; mp < M; mp += PSI2_UNR) {
        double dlc[PSI2_UNR];
#pragma unroll
        for (int u = 0; u < PSI2_UNR; ++u) {
          dlc[u] = dlv[u];
          dlv[u] = dlrow[(size_t)((mp + PSI2_UNR + u < M) ? mp + PSI2_UNR + u : last) * M];
        }
#pragma unroll
        for (int u = 0; u < PSI2_UNR; ++u) {
          const int mq = mp + u;
          if (u == 0 || mq < M)
            psi2_pair_fix<QP>(Zc + (size_t)mq * QP, dzl, y, v, base, hb[mq], dlc[u], p2acc + (mq * (mq + 1) / 2 + tr),
                              mq >= tr_own, etab);
        }
      }
    }

    asm volatile("" ::: "memory");
    STAMP(tC);
    // per-wave reductions over m, split at the team boundary: slot 0 = lanes < bnd (team ta), slot 1 = the rest.
    // LDS executes one wave's instructions in order, so the transposing write -> read needs no wait in between; the
    // 8-lane partial sums are finished with DPP moves (no ds_bpermute round trips).
    double* wres = res + (size_t)wave * 2 * RW;
    {
      double* wred = red + wave * RED_WAVE;
      const double wsA = wave_sum_dpp(inA ? v : 0.0);
      const double wsB = (bnd < 64) ? wave_sum_dpp(inA ? 0.0 : v) : 0.0;
      if (lane == 0) { wres[2 * QP] = wsA; wres[RW + 2 * QP] = wsB; }
      const int jv = lane >> 3, part = lane & 7;
      // this thread's z values of column group k + 1 are requested before group k is worked on: a load issued at its use
      // costs one global round trip per group, with nothing else to run in this part of the kernel
      double zn[RED_QC];
#pragma unroll
      for (int j = 0; j < RED_QC; ++j) zn[j] = zrow(j);
#pragma unroll
      for (int q0 = 0; q0 < QP; q0 += RED_QC) {
        double zc[RED_QC];
#pragma unroll
        for (int j = 0; j < RED_QC; ++j) zc[j] = zn[j];
        if (q0 + RED_QC < QP) {
#pragma unroll
          for (int j = 0; j < RED_QC; ++j) zn[j] = zrow(q0 + RED_QC + j);
        }
#pragma unroll
        for (int j = 0; j < RED_QC; ++j) {
          const double z = zc[j];
          wred[j * RED_RS + lane] = v * z;
          wred[(RED_QC + j) * RED_RS + lane] = 0.5 * z * (v * z + y[q0 + j]);
        }
        __builtin_amdgcn_wave_barrier();
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = wred[jv * RED_RS + part + 8 * k];
        __builtin_amdgcn_wave_barrier();
        double sA = 0.0, sB = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const bool a = part + 8 * k < bnd;
          sA += a ? x[k] : 0.0;
          sB += a ? 0.0 : x[k];
        }
        sA = sum8_dpp(sA);
        if (bnd < 64) sB = sum8_dpp(sB);
        if (part == 0) {
          const int o = (jv < RED_QC ? 0 : QP) + q0 + (jv % RED_QC);
          wres[o] = sA;
          wres[RW + o] = sB;
        }
      }
    }
#ifdef PSI2_STAMPS
    STAMP(tD);
    __syncthreads();
    STAMP(tG);
#endif
    // dZ accumulation  d (v (2 mu - z) - y)  into the LDS slab
    {
      constexpr int ZB = 4;                              // same one-group-ahead requests as above
      double zn[ZB];
#pragma unroll
      for (int j = 0; j < ZB; ++j) zn[j] = zrow(j);
#pragma unroll
      for (int q0 = 0; q0 < QP; q0 += ZB) {
        double zc[ZB];
#pragma unroll
        for (int j = 0; j < ZB; ++j) zc[j] = zn[j];
        if (q0 + ZB < QP) {
#pragma unroll
          for (int j = 0; j < ZB; ++j) zn[j] = zrow(q0 + ZB + j);
        }
#pragma unroll
        for (int j = 0; j < ZB; ++j) y[q0 + j] = rd[q0 + j] * (v * (2.0 * rd[QP + q0 + j] - zc[j]) - y[q0 + j]);
      }
    }
    STAMP(tH);
#ifdef PSI2_DZ_PHASES
    for (int k = 0; k < NT; ++k) {
      if (act && team_raw == k) {
        double tmp[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) tmp[q] = dzacc[q * RGP_PSI2_TEAM + tr];
#pragma unroll
        for (int q = 0; q < QP; ++q) dzacc[q * RGP_PSI2_TEAM + tr] = tmp[q] + y[q];
      }
      __syncthreads();
    }
#else
    // all teams at once with LDS fp64 atomics (hardware order, like the forward sums; the deterministic mode uses the
    // rotating-team kernel): one barrier instead of NT, which also publishes the wave partials to the row-gradient threads
    if (act) {
#pragma unroll
      for (int q = 0; q < QP; ++q) atomicAdd(&dzacc[q * RGP_PSI2_TEAM + tr], y[q]);
    }
    __syncthreads();
#endif
    STAMP(tE);
    // row gradients: thread (team, q) combines the wave partials that cover its team, in wave order
    if (grad_thread) {
      const int q = tr_raw;
      if (q < Q) {
        const int w0 = (team * TS) >> 6, w1 = (team * TS + M - 1) >> 6;
        double w = 0.0, a1 = 0.0, a23 = 0.0;
        for (int wv = w0; wv <= w1; ++wv) {
          const int slot = ((64 * wv) / TS == team) ? 0 : 1;
          const double* tres = res + ((size_t)wv * 2 + slot) * RW;
          w += tres[2 * QP];
          a1 += tres[q];
          a23 += tres[QP + q];
        }
        const double mu = rd[QP + q], dq = rd[q];
        const double dmu = 2.0 * dq * (a1 - mu * w);
        const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
        const double dS = -2.0 * dq * dq * dd;
        gmu_g[n * QP + q] = g_old_mu + dmu;
        gS_g[n * QP + q] = g_old_S + dS;
        const double il = 1.0 / lam_g[q];
        dlam_acc += dd * (dq * il) * (dq * il) - 0.5 * w * il;
        if (q == 0) wsum_acc += w;
      } else if (!accumulate) {
        gmu_g[n * QP + q] = 0.0;
        gS_g[n * QP + q] = 0.0;
      }
    }
    __syncthreads();                                  // everyone is done with rowd / res / hbuf of this iteration
    if (stager) rowd[t] = staged;
    if (stager2) rowd[t + PSI2_T4] = staged2;
    __syncthreads();
    STAMP(tF);
  }
#ifdef PSI2_STAMPS
  if (dbgout && t == 0) { double* o = dbgout + blockIdx.x * 8; o[0] = tA; o[1] = tB; o[2] = tC; o[3] = tD; o[4] = tE; o[5] = tF; o[6] = (double)iters; o[7] = tG; dbgout[256 * 8 + blockIdx.x] = tH; }
#endif

  {
    double* out = p2_part + (size_t)blockIdx.x * tri;
    for (int i = t; i < tri; i += PSI2_T4) out[i] = p2acc[i];
    double* outz = dz_part + (size_t)blockIdx.x * QP * RGP_PSI2_TEAM;
    for (int i = t; i < QP * RGP_PSI2_TEAM; i += PSI2_T4) outz[i] = dzacc[i];
  }
  if (team_raw < NT && tr_raw < QP) dlam_part[((size_t)blockIdx.x * NT + team) * QP + tr_raw] = (tr_raw < Q) ? dlam_acc : 0.0;
  if (team_raw < NT && tr_raw == 0) wsum_part[(size_t)blockIdx.x * NT + team] = wsum_acc;
}
#undef zrow

// ------------------------------------------------------------------------------------------------------------------
// The packed-team pass with its waves in two roles, half a step apart.
//   The sweep of psi2_rowsN_kernel issues at the rate of the fp64 pipe; the rest of an iteration (row reductions, dZ update,
//   next row's operands) is a third of its time and is bound by latencies -- LDS round trips, uncoalesced loads of the thread's
//   z row, four workgroup barriers that hold all eight waves in the same phase.  Here each SIMD's two waves take different
//   roles: role 0 runs [sweep(k), pass(k), barrier], role 1 runs [sweep(k), barrier, pass(k)], where pass(k) is everything
//   between two sweeps.  One barrier per iteration, and it releases role 0 into sweep(k+1) while role 1 does pass(k): from then
//   on a SIMD always has one wave in the sweep while the other is in its pass (a single wave issues an fp64 multiply-add every
//   ~6 cycles, two share the pipe at ~5).  What a wave publishes for the others goes through rings deep enough for the
//   half-step offset (b_k = the barrier of iteration k):
//     h values of a row's sweep   hb(k+1): role 0 in pass(k), before b_k; role 1 computes hb(k+2) in pass(k), after b_k;
//                                 read in sweep(k+1), after b_k; two slots
//     row operands                rowd(k+2) written by role 0 in pass(k), rowd(k+3) by role 1; last read by the row-gradient
//                                 threads of iteration k two barriers later; five slots
//     per-wave partial row sums   wres(k) written in pass(k), combined into the row gradients of iteration k by role 0 in
//                                 pass(k+2) and by role 1 in pass(k+1) -- both between b_{k+1} and b_{k+2}; three slots
//   pass(k) reads the thread's z row ONCE: reductions and dZ update of iteration k, then dzl and h of the coming rows, column
//   group by column group (the y registers of a column are dead when its dzl is written).
//   Roles are handed out by SIMD (hardware id register + an LDS ticket), so the two waves of a SIMD never share one.
// ------------------------------------------------------------------------------------------------------------------
#ifndef PSI2_PRIO_PASS
#define PSI2_PRIO_PASS 3        // issue priority (s_setprio) of a wave between two sweeps, -1: leave priorities alone
#endif
#ifndef PSI2_PRIO_SWEEP0
#define PSI2_PRIO_SWEEP0 1      // ... of a role-0 wave in the sweep (pass/sweep0/sweep1 = 3/1/0: 56.4, 3/0/0: 56.0, 3/0/1 and 3/0/2: 54.5 steps/s)
#endif
#ifndef PSI2_PRIO_SWEEP1
#define PSI2_PRIO_SWEEP1 0      // ... of a role-1 wave in the sweep
#endif
#ifndef PSI2_PRIO_ALT
#define PSI2_PRIO_ALT 8         // columns per turn of the alternating sweep priority (0: the fixed priorities below); the hardware
                                // otherwise serves the older wave of a SIMD first and the younger one only in its gaps:
                                // 58.0 steps/s with fixed priorities, 59.8 with turns of 4 or 8 columns, 59.5 with 16
#endif
#ifndef PSI2_PRIO_SW0
#define PSI2_PRIO_SW0 4         // eighths of the sweep after which a role-0 wave changes to PSI2_PRIO_SWEEP0B (8: never)
#endif
#ifndef PSI2_PRIO_SWEEP0B
#define PSI2_PRIO_SWEEP0B 0
#endif
#ifndef PSI2_PRIO_SW1
#define PSI2_PRIO_SW1 4         // the same for a role-1 wave (4/4: 58.0 steps/s against 57.0 for never/never on the same box)
#endif
#ifndef PSI2_PRIO_SWEEP1B
#define PSI2_PRIO_SWEEP1B 1
#endif
#define PSI2_RING_ROWD 5
#define PSI2_RING_RES 3

typedef __attribute__((address_space(1))) double gdouble;

// Wave-uniform values that only the code between two sweeps needs are parked in the lanes of one VGPR and read back with
// v_readlane after every sweep: the sweep keeps z[m',:] of the pair in work in 80 of the ~100 SGPRs, and every scalar that is
// live across it is spilled and reloaded inside the pair loop by the register allocator (72 extra instructions per pair).
enum { PK_M, PK_Q, PK_NT, PK_TS, PK_ACC, PK_BND, PK_R0, PK_R0H, PK_R1, PK_R1H, PK_XMU, PK_XMUH, PK_D2, PK_D2H, PK_G0, PK_G0H,
       PK_GMU, PK_GMUH, PK_GS, PK_GSH, PK_P2, PK_P2H, PK_DZ, PK_DZH, PK_DLAM, PK_DLAMH, PK_WSUM, PK_WSUMH, PK_K, PK_ROLE };
static __device__ __forceinline__ int pk_put(int pk, const int idx, int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm("v_writelane_b32 %0, %1, %2" : "+v"(pk) : "s"(v), "n"(idx));
  return pk;
}
static __device__ __forceinline__ int pk_putp(int pk, int idx, const void* p) {
  const unsigned long long u = (unsigned long long)p;
  return pk_put(pk_put(pk, idx, (int)(unsigned)u), idx + 1, (int)(unsigned)(u >> 32));
}
static __device__ __forceinline__ int pk_get(int pk, int idx) { return __builtin_amdgcn_readlane(pk, idx); }
static __device__ __forceinline__ long long pk_get64(int pk, int idx) {
  return (long long)(((unsigned long long)(unsigned)pk_get(pk, idx + 1) << 32) | (unsigned)pk_get(pk, idx));
}
static __device__ __forceinline__ gdouble* pk_getp(int pk, int idx) { return (gdouble*)pk_get64(pk, idx); }

template <int QP, int QC>
__global__ void __launch_bounds__(PSI2_T4, 2)
psi2_rowsP_kernel(const int64_t N, const int64_t per, const int M, const int Q, const int NT, const int TS,
                  const double* __restrict__ Zc, const double* __restrict__ lam_g, const double* __restrict__ dLs_g,
                  const double* __restrict__ Xmu_g, const double* __restrict__ d2_g, const double* __restrict__ g0_g,
                  double* __restrict__ p2_part, double* __restrict__ dz_part, double* __restrict__ dlam_part,
                  double* __restrict__ wsum_part, double* __restrict__ gmu_g, double* __restrict__ gS_g,
                  const int accumulate, double* dbgout) {
  extern __shared__ double lds[];
  constexpr int RW = 2 * QP + 2;                          // a1[QP] | a23[QP] | w | pad
  constexpr int RD = 2 * QP + 2;                          // staged row: d2[QP] | mu[QP] | g0 | pad
  constexpr int NV = 2 * QC;                              // value rows of one transposition
  constexpr int G = 64 / NV;                              // lanes that share one value row
  constexpr int RS = (QC == 4) ? 72 : 80;                 // row stride of the transposition scratch: the value rows a 32-lane
                                                          // group reads fall into disjoint banks
  __shared__ int simd_cnt[4];
  // LDS map (doubles): p2acc [tri] | dzacc [QP x DZS] | hbuf [2][NT x TS] | rowd [5][NT x RD] | red [8 waves][NV x RS] |
  //                    res [3][8 waves x 2 x RW] | etab [256] | lamL [QP]
#define P2_TRI(M_) ((M_) * ((M_) + 1) / 2)
#define P2_DZS(M_) (((M_) + 3) & ~3)
#define P2_DZACC(M_) (lds + P2_TRI(M_))
#define P2_HBUF(M_) (P2_DZACC(M_) + QP * P2_DZS(M_))
#define P2_ROWD(M_, NT_, TS_) (P2_HBUF(M_) + 2 * (NT_) * (TS_))
#define P2_RED(M_, NT_, TS_) (P2_ROWD(M_, NT_, TS_) + PSI2_RING_ROWD * (NT_) * RD)
#define P2_RES(M_, NT_, TS_) (P2_RED(M_, NT_, TS_) + 8 * NV * RS)
#define P2_ETAB(M_, NT_, TS_) (P2_RES(M_, NT_, TS_) + PSI2_RING_RES * 8 * 2 * RW)
#define P2_LAM(M_, NT_, TS_) (P2_ETAB(M_, NT_, TS_) + PSI2_EXP_TAB)

  const int t = threadIdx.x;
  const int team_raw = t / TS;
  const int tr_raw = t - team_raw * TS;
  int pk = 0;
  double ellq = 0.0;
  int K;
  int role;
  double dzl[QP];
  double base;
  const double* __restrict__ zrowp;
#define zrow(q) zrowp[q]
  {
    const int wave = t >> 6, lane = t & 63;
    const bool act = team_raw < NT && tr_raw < M;
    const int team = team_raw < NT ? team_raw : NT - 1;
    const int tr = act ? tr_raw : M - 1;
    zrowp = Zc + (size_t)tr * QP;
    const int tri = P2_TRI(M);
    double* p2acc = lds;
    double* dzacc = P2_DZACC(M);
    double* hbuf = P2_HBUF(M);
    double* rowd = P2_ROWD(M, NT, TS);
    double* etab = P2_ETAB(M, NT, TS);
    double* lamL = P2_LAM(M, NT, TS);
    for (int i = t; i < tri; i += PSI2_T4) p2acc[i] = 0.0;
    for (int i = t; i < QP * P2_DZS(M); i += PSI2_T4) dzacc[i] = 0.0;
    if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
    if (t < QP) lamL[t] = lam_g[t];
    if (t < 4) simd_cnt[t] = 0;
#pragma unroll
    for (int q = 0; q < QP; ++q) ellq += 0.25 * lam_g[q] * zrow(q) * zrow(q);

    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < N) ? r0 + per : N;
    K = (r1 > r0) ? (int)((r1 - r0 + NT - 1) / NT) : 0;     // iterations of this workgroup, the same for all its waves
    const int ta = (64 * wave) / TS;
    const int bnd = ((ta + 1) * TS - 64 * wave < 64) ? (ta + 1) * TS - 64 * wave : 64;

    // operands of the first three iterations
    for (int k = 0; k < 3; ++k) {
      for (int e = t; e < NT * RD; e += PSI2_T4) {
        const int sk = e / RD, sj = e - sk * RD;
        int64_t nk = r0 + (int64_t)k * NT + sk;
        nk = (nk < r1) ? nk : r1 - 1;
        nk = (nk < 0) ? 0 : nk;
        rowd[k * NT * RD + e] = (sj < QP) ? d2_g[nk * QP + sj] : (sj < 2 * QP) ? Xmu_g[nk * QP + (sj - QP)] : (sj == 2 * QP) ? g0_g[nk] : 0.0;
      }
    }
    __syncthreads();
    // role of this wave: the two waves of a SIMD draw tickets 0 and 1
    {
      const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3u;      // HW_ID bits [5:4]
      int tk = 0;
      if (lane == 0) tk = atomicAdd(&simd_cnt[simd], 1);
      role = __builtin_amdgcn_readfirstlane(tk) & 1;
      if (dbgout && lane == 0) dbgout[blockIdx.x * 8 + wave] = (double)(simd * 10 + role);
    }
    pk = pk_put(pk, PK_M, M); pk = pk_put(pk, PK_Q, Q); pk = pk_put(pk, PK_NT, NT); pk = pk_put(pk, PK_TS, TS);
    pk = pk_put(pk, PK_ACC, accumulate); pk = pk_put(pk, PK_BND, bnd);
    pk = pk_putp(pk, PK_R0, (const void*)r0); pk = pk_putp(pk, PK_R1, (const void*)r1);
    pk = pk_putp(pk, PK_XMU, Xmu_g); pk = pk_putp(pk, PK_D2, d2_g); pk = pk_putp(pk, PK_G0, g0_g);
    pk = pk_putp(pk, PK_GMU, gmu_g); pk = pk_putp(pk, PK_GS, gS_g);
    pk = pk_putp(pk, PK_P2, p2_part); pk = pk_putp(pk, PK_DZ, dz_part); pk = pk_putp(pk, PK_DLAM, dlam_part);
    pk = pk_putp(pk, PK_WSUM, wsum_part);
    pk = pk_put(pk, PK_K, K); pk = pk_put(pk, PK_ROLE, role);

    const double* rd = rowd + team * RD;
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int q = 0; q < QP; q += 2) {
      const double z0 = zrow(q), z1 = zrow(q + 1);
      const double d0 = rd[q], d1 = rd[q + 1];
      dzl[q] = -0.5 * z0 * (d0 - lam_g[q]);
      dzl[q + 1] = -0.5 * z1 * (d1 - lam_g[q + 1]);
      h0 += z0 * d0 * (rd[QP + q] - 0.25 * z0);
      h1 += z1 * d1 * (rd[QP + q + 1] - 0.25 * z1);
    }
    const double h = (h0 + h1) - ellq;
    if (act) hbuf[team * TS + tr] = h;
    base = (act && r0 + team < r1) ? (rd[2 * QP] + h) : -INFINITY;
    if (role) {
      const double* rh = rowd + NT * RD + team * RD;
      double g0 = 0.0, g1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; q += 2) {
        const double z0 = zrow(q), z1 = zrow(q + 1);
        g0 += z0 * rh[q] * (rh[QP + q] - 0.25 * z0);
        g1 += z1 * rh[q + 1] * (rh[QP + q + 1] - 0.25 * z1);
      }
      if (act) hbuf[NT * TS + team * TS + tr] = (g0 + g1) - ellq;
    }
    __syncthreads();
  }

  // thread state that lives through the sweeps (VGPRs): t, team_raw, tr_raw, zrowp, ellq, pk, the two sums below
  double dlam_acc = 0.0, wsum_acc = 0.0;

  // row gradients of iteration j: thread (team, q) combines the wave partials that cover its team, in wave order
  auto rowgrad = [&](const int pkv, const int j, const double g_old_mu, const double g_old_S) {
    const int M_ = pk_get(pkv, PK_M), Q_ = pk_get(pkv, PK_Q), NT_ = pk_get(pkv, PK_NT), TS_ = pk_get(pkv, PK_TS);
    const int64_t r0 = pk_get64(pkv, PK_R0), r1 = pk_get64(pkv, PK_R1);
    const int team = team_raw < NT_ ? team_raw : NT_ - 1;
    const int64_t n = r0 + (int64_t)j * NT_ + team;
    if (!(team_raw < NT_ && tr_raw < QP) || n >= r1) return;
    gdouble* gmu = pk_getp(pkv, PK_GMU);
    gdouble* gS = pk_getp(pkv, PK_GS);
    const int q = tr_raw;
    if (q < Q_) {
      const double* rj = P2_ROWD(M_, NT_, TS_) + (j % PSI2_RING_ROWD) * NT_ * RD + team * RD;
      const double* resj = P2_RES(M_, NT_, TS_) + (size_t)(j % PSI2_RING_RES) * 8 * 2 * RW;
      const int w0 = (team * TS_) >> 6, w1 = (team * TS_ + M_ - 1) >> 6;
      double w = 0.0, a1 = 0.0, a23 = 0.0;
      for (int wv = w0; wv <= w1; ++wv) {
        const int slot = ((64 * wv) / TS_ == team) ? 0 : 1;
        const double* tres = resj + ((size_t)wv * 2 + slot) * RW;
        w += tres[2 * QP];
        a1 += tres[q];
        a23 += tres[QP + q];
      }
      const double mu = rj[QP + q], dq = rj[q];
      const double dmu = 2.0 * dq * (a1 - mu * w);
      const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
      const double dS = -2.0 * dq * dq * dd;
      gmu[n * QP + q] = g_old_mu + dmu;
      gS[n * QP + q] = g_old_S + dS;
      const double il = 1.0 / P2_LAM(M_, NT_, TS_)[q];
      dlam_acc += dd * (dq * il) * (dq * il) - 0.5 * w * il;
      if (q == 0) wsum_acc += w;
    } else if (!pk_get(pkv, PK_ACC)) {
      gmu[n * QP + q] = 0.0;
      gS[n * QP + q] = 0.0;
    }
  };
  auto g_old_load = [&](const int pkv, const int j, double& gm, double& gs) {
    gm = 0.0; gs = 0.0;
    const int NT_ = pk_get(pkv, PK_NT);
    if (j < 0 || !pk_get(pkv, PK_ACC) || !(team_raw < NT_ && tr_raw < QP) || tr_raw >= pk_get(pkv, PK_Q)) return;
    const int team = team_raw < NT_ ? team_raw : NT_ - 1;
    const int64_t n = pk_get64(pkv, PK_R0) + (int64_t)j * NT_ + team;
    if (n >= pk_get64(pkv, PK_R1)) return;
    gm = pk_getp(pkv, PK_GMU)[n * QP + tr_raw];
    gs = pk_getp(pkv, PK_GS)[n * QP + tr_raw];
  };

#ifdef PSI2_STAMPS
  long long tS = 0, tP = 0, tW = 0, tH = 0, t0 = 0, t1;
#endif
  for (int k = 0; k < K; ++k) {
    asm volatile("" ::: "memory");
#ifdef PSI2_STAMPS
    t0 = __builtin_readcyclecounter();
#endif
    double y[QP];
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < QP; ++q) y[q] = 0.0;
    {
      const bool act = team_raw < NT && tr_raw < M;
      const int team = team_raw < NT ? team_raw : NT - 1;
      const int tr = act ? tr_raw : M - 1;
      const int tr_own = act ? tr : 0x7fffffff;
      const double* __restrict__ dlrow = dLs_g + tr;
      double* p2acc = lds;
      const double* etab = P2_ETAB(M, NT, TS);
      const double* hb = P2_HBUF(M) + (k & 1) * NT * TS + team * TS;
      const int last = M - 1;
      double dlv[PSI2_UNR];
#pragma unroll
      for (int u = 0; u < PSI2_UNR; ++u) dlv[u] = dlrow[(size_t)((u < M) ? u : last) * M];
      const int sw0 = (M * PSI2_PRIO_SW0 / 8) & ~(PSI2_UNR - 1), sw1 = (M * PSI2_PRIO_SW1 / 8) & ~(PSI2_UNR - 1);
      for (int mp = 0; mp < M; mp += PSI2_UNR) {
#ifdef PSI2_STAMPS
        if (mp == ((M / 2) & ~(PSI2_UNR - 1))) { t1 = __builtin_readcyclecounter(); tH += t1 - t0; }
#endif
#if PSI2_PRIO_PASS >= 0
#if PSI2_PRIO_ALT > 0
        // while both waves of a SIMD sweep they take the higher issue priority in turns of PSI2_PRIO_ALT columns
        if ((mp & (PSI2_PRIO_ALT - 1)) == 0) {
          if ((((mp / PSI2_PRIO_ALT) ^ role) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
#else
        if (!role && mp == sw0 && mp > 0) __builtin_amdgcn_s_setprio(PSI2_PRIO_SWEEP0B);
        if (role && mp == sw1 && mp > 0) __builtin_amdgcn_s_setprio(PSI2_PRIO_SWEEP1B);
#endif
#endif
        double dlc[PSI2_UNR];
#pragma unroll
        for (int u = 0; u < PSI2_UNR; ++u) {
          dlc[u] = dlv[u];
          dlv[u] = dlrow[(size_t)((mp + PSI2_UNR + u < M) ? mp + PSI2_UNR + u : last) * M];
        }
#pragma unroll
        for (int u = 0; u < PSI2_UNR; ++u) {
          const int mq = mp + u;
          if (u == 0 || mq < M)
            psi2_pair_fix<QP>(Zc + (size_t)mq * QP, dzl, y, v, base, hb[mq], dlc[u], p2acc + (mq * (mq + 1) / 2 + tr),
                              mq >= tr_own, etab);
        }
      }
    }
    asm volatile("" ::: "memory");
    STAMP(tS);
    if (role) __syncthreads();                          // b_k for role 1: the others have published what its pass reads
    STAMP(tW);

    // pass(k): every wave-uniform value is read back from the parked register
    {
#if PSI2_PRIO_PASS >= 0
      __builtin_amdgcn_s_setprio(PSI2_PRIO_PASS);
#endif
      int pkv = pk;
      asm volatile("" : "+v"(pkv));
      int tv = t, teamv = team_raw, trv = tr_raw;
      asm volatile("" : "+v"(tv), "+v"(teamv), "+v"(trv));
      const int M_ = pk_get(pkv, PK_M), NT_ = pk_get(pkv, PK_NT), TS_ = pk_get(pkv, PK_TS), bnd = pk_get(pkv, PK_BND);
      const int role_ = pk_get(pkv, PK_ROLE), K_ = pk_get(pkv, PK_K);
      const int64_t r0 = pk_get64(pkv, PK_R0), r1 = pk_get64(pkv, PK_R1);
      const int hoff = 1 + role_;                         // role 1 computes h one iteration further ahead
      const int ks = k + 2 + role_;                       // and stages row operands one iteration further ahead
      const int jr = k - 2 + role_;
      const int wave = tv >> 6, lane = tv & 63;
      const bool act = teamv < NT_ && trv < M_;
      const int team = teamv < NT_ ? teamv : NT_ - 1;
      const int tr = act ? trv : M_ - 1;
      const bool inA = lane < bnd;
      const int DZS = P2_DZS(M_);
      double* dzacc = P2_DZACC(M_);
      double* hbuf = P2_HBUF(M_);
      double* rowd = P2_ROWD(M_, NT_, TS_);
      const double* lamL = P2_LAM(M_, NT_, TS_);

      // loads whose latency hides under this pass: row operands of a later iteration, gradient slots of an earlier one
      const bool do_stage = ks >= 3 && ks < K_;
      const bool stager = tv < NT_ * RD, stager2 = tv + PSI2_T4 < NT_ * RD;
      double staged = 0.0, staged2 = 0.0;
      if (do_stage) {
        gdouble* Xmu_p = pk_getp(pkv, PK_XMU);
        gdouble* d2_p = pk_getp(pkv, PK_D2);
        gdouble* g0_p = pk_getp(pkv, PK_G0);
        auto stage_load = [&](int e) -> double {
          const int sk = e / RD, sj = e - sk * RD;
          int64_t nk = r0 + (int64_t)ks * NT_ + sk;
          nk = (nk < r1) ? nk : r1 - 1;
          if (sj < QP) return d2_p[nk * QP + sj];
          if (sj < 2 * QP) return Xmu_p[nk * QP + (sj - QP)];
          if (sj == 2 * QP) return g0_p[nk];
          return 0.0;
        };
        if (stager) staged = stage_load(tv);
        if (stager2) staged2 = stage_load(tv + PSI2_T4);
      }
      double g_old_mu, g_old_S;
      g_old_load(pkv, jr, g_old_mu, g_old_S);

      const double* rk = rowd + (k % PSI2_RING_ROWD) * NT_ * RD + team * RD;
      const double* rn = rowd + ((k + 1) % PSI2_RING_ROWD) * NT_ * RD + team * RD;
      const double* rh = rowd + ((k + hoff) % PSI2_RING_ROWD) * NT_ * RD + team * RD;
      double* wres = P2_RES(M_, NT_, TS_) + ((size_t)(k % PSI2_RING_RES) * 8 + wave) * 2 * RW;
      double* wred = P2_RED(M_, NT_, TS_) + wave * NV * RS;
      const double wsA = wave_sum_dpp(inA ? v : 0.0);
      const double wsB = (bnd < 64) ? wave_sum_dpp(inA ? 0.0 : v) : 0.0;
      if (lane == 0) { wres[2 * QP] = wsA; wres[RW + 2 * QP] = wsB; }
      const int jv = lane / G, part = lane % G;
      // lanes without an inducing row add 0.0 to a slot of their own (no branch around the atomic: a chunk stays one
      // basic block, so its LDS reads are issued together and not one round trip after the other)
      const int trz = act ? tr : (trv % M_);
      double* dzp = dzacc + trz;
      // the eight (or sixteen) lanes that hold the finished sums of a value row store them; the other lanes store into the
      // transposition scratch, which the next chunk overwrites
      double h0 = 0.0, h1 = 0.0;
      double zn[QC];
#pragma unroll
      for (int j = 0; j < QC; ++j) zn[j] = zrow(j);
#pragma unroll
      for (int q0 = 0; q0 < QP; q0 += QC) {
        double zc[QC];
#pragma unroll
        for (int j = 0; j < QC; ++j) zc[j] = zn[j];
        if (q0 + QC < QP) {
#pragma unroll
          for (int j = 0; j < QC; ++j) zn[j] = zrow(q0 + QC + j);
        }
        asm volatile("" ::: "memory");                  // one column group of loads in flight, not all of them
        // row operands of the chunk's columns, all requested before anything waits for one of them
        double dk[QC], mk[QC], dn[QC], dh[QC], mh[QC], lm[QC];
#pragma unroll
        for (int j = 0; j < QC; ++j) {
          const int q = q0 + j;
          dk[j] = rk[q]; mk[j] = rk[QP + q]; dn[j] = rn[q]; dh[j] = rh[q]; mh[j] = rh[QP + q]; lm[j] = lamL[q];
        }
#pragma unroll
        for (int j = 0; j < QC; ++j) {
          const double z = zc[j];
          wred[j * RS + lane] = v * z;
          wred[(QC + j) * RS + lane] = 0.5 * z * (v * z + y[q0 + j]);
        }
        __builtin_amdgcn_wave_barrier();
        double x[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = wred[jv * RS + part + G * i];
        __builtin_amdgcn_wave_barrier();
        // dZ update of iteration k, then the coming rows' dzl and h for these columns
#pragma unroll
        for (int j = 0; j < QC; ++j) {
          const int q = q0 + j;
          const double z = zc[j];
          const double dzv = dk[j] * (v * (2.0 * mk[j] - z) - y[q]);
          atomicAdd(dzp + q * DZS, act ? dzv : 0.0);
          dzl[q] = -0.5 * z * (dn[j] - lm[j]);
          const double hz = z * dh[j] * (mh[j] - 0.25 * z);
          if (j & 1) h1 += hz; else h0 += hz;
          asm volatile("" : "+v"(dzl[q]));              // computed here, not sunk to the end of the pass with its operands spilled
        }
        asm volatile("" : "+v"(h0), "+v"(h1));
        double sA = 0.0, sB = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const bool a = part + G * i < bnd;
          sA += a ? x[i] : 0.0;
          sB += a ? 0.0 : x[i];
        }
        sA = sum8_dpp(sA);
        if (G == 16) sA += dpp_f64<0x140>(sA);
        if (bnd < 64) {
          sB = sum8_dpp(sB);
          if (G == 16) sB += dpp_f64<0x140>(sB);
        }
        {
          const int o = (jv < QC ? 0 : QP) + q0 + (jv % QC);
          double* pa = (part == 0) ? wres + o : wred + lane;
          double* pb = (part == 0) ? wres + RW + o : wred + RS + lane;
          *pa = sA;
          *pb = sB;
        }
      }
      if (act) hbuf[((k + hoff) & 1) * NT_ * TS_ + team * TS_ + tr] = (h0 + h1) - ellq;
      const double hn = hbuf[((k + 1) & 1) * NT_ * TS_ + team * TS_ + tr];      // this thread's own value (role 1: from its last pass)
      base = (act && r0 + (int64_t)(k + 1) * NT_ + team < r1) ? (rn[2 * QP] + hn) : -INFINITY;
      if (do_stage) {
        double* rs = rowd + (ks % PSI2_RING_ROWD) * NT_ * RD;
        if (stager) rs[tv] = staged;
        if (stager2) rs[tv + PSI2_T4] = staged2;
      }
      if (jr >= 0) rowgrad(pkv, jr, g_old_mu, g_old_S);
#if PSI2_PRIO_PASS >= 0
      if (role) __builtin_amdgcn_s_setprio(PSI2_PRIO_SWEEP1); else __builtin_amdgcn_s_setprio(PSI2_PRIO_SWEEP0);
#endif
    }
    STAMP(tP);
    if (!role) __syncthreads();                         // b_k for role 0
    STAMP(tW);
  }
  __syncthreads();                                      // the last passes of role 1 are done
#ifdef PSI2_STAMPS
  if (dbgout && (t & 63) == 0) {
    double* o = dbgout + 2048 + (blockIdx.x * 8 + (t >> 6)) * 4;
    o[0] = (double)tS; o[1] = (double)tP; o[2] = (double)tW; o[3] = (double)K; dbgout[2048 + 256 * 8 * 4 + blockIdx.x * 8 + (t >> 6)] = (double)tH;
  }
#endif
  {
    int pkv = pk;
    asm volatile("" : "+v"(pkv));
    const int role_ = pk_get(pkv, PK_ROLE), K_ = pk_get(pkv, PK_K);
    for (int j = (K_ - 2 + role_ > 0 ? K_ - 2 + role_ : 0); j < K_; ++j) {
      double gm, gs;
      g_old_load(pkv, j, gm, gs);
      rowgrad(pkv, j, gm, gs);
    }
    const int M_ = pk_get(pkv, PK_M), NT_ = pk_get(pkv, PK_NT), Q_ = pk_get(pkv, PK_Q);
    const int tri = P2_TRI(M_), DZS = P2_DZS(M_);
    const double* p2acc = lds;
    const double* dzacc = P2_DZACC(M_);
    gdouble* out = pk_getp(pkv, PK_P2) + (size_t)blockIdx.x * tri;
    for (int i = t; i < tri; i += PSI2_T4) out[i] = p2acc[i];
    gdouble* outz = pk_getp(pkv, PK_DZ) + (size_t)blockIdx.x * QP * RGP_PSI2_TEAM;
    for (int i = t; i < QP * RGP_PSI2_TEAM; i += PSI2_T4) {
      const int q = i / RGP_PSI2_TEAM, m = i - q * RGP_PSI2_TEAM;
      outz[i] = (m < M_) ? dzacc[q * DZS + m] : 0.0;
    }
    const int team = team_raw < NT_ ? team_raw : NT_ - 1;
    if (team_raw < NT_ && tr_raw < QP) pk_getp(pkv, PK_DLAM)[((size_t)blockIdx.x * NT_ + team) * QP + tr_raw] = (tr_raw < Q_) ? dlam_acc : 0.0;
    if (team_raw < NT_ && tr_raw == 0) pk_getp(pkv, PK_WSUM)[(size_t)blockIdx.x * NT_ + team] = wsum_acc;
  }
}
#undef zrow

static int psi2_team_stride(int M) { return M < 64 ? 64 : M; }

static size_t psi2_rowsN_lds(int M, int QP, int NT) {
  const size_t tri = (size_t)M * (M + 1) / 2;
  return (tri + (size_t)QP * RGP_PSI2_TEAM + (size_t)NT * RGP_PSI2_TEAM + (size_t)NT * (2 * (size_t)QP + 2) +
          8 * RED_WAVE + 8 * 2 * (2 * (size_t)QP + 8) + PSI2_EXP_TAB) * sizeof(double);
}

// LDS bytes of the two-role kernel for QC columns per reduction chunk (0: the shape does not fit)
static size_t psi2_rowsP_lds(int M, int QP, int NT, int QC) {
  const size_t tri = (size_t)M * (M + 1) / 2, ts = psi2_team_stride(M), dzs = (M + 3) & ~3, rw = 2 * (size_t)QP + 2;
  const size_t n = tri + QP * dzs + 2 * NT * ts + PSI2_RING_ROWD * NT * rw + 8 * (2 * (size_t)QC) * (QC == 4 ? 72 : 80) +
                   PSI2_RING_RES * 8 * 2 * rw + PSI2_EXP_TAB + QP;
  return n * sizeof(double);
}
static int psi2_rowsP_qc(int M, int QP, int NT) {
  static const bool off = getenv("RGP_PSI2_NO_SKEW") != nullptr;
  if (off) return 0;
  const size_t cap = 160 * 1024 - 64;                     // the kernel's static LDS (role tickets) comes on top
  if (psi2_rowsP_lds(M, QP, NT, 4) <= cap) return 4;
  if (psi2_rowsP_lds(M, QP, NT, 2) <= cap) return 2;
  return 0;
}

// number of packed teams (0: use the four-team / two-team kernels)
int psi2_packed_teams(int M, int QP) {
  static const bool off = getenv("RGP_PSI2_NO_PACKED") != nullptr;
  if (off || M < 1 || M > RGP_MAX_M) return 0;
  const int nt = PSI2_T4 / psi2_team_stride(M);
  if (nt > PSI2_MAX_NT || (size_t)nt * (2 * (size_t)QP + 2) > 2 * PSI2_T4) return 0;
  if (psi2_rowsN_lds(M, QP, nt) > 160 * 1024) return 0;
  return nt;
}

// LDS bytes of the four-team kernel
static size_t psi2_rows4_lds(int M, int QP) {
  const size_t tri = (size_t)M * (M + 1) / 2;
  return (tri + RGP_PSI2_TEAM + (size_t)QP * RGP_PSI2_TEAM + 4 * RGP_PSI2_TEAM + 8 * RED_WAVE + 8 * (2 * (size_t)QP + 8) +
          PSI2_EXP_TAB /* static table */) * sizeof(double);
}

bool psi2_use_rows4(int M, int QP) {
  static const bool off = getenv("RGP_PSI2_NO_ROWS4") != nullptr;
  return !off && psi2_rows4_lds(M, QP) <= 160 * 1024;
}

Psi2Plan psi2_plan(rgp_handle h, int64_t N, int M, int QP, bool fwd, bool bwd) {
  Psi2Plan pl;
  PairGeom pg;
  if (fwd && bwd && psi2_pair_selected(h, N, M, QP, pg)) {       // pair form: partials per row range / per row-kernel workgroup
    pl.grid = pg.G * pg.RR;
    pl.n_p2 = pg.RR; pl.n_dz = pg.NB + 1; pl.n_dl = pg.NB;
    return pl;
  }
  const int nt = (fwd && bwd && !h->deterministic) ? psi2_packed_teams(M, QP) : 0;
  if (nt >= 4) {
    int64_t g = h->num_cu - (h->overlap && h->num_cu > 64 ? 1 : 0);     // one 512-thread workgroup per CU, nt rows in flight;
                                                                          // one CU stays free for the side-stream M x M chains
    // Short passes leave CUs free in every XCD.  A workgroup of the psi2 pass owns its CU (LDS and registers), so a side-stream
    // kernel can only start on a CU without one; the dispatcher hands each new kernel's first workgroup to the next XCD of a
    // global round-robin (tools/coresidency_probe.hip), so a single-workgroup M x M chain kernel whose turn falls on a full XCD
    // waits for a psi2 workgroup to drain -- the whole pass.  With 8 x 29 workgroups the chains of the other layers never
    // wait; the fewer sweeps a workgroup has, the more that latency matters and the less the idle CUs cost
    // (profiles/r1_grid_ab.txt: +19 % per step at 2 sequences per rank, +10 % at 4, +5 % at 8, +3 % at 16).
    if (h->overlap && h->num_cu == 256) {
      const int64_t sweeps = (N + g * nt - 1) / (g * nt);
      // multiples of 8 only: 233 workgroups (one sweep fewer at 4 sequences per rank) run 13 % slower per step than 232, because
      // the dispatcher's round-robin position then moves on by the remainder after every pass
      if (sweeps <= 16) g = 232;
      else if (sweeps <= 32) g = 240;
      else if (sweeps <= 64) g = 248;
      else g = 256;      // long passes: the chains of a step are issued a whole step ahead and find their CU between two passes;
                         // 256 measured +0.9 % per step over 255 at 32 sequences (103 sweeps)
    }
    { static int env_g = -1; if (env_g < 0) { const char* e = getenv("RGP_PSI2_GRID"); env_g = e ? atoi(e) : 0; } if (env_g > 0) g = env_g; }   // tuning override
    if (g > (N + nt - 1) / nt) g = (N + nt - 1) / nt;
    pl.grid = (int)(g < 1 ? 1 : g);
    pl.n_p2 = pl.grid; pl.n_dz = pl.grid; pl.n_dl = nt * pl.grid;
  } else if (fwd && bwd && psi2_use_rows4(M, QP)) {
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
static hipError_t psi2_dispatch(hipStream_t s, bool fwd, bool bwd, const Psi2Args& a, int nblk, bool packed_ok) {
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
  const int nt = (fwd && bwd && packed_ok) ? psi2_packed_teams(a.M, QP) : 0;
  if (nt >= 4) {
    const size_t ln = psi2_rowsN_lds(a.M, QP, nt);
    double* dbgbuf = nullptr;
#ifdef PSI2_STAMPS                                   // per-stage cycle stamps (profiling builds: make CXXFLAGS+=-DPSI2_STAMPS, RGP_STAMPS=1)
    static double* dbgbuf_s = nullptr;
    if (!dbgbuf_s && getenv("RGP_STAMPS")) (void)hipMalloc(&dbgbuf_s, (2048 + 256 * 8 * 5 + 256 * 9) * sizeof(double));
    dbgbuf = dbgbuf_s;
#endif
    const int qc = psi2_rowsP_qc(a.M, QP, nt);
    if (qc) {
      const size_t lp = psi2_rowsP_lds(a.M, QP, nt, qc);
      const int64_t per = ((a.N + nblk - 1) / nblk + nt - 1) / nt * nt;       // rows per workgroup, a multiple of the teams
      const void* fn = (qc == 4) ? (const void*)psi2_rowsP_kernel<QP, 4> : (const void*)psi2_rowsP_kernel<QP, 2>;
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp);
      if (e != hipSuccess) return e;
      if (qc == 4)
        hipLaunchKernelGGL((psi2_rowsP_kernel<QP, 4>), dim3(nblk), dim3(PSI2_T4), lp, s, a.N, per, a.M, a.Q, nt, psi2_team_stride(a.M), a.Zc,
                           a.lam, a.dLs, a.Xmu, a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu,
                           a.o.gS, a.o.accumulate, dbgbuf);
      else
        hipLaunchKernelGGL((psi2_rowsP_kernel<QP, 2>), dim3(nblk), dim3(PSI2_T4), lp, s, a.N, per, a.M, a.Q, nt, psi2_team_stride(a.M), a.Zc,
                           a.lam, a.dLs, a.Xmu, a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu,
                           a.o.gS, a.o.accumulate, dbgbuf);
#ifdef PSI2_STAMPS
      if (dbgbuf) {
        (void)hipStreamSynchronize(s);
        static double hb[2048 + 256 * 8 * 5];
        (void)hipMemcpy(hb, dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        double acc[2][4] = {{0}}; int cnt[2] = {0, 0}; double iters = 0;
        for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) {
          const int r = (int)hb[b * 8 + w] % 10;
          const double* o = hb + 2048 + (b * 8 + w) * 4;
          for (int i = 0; i < 3; ++i) acc[r][i] += o[i] / (o[3] > 0 ? o[3] : 1);
          acc[r][3] += hb[2048 + 256 * 8 * 4 + b * 8 + w] / (o[3] > 0 ? o[3] : 1);
          ++cnt[r]; iters = o[3];
        }
        fprintf(stderr, "psi2P QP=%d iters=%.0f wave0..7 simd/role:", QP, iters);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " %02d", (int)hb[w]);
        for (int r = 0; r < 2; ++r)
          fprintf(stderr, " | role %d (%d waves) cycles/iter: sweep %.0f (first half of the pairs %.0f) pass %.0f barrier %.0f", r, cnt[r], acc[r][0] / cnt[r], acc[r][3] / cnt[r], acc[r][1] / cnt[r], acc[r][2] / cnt[r]);
        fprintf(stderr, "\n");
      }
#endif
      return hipSuccess;
    }
    hipError_t e = hipFuncSetAttribute((const void*)psi2_rowsN_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((psi2_rowsN_kernel<QP>), dim3(nblk), dim3(PSI2_T4), ln, s, a.N, a.M, a.Q, nt, psi2_team_stride(a.M), a.Zc, a.lam, a.dLs, a.Xmu,
                       a.d2, a.g0, a.o.p2_part, a.o.dz_part, a.o.dlam_part, a.o.wsum_part, a.o.gmu, a.o.gS,
                       a.o.accumulate, dbgbuf);
#ifdef PSI2_STAMPS
    if (dbgbuf) {
      (void)hipStreamSynchronize(s);
      static double hb[256 * 9];
      (void)hipMemcpy(hb, dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
      double a[7] = {0};
      for (int b = 0; b < nblk; ++b) for (int k = 0; k < 7; ++k) a[k] += hb[b * 8 + k] / nblk;
      double th = 0; for (int b = 0; b < nblk; ++b) th += hb[256 * 8 + b] / nblk;
      fprintf(stderr, "psi2N QP=%d iters=%.0f cycles/iter: stage %.0f prologue %.0f sweep %.0f red %.0f imbalance %.0f ycalc %.0f dz %.0f rowgrad %.0f\n", QP, a[6],
              a[0] / a[6], a[1] / a[6], a[2] / a[6], a[3] / a[6], hb[7] / a[6], th / a[6], a[4] / a[6], a[5] / a[6]);
    }
#endif
    return hipSuccess;
  }
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
  PairGeom pg;
  if (fwd && bwd && out.pair_ws && psi2_pair_selected(h, N, M, zt.QP, pg))
    return launch_psi2_pair(h, pg, N, Q, zt, rt, dLs, out, out.pair_ws);
  Psi2Args a{};
  a.N = N; a.M = M; a.Q = Q; a.Zc = zt.Zc; a.ZcT = zt.ZcT; a.zd = zt.zd; a.lam = zt.lam; a.dLs = dLs;
  a.Xmu = rt.Xmu; a.d2 = rt.d2; a.g0 = rt.g0; a.o = out;
  hipStream_t s = h->stream;
  hipError_t e = hipSuccess;
  switch (zt.QP) {
    case 8:  e = psi2_dispatch<8>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 16: e = psi2_dispatch<16>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 20: e = psi2_dispatch<20>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 24: e = psi2_dispatch<24>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 32: e = psi2_dispatch<32>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 40: e = psi2_dispatch<40>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 48: e = psi2_dispatch<48>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 56: e = psi2_dispatch<56>(s, fwd, bwd, a, nblk, !h->deterministic); break;
    case 64: e = psi2_dispatch<64>(s, fwd, bwd, a, nblk, !h->deterministic); break;
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
    int b = g;
    for (; b + 56 < nblk; b += 64) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = part[(size_t)(b + 8 * u) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += x[u];
    }
    for (; b < nblk; b += 8) s += part[(size_t)b * stride + i];
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

// Several reductions in one launch (same arithmetic and order as reduce_kernel): the partial arrays of one pass are
// reduced together, which removes 3-4 launches of latency-bound 30 us kernels per pass from the step's critical path.
__global__ void __launch_bounds__(256) reduce_batch_kernel(ReduceBatch rb) {
  __shared__ double sm[8][33];
  int k = 0;
#pragma unroll
  for (int j = 1; j < REDUCE_BATCH_MAX; ++j)
    if (j < rb.nseg && (int)blockIdx.x >= rb.blk_off[j]) k = j;
  const ReduceSeg sg = rb.seg[k];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t i = (int64_t)(blockIdx.x - rb.blk_off[k]) * 32 + e;
  double s = 0.0;
  if (i < sg.len) {
    // same summation order as a plain loop; the loads of eight partials are in flight together
    int b = g;
    for (; b + 56 < sg.nblk; b += 64) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = sg.part[(size_t)(b + 8 * u) * sg.stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += x[u];
    }
    for (; b < sg.nblk; b += 8) s += sg.part[(size_t)b * sg.stride + i];
  }
  sm[g][e] = s;
  __syncthreads();
  if (g == 0 && i < sg.len) {
    double tot = 0.0;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) tot += sm[kk][e];
    sg.out[i] = tot;
  }
}

void launch_reduce_batch(hipStream_t s, int nseg, const ReduceSeg* segs) {
  ReduceBatch rb{};
  rb.nseg = nseg;
  int off = 0;
  for (int j = 0; j < nseg; ++j) {
    rb.seg[j] = segs[j];
    rb.blk_off[j] = off;
    off += (int)((segs[j].len + 31) / 32);
  }
  hipLaunchKernelGGL(reduce_batch_kernel, dim3(off), dim3(256), 0, s, rb);
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
