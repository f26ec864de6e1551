// rgp_predict.hip -- free-run prediction with moment matching on the device (Layer.freerun, autoreg/layers.py:1551-1571,
// which calls GPy SparseGP._raw_predict with uncertain inputs at every step; restated from upstream GPy -- "parity
// unpinned" by the reference, pinned against oracle/model.py::_raw_predict).
//
// Strictly sequential in time: one persistent 512-thread workgroup per sequence runs the whole time loop; per step it
// evaluates psi1* (M) and psi2n* (M x M) of the single input row and contracts them with the posterior:
//   mean_d = sum_m psi1_m wv_md
//   var_d  = sum_mm' (psi2n - psi1 psi1^T)_mm' wv_md wv_m'd + variance - sum_mm' Winv_mm' psi2n_mm'   (clipped at 1e-15)
// Thread (m, part) owns inducing row m and a quarter of the m' sweep; z[m',:], wv[m',:] are wave-uniform (SGPR operands).
#include <math.h>

#include "rgp_psi.h"

#define FR_THREADS 512
#define FR_DC 16            // outputs handled per sweep (register accumulators); D > 16 repeats the sweep

struct FreerunArgs {
  int n_seq, M, Q, D, X_win, X_dim, U_win, U_dim, steps;
  int64_t u_rows, u_skip;
  const double *Zc, *lam, *c, *variance, *winv, *wv;
  const double *init_mean, *init_var, *U_mean, *U_var;
  double u_var_const;
  double *X_mean, *X_var;
};

template <int QP>
__global__ void __launch_bounds__(FR_THREADS) freerun_kernel(FreerunArgs a, const double* __restrict__ Zc,
                                                              const double* __restrict__ wv_g,
                                                              const double* __restrict__ winv_g) {
  __shared__ double xin_m[QP], d2s[QP], dmls[QP], e1s[QP], mus[QP];
  __shared__ double hbuf[RGP_PSI2_TEAM], p1buf[RGP_PSI2_TEAM];
  __shared__ double red[8][FR_DC + 2];
  __shared__ double scratch[16];
  __shared__ double gsh[2];
  const int M = a.M, Q = a.Q, D = a.D;
  const int t = threadIdx.x, tr = t & (RGP_PSI2_TEAM - 1);
  const int part = __builtin_amdgcn_readfirstlane(t >> 7);            // 4 parts
  const bool act = tr < M;
  const int tm = act ? tr : M - 1;
  const int seq = blockIdx.x;
  const int Mq = (M + 3) / 4;
  const int mp0 = part * Mq, mp1 = (mp0 + Mq < M) ? mp0 + Mq : M;
  const int Qx = a.X_win * a.X_dim;
  const int64_t T_out = a.X_win + a.steps;
  double* Xm = a.X_mean + (size_t)seq * T_out * a.X_dim;
  double* Xv = a.X_var + (size_t)seq * T_out * a.X_dim;
  const double* Um = a.U_mean ? a.U_mean + ((size_t)seq * a.u_rows + a.u_skip) * a.U_dim : nullptr;
  const double* Uv = a.U_var ? a.U_var + ((size_t)seq * a.u_rows + a.u_skip) * a.U_dim : nullptr;
  const double var = a.variance[0], logvar = log(var);

  // initial window (layers.py:1543-1555): given, or N(0, 1)
  for (int i = t; i < Qx; i += FR_THREADS) {
    Xm[i] = a.init_mean ? a.init_mean[(size_t)seq * Qx + i] : 0.0;
    Xv[i] = a.init_var ? a.init_var[(size_t)seq * Qx + i] : 1.0;
  }
  double z[QP];
  double ellq = 0.0;
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    z[q] = Zc[(size_t)tm * QP + q];
    ellq += 0.25 * a.lam[q] * z[q] * z[q];
  }
  __syncthreads();

  for (int n = 0; n < a.steps; ++n) {
    // ---- the input row: [X[n .. n+X_win) | U[n .. n+U_win)] ----
    double sl2 = 0.0, sl1 = 0.0, sdm = 0.0;
    if (t < QP) {
      double mu = 0.0, d2 = 0.0, e1 = 0.0, dml = 0.0;
      if (t < Q) {
        double m, S;
        if (t < Qx) { m = Xm[(size_t)n * a.X_dim + t]; S = Xv[(size_t)n * a.X_dim + t]; }
        else { const int64_t i = (int64_t)n * a.U_dim + (t - Qx); m = Um[i]; S = Uv ? Uv[i] : a.u_var_const; }
        const double lam = a.lam[t];
        mu = m - a.c[t];
        const double t2 = 1.0 + 2.0 * S * lam, t1 = 1.0 + S * lam;
        d2 = lam / t2; e1 = lam / t1; dml = -2.0 * S * lam * d2;
        sl2 = log(t2); sl1 = log(t1); sdm = d2 * mu * mu;
      }
      mus[t] = mu; d2s[t] = d2; e1s[t] = e1; dmls[t] = dml;
    }
    sl2 = block_sum(sl2, scratch); sl1 = block_sum(sl1, scratch); sdm = block_sum(sdm, scratch);
    const double g0 = 2.0 * logvar - 0.5 * sl2 - sdm, g1 = logvar - 0.5 * sl1;
    __syncthreads();
    // ---- per inducing row: dzl, h, psi1 ----
    double dzl[QP];
    double h = 0.0, e = 0.0;
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      dzl[q] = z[q] * dmls[q];
      h += z[q] * d2s[q] * (mus[q] - 0.25 * z[q]);
      const double df = mus[q] - z[q];
      e += e1s[q] * df * df;
    }
    h -= ellq;
    const double p1 = exp(g1 - 0.5 * e);
    if (part == 0) { hbuf[tr] = h; p1buf[tr] = act ? p1 : 0.0; }
    const double base = act ? (g0 + h) : -INFINITY;
    __syncthreads();
    // ---- sweep over m' (repeated per chunk of FR_DC outputs) ----
    for (int d0 = 0; d0 < D; d0 += FR_DC) {
      const int dc = (D - d0 < FR_DC) ? D - d0 : FR_DC;
      double wvm[FR_DC], acc[FR_DC];
#pragma unroll
      for (int d = 0; d < FR_DC; ++d) { wvm[d] = (d < dc) ? wv_g[(size_t)tm * D + d0 + d] : 0.0; acc[d] = 0.0; }
      double trc = 0.0;
      for (int mp = mp0; mp < mp1; ++mp) {
        const double* __restrict__ zs = Zc + (size_t)mp * QP;
        const double* __restrict__ ws = wv_g + (size_t)mp * D + d0;
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int q = 0; q < QP; q += 2) { c0 += dzl[q] * zs[q]; c1 += dzl[q + 1] * zs[q + 1]; }
        const double p = exp((base + hbuf[mp]) - 0.5 * (c0 + c1));
        const double tt = p - (act ? p1 : 0.0) * p1buf[mp];
        trc += winv_g[(size_t)mp * M + tm] * p;
#pragma unroll
        for (int d = 0; d < FR_DC; ++d)
          if (d < dc) acc[d] += tt * wvm[d] * ws[d];
      }
      // ---- reductions over all threads: acc[d], trace, mean ----
      const int wave = t >> 6, lane = t & 63;
#pragma unroll
      for (int d = 0; d < FR_DC; ++d) {
        const double s = wave_sum(acc[d]);
        if (lane == 0) red[wave][d] = s;
      }
      {
        const double s = wave_sum(trc);
        if (lane == 0) red[wave][FR_DC] = s;
      }
      __syncthreads();
      if (t < dc) {
        double qf = 0.0, tr_ = 0.0, mean = 0.0;
        for (int w = 0; w < 8; ++w) { qf += red[w][t]; tr_ += red[w][FR_DC]; }
        for (int m = 0; m < M; ++m) mean += p1buf[m] * wv_g[(size_t)m * D + d0 + t];
        double v = qf + var - tr_;
        v = fmax(v, 1e-15);                               // GPy _raw_predict clip
        if (v <= 0.0) v = 1e-10;                          // layers.py:1570
        Xm[(size_t)(a.X_win + n) * a.X_dim + d0 + t] = mean;
        Xv[(size_t)(a.X_win + n) * a.X_dim + d0 + t] = v;
      }
      __syncthreads();
    }
  }
}

// M > 128 (up to FRB_MAX_M): thread = inducing row m (blocks of 512 rows when M > 512), every thread sweeps all m' of its part;
// for M <= 256 the 512 threads split the m' sweep into parts.  z[m,:] is re-read per step (L2 resident) instead of held in
// registers; same arithmetic as freerun_kernel.
#define FRB_MAX_M 1024
template <int QP>
__global__ void __launch_bounds__(FR_THREADS) freerun_big_kernel(FreerunArgs a, const double* __restrict__ Zc,
                                                                  const double* __restrict__ wv_g,
                                                                  const double* __restrict__ winv_g) {
  __shared__ double d2s[QP], dmls[QP], e1s[QP], mus[QP];
  __shared__ double hbuf[FRB_MAX_M], p1buf[FRB_MAX_M];
  __shared__ double red[8][FR_DC + 2];
  __shared__ double scratch[16];
  const int M = a.M, Q = a.Q, D = a.D;
  const int t = threadIdx.x;
  const int ru = (M + 63) / 64 * 64;                                   // lanes per part (whole waves)
  const int parts = (ru <= FR_THREADS / 2) ? FR_THREADS / ru : 1;
  const int part = __builtin_amdgcn_readfirstlane(parts > 1 ? t / ru : 0);
  const int tl = parts > 1 ? t - part * ru : t;
  const bool in_part = part < parts;
  const int Mq = (M + parts - 1) / parts;
  const int mp0 = in_part ? part * Mq : 0, mp1 = in_part ? ((mp0 + Mq < M) ? mp0 + Mq : M) : 0;
  const int mstep = parts > 1 ? M : FR_THREADS;                        // parts > 1: one pass (m = tl); else blocks of 512 rows
  const int seq = blockIdx.x;
  const int Qx = a.X_win * a.X_dim;
  const int64_t T_out = a.X_win + a.steps;
  double* Xm = a.X_mean + (size_t)seq * T_out * a.X_dim;
  double* Xv = a.X_var + (size_t)seq * T_out * a.X_dim;
  const double* Um = a.U_mean ? a.U_mean + ((size_t)seq * a.u_rows + a.u_skip) * a.U_dim : nullptr;
  const double* Uv = a.U_var ? a.U_var + ((size_t)seq * a.u_rows + a.u_skip) * a.U_dim : nullptr;
  const double var = a.variance[0], logvar = log(var);
  for (int i = t; i < Qx; i += FR_THREADS) {
    Xm[i] = a.init_mean ? a.init_mean[(size_t)seq * Qx + i] : 0.0;
    Xv[i] = a.init_var ? a.init_var[(size_t)seq * Qx + i] : 1.0;
  }
  __syncthreads();

  for (int n = 0; n < a.steps; ++n) {
    double sl2 = 0.0, sl1 = 0.0, sdm = 0.0;
    if (t < QP) {
      double mu = 0.0, d2 = 0.0, e1 = 0.0, dml = 0.0;
      if (t < Q) {
        double m, S;
        if (t < Qx) { m = Xm[(size_t)n * a.X_dim + t]; S = Xv[(size_t)n * a.X_dim + t]; }
        else { const int64_t i = (int64_t)n * a.U_dim + (t - Qx); m = Um[i]; S = Uv ? Uv[i] : a.u_var_const; }
        const double lam = a.lam[t];
        mu = m - a.c[t];
        const double t2 = 1.0 + 2.0 * S * lam, t1 = 1.0 + S * lam;
        d2 = lam / t2; e1 = lam / t1; dml = -2.0 * S * lam * d2;
        sl2 = log(t2); sl1 = log(t1); sdm = d2 * mu * mu;
      }
      mus[t] = mu; d2s[t] = d2; e1s[t] = e1; dmls[t] = dml;
    }
    sl2 = block_sum(sl2, scratch); sl1 = block_sum(sl1, scratch); sdm = block_sum(sdm, scratch);
    const double g0 = 2.0 * logvar - 0.5 * sl2 - sdm, g1 = logvar - 0.5 * sl1;
    __syncthreads();
    // ---- per inducing row: h, psi1 ----
    for (int m = t; m < M; m += FR_THREADS) {
      const double* __restrict__ zr = Zc + (size_t)m * QP;
      double h = 0.0, e = 0.0;
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const double z = zr[q];
        h += z * d2s[q] * (mus[q] - 0.25 * z) - 0.25 * a.lam[q] * z * z;
        const double df = mus[q] - z;
        e += e1s[q] * df * df;
      }
      hbuf[m] = h;
      p1buf[m] = exp(g1 - 0.5 * e);
    }
    __syncthreads();
    for (int d0 = 0; d0 < D; d0 += FR_DC) {
      const int dc = (D - d0 < FR_DC) ? D - d0 : FR_DC;
      double acc[FR_DC];
#pragma unroll
      for (int d = 0; d < FR_DC; ++d) acc[d] = 0.0;
      double trc = 0.0;
      for (int mb = 0; mb < M; mb += mstep) {
        const int m = mb + tl;
        const bool act = in_part && tl < mstep && m < M;
        const int tm = act ? m : M - 1;
        const double* __restrict__ zr = Zc + (size_t)tm * QP;
        double dzl[QP], wvm[FR_DC];
#pragma unroll
        for (int q = 0; q < QP; ++q) dzl[q] = zr[q] * dmls[q];
#pragma unroll
        for (int d = 0; d < FR_DC; ++d) wvm[d] = (d < dc) ? wv_g[(size_t)tm * D + d0 + d] : 0.0;
        const double base = act ? (g0 + hbuf[tm]) : -INFINITY;
        const double p1 = act ? p1buf[tm] : 0.0;
        for (int mp = mp0; mp < mp1; ++mp) {
          const double* __restrict__ zs = Zc + (size_t)mp * QP;
          const double* __restrict__ ws = wv_g + (size_t)mp * D + d0;
          double c0 = 0.0, c1 = 0.0;
#pragma unroll
          for (int q = 0; q < QP; q += 2) { c0 += dzl[q] * zs[q]; c1 += dzl[q + 1] * zs[q + 1]; }
          const double p = exp((base + hbuf[mp]) - 0.5 * (c0 + c1));
          const double tt = p - p1 * p1buf[mp];
          trc += winv_g[(size_t)mp * M + tm] * p;
#pragma unroll
          for (int d = 0; d < FR_DC; ++d)
            if (d < dc) acc[d] += tt * wvm[d] * ws[d];
        }
      }
      const int wave = t >> 6, lane = t & 63;
#pragma unroll
      for (int d = 0; d < FR_DC; ++d) {
        const double sv = wave_sum(acc[d]);
        if (lane == 0) red[wave][d] = sv;
      }
      {
        const double sv = wave_sum(trc);
        if (lane == 0) red[wave][FR_DC] = sv;
      }
      __syncthreads();
      if (t < dc) {
        double qf = 0.0, tr_ = 0.0, mean = 0.0;
        for (int w = 0; w < 8; ++w) { qf += red[w][t]; tr_ += red[w][FR_DC]; }
        for (int m = 0; m < M; ++m) mean += p1buf[m] * wv_g[(size_t)m * D + d0 + t];
        double v = qf + var - tr_;
        v = fmax(v, 1e-15);                               // GPy _raw_predict clip
        if (v <= 0.0) v = 1e-10;                          // layers.py:1570
        Xm[(size_t)(a.X_win + n) * a.X_dim + d0 + t] = mean;
        Xv[(size_t)(a.X_win + n) * a.X_dim + d0 + t] = v;
      }
      __syncthreads();
    }
  }
}

// Windows wider than 64 columns (any padded Q up to FRW_MAX_QP, any M up to FRB_MAX_M): the Gram terms of a step,
// c[m,m'] = sum_q dml[q] z[m,q] z[m',q], no longer fit a register copy of z[m,:].  The M x M pairs are visited in tiles of 128
// inducing rows x 128 columns: thread (m, part) keeps the FRW_MP accumulators of its quarter of the tile's columns in registers
// and walks q once per tile, reading z from a transposed copy ZT[q][Mld] (its own entry coalesced, the columns' entries
// wave-uniform); the input row's terms are in dynamic LDS (4 x QP doubles).  Same arithmetic as freerun_kernel.
#define FRW_MP 32
#define FRW_MAX_QP 1024
// ZT[q][m] = Zc[m][q], zero beyond M   (QP x Mld)
__global__ void __launch_bounds__(256) frw_transpose_kernel(int M, int QP, int Mld, const double* __restrict__ Zc, double* __restrict__ ZT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= QP * Mld) return;
  const int q = i / Mld, m = i % Mld;
  ZT[i] = m < M ? Zc[(size_t)m * QP + q] : 0.0;
}

__global__ void __launch_bounds__(FR_THREADS) freerun_wide_kernel(FreerunArgs a, const int QP, const int Mld, const double* __restrict__ ZT,
                                                                   const double* __restrict__ wv_g,
                                                                   const double* __restrict__ winv_g) {
  extern __shared__ double frw_rows[];
  double* d2s = frw_rows;
  double* dmls = frw_rows + QP;
  double* e1s = frw_rows + 2 * QP;
  double* mus = frw_rows + 3 * QP;
  __shared__ double hbuf[FRB_MAX_M], p1buf[FRB_MAX_M];
  __shared__ double red[8][FR_DC + 2];
  __shared__ double scratch[16];
  const int M = a.M, Q = a.Q, D = a.D;
  const int t = threadIdx.x, tr = t & (RGP_PSI2_TEAM - 1);
  const int part = __builtin_amdgcn_readfirstlane(t >> 7);            // 4 parts of 32 columns
  const int seq = blockIdx.x;
  const int Qx = a.X_win * a.X_dim;
  const int64_t T_out = a.X_win + a.steps;
  double* Xm = a.X_mean + (size_t)seq * T_out * a.X_dim;
  double* Xv = a.X_var + (size_t)seq * T_out * a.X_dim;
  const double* Um = a.U_mean ? a.U_mean + ((size_t)seq * a.u_rows + a.u_skip) * a.U_dim : nullptr;
  const double* Uv = a.U_var ? a.U_var + ((size_t)seq * a.u_rows + a.u_skip) * a.U_dim : nullptr;
  const double var = a.variance[0], logvar = log(var);

  for (int i = t; i < Qx; i += FR_THREADS) {
    Xm[i] = a.init_mean ? a.init_mean[(size_t)seq * Qx + i] : 0.0;
    Xv[i] = a.init_var ? a.init_var[(size_t)seq * Qx + i] : 1.0;
  }
  __syncthreads();

  for (int n = 0; n < a.steps; ++n) {
    // ---- the input row: [X[n .. n+X_win) | U[n .. n+U_win)] ----
    double sl2 = 0.0, sl1 = 0.0, sdm = 0.0;
    for (int q = t; q < QP; q += FR_THREADS) {
      double mu = 0.0, d2 = 0.0, e1 = 0.0, dml = 0.0;
      if (q < Q) {
        double m, S;
        if (q < Qx) { m = Xm[(size_t)n * a.X_dim + q]; S = Xv[(size_t)n * a.X_dim + q]; }
        else { const int64_t i = (int64_t)n * a.U_dim + (q - Qx); m = Um[i]; S = Uv ? Uv[i] : a.u_var_const; }
        const double lam = a.lam[q];
        mu = m - a.c[q];
        const double t2 = 1.0 + 2.0 * S * lam, t1 = 1.0 + S * lam;
        d2 = lam / t2; e1 = lam / t1; dml = -2.0 * S * lam * d2;
        sl2 += log(t2); sl1 += log(t1); sdm += d2 * mu * mu;
      }
      mus[q] = mu; d2s[q] = d2; e1s[q] = e1; dmls[q] = dml;
    }
    sl2 = block_sum(sl2, scratch); sl1 = block_sum(sl1, scratch); sdm = block_sum(sdm, scratch);
    const double g0 = 2.0 * logvar - 0.5 * sl2 - sdm, g1 = logvar - 0.5 * sl1;
    __syncthreads();
    // ---- per inducing row: h, psi1 ----
    for (int m = t; m < M; m += FR_THREADS) {
      double h = 0.0, e = 0.0;
      for (int q = 0; q < QP; ++q) {
        const double z = ZT[(size_t)q * Mld + m];
        h += z * d2s[q] * (mus[q] - 0.25 * z) - 0.25 * a.lam[q] * z * z;
        const double df = mus[q] - z;
        e += e1s[q] * df * df;
      }
      hbuf[m] = h;
      p1buf[m] = exp(g1 - 0.5 * e);
    }
    __syncthreads();
    // ---- the pairs, tile by tile (repeated per chunk of FR_DC outputs) ----
    for (int d0 = 0; d0 < D; d0 += FR_DC) {
      const int dc = (D - d0 < FR_DC) ? D - d0 : FR_DC;
      double acc[FR_DC];
#pragma unroll
      for (int d = 0; d < FR_DC; ++d) acc[d] = 0.0;
      double trc = 0.0;
      for (int mb = 0; mb < M; mb += RGP_PSI2_TEAM) {
        const int m = mb + tr;
        const bool act = m < M;
        const int tm = act ? m : M - 1;
        double wvm[FR_DC];
#pragma unroll
        for (int d = 0; d < FR_DC; ++d) wvm[d] = (d < dc) ? wv_g[(size_t)tm * D + d0 + d] : 0.0;
        const double base = act ? (g0 + hbuf[tm]) : -INFINITY;
        const double p1 = act ? p1buf[tm] : 0.0;
        for (int cb = 0; cb < M; cb += RGP_PSI2_TEAM) {
          const int mp0 = cb + part * FRW_MP;                          // wave-uniform
          double c[FRW_MP];
#pragma unroll
          for (int j = 0; j < FRW_MP; ++j) c[j] = 0.0;
          if (mp0 < M) {
            for (int q = 0; q < QP; ++q) {
              const double aq = ZT[(size_t)q * Mld + tm] * dmls[q];
              const double* __restrict__ zr = ZT + (size_t)q * Mld + mp0;
#pragma unroll
              for (int j = 0; j < FRW_MP; ++j) c[j] += aq * zr[j];
            }
          }
#pragma unroll
          for (int j = 0; j < FRW_MP; ++j) {
            const int mp = mp0 + j;
            if (mp < M) {
              const double* __restrict__ ws = wv_g + (size_t)mp * D + d0;
              const double p = exp((base + hbuf[mp]) - 0.5 * c[j]);
              const double tt = p - p1 * p1buf[mp];
              trc += winv_g[(size_t)mp * M + tm] * p;
#pragma unroll
              for (int d = 0; d < FR_DC; ++d)
                if (d < dc) acc[d] += tt * wvm[d] * ws[d];
            }
          }
        }
      }
      const int wave = t >> 6, lane = t & 63;
#pragma unroll
      for (int d = 0; d < FR_DC; ++d) {
        const double sv = wave_sum(acc[d]);
        if (lane == 0) red[wave][d] = sv;
      }
      {
        const double sv = wave_sum(trc);
        if (lane == 0) red[wave][FR_DC] = sv;
      }
      __syncthreads();
      if (t < dc) {
        double qf = 0.0, tr_ = 0.0, mean = 0.0;
        for (int w = 0; w < 8; ++w) { qf += red[w][t]; tr_ += red[w][FR_DC]; }
        for (int m = 0; m < M; ++m) mean += p1buf[m] * wv_g[(size_t)m * D + d0 + t];
        double v = qf + var - tr_;
        v = fmax(v, 1e-15);                               // GPy _raw_predict clip
        if (v <= 0.0) v = 1e-10;                          // layers.py:1570
        Xm[(size_t)(a.X_win + n) * a.X_dim + d0 + t] = mean;
        Xv[(size_t)(a.X_win + n) * a.X_dim + d0 + t] = v;
      }
      __syncthreads();
    }
  }
}

template <int QP>
static void freerun_big_launch(hipStream_t s, const FreerunArgs& a) {
  hipLaunchKernelGGL((freerun_big_kernel<QP>), dim3(a.n_seq), dim3(FR_THREADS), 0, s, a, a.Zc, a.wv, a.winv);
}

template <int QP>
static void freerun_launch(hipStream_t s, const FreerunArgs& a) {
  hipLaunchKernelGGL((freerun_kernel<QP>), dim3(a.n_seq), dim3(FR_THREADS), 0, s, a, a.Zc, a.wv, a.winv);
}

extern "C" rgp_status rgp_freerun(rgp_handle h, int n_seq, int M, int D, int X_win, int U_win, int U_dim, int steps,
                                  const double* Z, const double* variance, const double* inv_l, const double* woodbury_inv,
                                  const double* woodbury_vector, const double* init_mean, const double* init_var,
                                  const double* U_mean, const double* U_var, double u_var_const, int64_t u_rows,
                                  int64_t u_skip, double* X_mean, double* X_var) {
  if (!h) return RGP_ERR_BAD_ARG;
  const int X_dim = D;
  const int Q = X_win * X_dim + U_win * U_dim;
  const bool wide = rgp_qp(Q) > RGP_MAX_QP;
  if (M > FRB_MAX_M || Q <= 0 || rgp_qp(Q) > FRW_MAX_QP)
    return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_freerun: needs M <= 1024 and 0 < Q <= 1024 in this build");
  const bool big = M > RGP_MAX_M;
  if (U_win > 0 && (!U_mean || u_rows - u_skip < steps + U_win - 1))
    return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_freerun: not enough input rows for the requested steps");
  const int QP = rgp_qp(Q);
  // Z-derived terms in the handle's workspace
  rgp_status st;
  const int Mld = (M + RGP_PSI2_TEAM - 1) / RGP_PSI2_TEAM * RGP_PSI2_TEAM;
  const size_t need = ((size_t)2 * QP + (size_t)M * QP + (size_t)QP * RGP_PSI2_TEAM + 2 * (size_t)M * M + 2 + (wide ? (size_t)QP * Mld : 0)) * sizeof(double) + 10 * 256;
  if ((st = rgp_ensure_ws(h, need))) return st;
  WsAlloc ws(h->ws);
  ZTerms zt{};
  zt.M = M; zt.Q = Q; zt.QP = QP;
  zt.c = ws.take<double>(QP); zt.lam = ws.take<double>(QP); zt.Zc = ws.take<double>((size_t)M * QP);
  zt.ZcT = ws.take<double>((size_t)QP * RGP_PSI2_TEAM); zt.zd = ws.take<double>((size_t)M * M); zt.Kuu = ws.take<double>((size_t)M * M);
  double* ZT = wide ? ws.take<double>((size_t)QP * Mld) : nullptr;
  launch_prep_z(h->stream, M, Q, QP, Z, variance, inv_l, RGP_JITTER, zt);
  FreerunArgs a{};
  a.n_seq = n_seq; a.M = M; a.Q = Q; a.D = D; a.X_win = X_win; a.X_dim = X_dim; a.U_win = U_win; a.U_dim = U_dim; a.steps = steps;
  a.u_rows = u_rows; a.u_skip = u_skip; a.Zc = zt.Zc; a.lam = zt.lam; a.c = zt.c; a.variance = variance; a.winv = woodbury_inv;
  a.wv = woodbury_vector; a.init_mean = init_mean; a.init_var = init_var; a.U_mean = U_mean; a.U_var = U_var;
  a.u_var_const = u_var_const; a.X_mean = X_mean; a.X_var = X_var;
  if (wide) {
    hipLaunchKernelGGL(frw_transpose_kernel, dim3((QP * Mld + 255) / 256), dim3(256), 0, h->stream, M, QP, Mld, zt.Zc, ZT);
    hipLaunchKernelGGL(freerun_wide_kernel, dim3(n_seq), dim3(FR_THREADS), 4 * (size_t)QP * sizeof(double), h->stream, a, QP, Mld,
                       ZT, a.wv, a.winv);
  } else if (big) {
    switch (QP) {
#define FRB(qp) case qp: freerun_big_launch<qp>(h->stream, a); break;
      FRB(8) FRB(16) FRB(20) FRB(24) FRB(32) FRB(40) FRB(48) FRB(56) FRB(64)
#undef FRB
      default: return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_freerun: unsupported padded Q");
    }
  } else
  switch (QP) {
    case 8: freerun_launch<8>(h->stream, a); break;
    case 16: freerun_launch<16>(h->stream, a); break;
    case 20: freerun_launch<20>(h->stream, a); break;
    case 24: freerun_launch<24>(h->stream, a); break;
    case 32: freerun_launch<32>(h->stream, a); break;
    case 40: freerun_launch<40>(h->stream, a); break;
    case 48: freerun_launch<48>(h->stream, a); break;
    case 56: freerun_launch<56>(h->stream, a); break;
    case 64: freerun_launch<64>(h->stream, a); break;
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_freerun: unsupported padded Q");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("rgp_freerun: ") + hipGetErrorString(e));
  return RGP_OK;
}
