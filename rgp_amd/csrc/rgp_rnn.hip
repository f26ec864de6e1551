// rgp_rnn.hip -- the recurrence of the sequence-encoder back-constraint (autoreg/rnn_encoder.py:21-83: PyTorch nn.RNN /
// nn.LSTM / nn.GRU there) as a per-sequence persistent scan; GRU shown, the Elman (tanh) and LSTM cells are the other two
// instantiations of the same kernels.  The input projections gi[t] = W_ih x_t + b_ih of ALL time steps are one GEMM and
// stay with the caller; what is strictly sequential is
//     gh = W_hh h + b_hh;  r = s(gi_r + gh_r);  z = s(gi_z + gh_z);  n = tanh(gi_n + r gh_n);  h' = (1 - z) n + z h,
// 512 dependent steps per sequence at config-3 lengths, each a 3H x H matrix-vector product.  One 256-thread workgroup per
// sequence keeps W_hh in LDS (odd leading dimension) and walks the time loop; P lanes per output row, DPP lane sums.
// Forward stores h_t and the gate values (r, z, n, gh_n); the backward scan walks t = T-1..0 with the carried dL/dh in
// registers and writes dL/d(gi_t) and dL/d(gh_t), from which the weight gradients are GEMMs on the caller's side
// (rgp_amd/rnn_encoder.py).  h_0 = 0 (rnn_encoder.py h_0_type='zero').
#include <math.h>

#include "rgp_common.h"

#define RNN_THREADS 256

struct GruDev {
  rgp_rnn_scan a;
  int P, PT, ld;
};

static __device__ __forceinline__ double gsum(double v, int P) {
  if (P >= 2) v += dpp_f64<0xB1>(v);
  if (P >= 4) v += dpp_f64<0x4E>(v);
  if (P >= 8) v += dpp_f64<0x141>(v);
  return v;
}

static __device__ __forceinline__ double dot4(const double* w, int ws, const double* x, int xs, int cnt) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = 0;
  for (; j + 3 < cnt; j += 4) {
    const double w0 = w[j * ws], w1 = w[(j + 1) * ws], w2 = w[(j + 2) * ws], w3 = w[(j + 3) * ws];
    const double x0 = x[j * xs], x1 = x[(j + 1) * xs], x2 = x[(j + 2) * xs], x3 = x[(j + 3) * xs];
    s0 += w0 * x0; s1 += w1 * x1; s2 += w2 * x2; s3 += w3 * x3;
  }
  for (; j < cnt; ++j) s0 += w[j * ws] * x[j * xs];
  return (s0 + s1) + (s2 + s3);
}

// branch-free fp64 tanh (see rgp_mlp.hip): em / (em + 2), em = expm1(2|x|)
static __device__ __forceinline__ double tanh64(double x) {
  const double u = 2.0 * fmin(fabs(x), 40.0);
  const double kf = rint(u * 1.4426950408889634074);
  double r = fma(-kf, 6.93147180369123816490e-01, u);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double q = 1.0 / 87178291200.0;
  q = fma(q, r, 1.0 / 6227020800.0);
  q = fma(q, r, 1.0 / 479001600.0);
  q = fma(q, r, 1.0 / 39916800.0);
  q = fma(q, r, 1.0 / 3628800.0);
  q = fma(q, r, 1.0 / 362880.0);
  q = fma(q, r, 1.0 / 40320.0);
  q = fma(q, r, 1.0 / 5040.0);
  q = fma(q, r, 1.0 / 720.0);
  q = fma(q, r, 1.0 / 120.0);
  q = fma(q, r, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  const double p = fma(r * r, q, r);
  const double two_k = ldexp(1.0, (int)kf);
  const double em = fma(two_k, p, two_k - 1.0);
  return copysign(em / (em + 2.0), x);
}
static __device__ __forceinline__ double sigmoid64(double x) { return 0.5 + 0.5 * tanh64(0.5 * x); }

// CELL: 0 = GRU (gates r, z, n), 1 = Elman RNN with tanh, 2 = LSTM (gates i, f, g, o; PyTorch orders).  G = gate rows / H.
template <int CELL>
__global__ void __launch_bounds__(RNN_THREADS) rnn_scan_fwd_kernel(GruDev d) {
  constexpr int G = CELL == 0 ? 3 : (CELL == 1 ? 1 : 4);
  constexpr int NS = CELL == 0 ? 4 : (CELL == 1 ? 0 : 5);      // stored values per unit and step
  extern __shared__ double lds[];
  const rgp_rnn_scan& a = d.a;
  const int t = threadIdx.x, b = blockIdx.x;
  const int H = a.H, HG = G * H, T = a.T, B = a.B, ld = d.ld, P = d.P;
  double* Wl = lds;                     // [G H][ld]
  double* bl = Wl + (size_t)HG * ld;    // [G H]
  double* hl = bl + HG;                 // [H]
  double* gh = hl + H;                  // [G H]
  for (int e = t; e < HG * H; e += RNN_THREADS) Wl[(e / H) * ld + e % H] = a.W_hh[e];
  for (int e = t; e < HG; e += RNN_THREADS) bl[e] = a.b_hh[e];
  if (t < H) hl[t] = 0.0;
  const int o = t / P, part = t - o * P;
  const bool live = o < HG;
  const double* Wr = Wl + (size_t)(live ? o : 0) * ld + part;
  const int cnt = (H - part + P - 1) / P;
  double hreg = 0.0, creg = 0.0;        // thread j < H keeps h[j] (and the LSTM cell state)
  // gi of step s+1 is fetched while step s runs
  double g[G], gnext[G];
#pragma unroll
  for (int k = 0; k < G; ++k) { g[k] = 0.0; gnext[k] = 0.0; }
  if (t < H && T > 0) {
    const double* g0 = a.gi + (size_t)b * HG;
#pragma unroll
    for (int k = 0; k < G; ++k) g[k] = g0[k * H + t];
  }
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    if (t < H && s + 1 < T) {
      const double* g1 = a.gi + ((size_t)(s + 1) * B + b) * HG;
#pragma unroll
      for (int k = 0; k < G; ++k) gnext[k] = g1[k * H + t];
    }
    double acc = 0.0;
    if (live) acc = dot4(Wr, P, hl + part, P, cnt);
    acc = gsum(acc, P);
    if (live && part == 0) gh[o] = acc + bl[o];
    __syncthreads();
    if (t < H) {
      const size_t row = (size_t)s * B + b;
      double* gt = NS ? a.gates + row * NS * H : nullptr;
      if (CELL == 0) {
        const double r = sigmoid64(g[0] + gh[t]);
        const double z = sigmoid64(g[1 % G] + gh[(1 % G) * H + t]);
        const double ghn = gh[(2 % G) * H + t];
        const double n = tanh64(g[2 % G] + r * ghn);
        hreg = (1.0 - z) * n + z * hreg;
        gt[t] = r; gt[H + t] = z; gt[2 * H + t] = n; gt[3 * H + t] = ghn;
      } else if (CELL == 1) {
        hreg = tanh64(g[0] + gh[t]);
      } else {
        const double ig = sigmoid64(g[0] + gh[t]);
        const double fg = sigmoid64(g[1 % G] + gh[(1 % G) * H + t]);
        const double gg = tanh64(g[2 % G] + gh[(2 % G) * H + t]);
        const double og = sigmoid64(g[3 % G] + gh[(3 % G) * H + t]);
        creg = fg * creg + ig * gg;
        hreg = og * tanh64(creg);
        gt[t] = ig; gt[H + t] = fg; gt[2 * H + t] = gg; gt[3 * H + t] = og; gt[4 * H + t] = creg;
      }
      hl[t] = hreg;
      a.hseq[row * H + t] = hreg;
    }
#pragma unroll
    for (int k = 0; k < G; ++k) g[k] = gnext[k];
    __syncthreads();
  }
}

template <int CELL>
__global__ void __launch_bounds__(RNN_THREADS) rnn_scan_bwd_kernel(GruDev d) {
  constexpr int G = CELL == 0 ? 3 : (CELL == 1 ? 1 : 4);
  constexpr int NS = CELL == 0 ? 4 : (CELL == 1 ? 0 : 5);
  constexpr int NSR = NS ? NS : 1;
  extern __shared__ double lds[];
  const rgp_rnn_scan& a = d.a;
  const int t = threadIdx.x, b = blockIdx.x;
  const int H = a.H, HG = G * H, T = a.T, B = a.B, ld = d.ld, P = d.PT;
  double* Wl = lds;                     // [G H][ld]
  double* dgs = Wl + (size_t)HG * ld;   // [G H]  dL/d gh of the step
  double* tmp = dgs + HG;               // [H]    W_hh^T dgh
  for (int e = t; e < HG * H; e += RNN_THREADS) Wl[(e / H) * ld + e % H] = a.W_hh[e];
  const int i = t / P, part = t - i * P;
  const bool live = i < H;
  const double* Wc = Wl + (size_t)part * ld + (live ? i : 0);
  const int cnt = (HG - part + P - 1) / P;
  double dh = 0.0, dc = 0.0;            // carried dL/dh (and dL/dc) of thread j < H
  // operands of step s are fetched one step ahead: g_out, the stored gate values, h_{s} (RNN), h_{s-1} / c_{s-1}
  double go = 0.0, sv[NSR], hp = 0.0, hs = 0.0;
#pragma unroll
  for (int k = 0; k < NSR; ++k) sv[k] = 0.0;
  auto fetch = [&](int s) {
    if (t < H) {
      const size_t row = (size_t)s * B + b;
      go = a.g_out[row * H + t];
      if (NS) {
        const double* gt = a.gates + row * NS * H;
#pragma unroll
        for (int k = 0; k < NS; ++k) sv[k] = gt[k * H + t];
      }
      if (CELL == 0) hp = s > 0 ? a.hseq[((size_t)(s - 1) * B + b) * H + t] : 0.0;
      if (CELL == 1) hs = a.hseq[row * H + t];
      if (CELL == 2) hp = s > 0 ? a.gates[((size_t)(s - 1) * B + b) * NS * H + 4 * H + t] : 0.0;      // c_{s-1}
    }
  };
  if (T > 0) fetch(T - 1);
  __syncthreads();
  for (int s = T - 1; s >= 0; --s) {
    const double go_c = go, hp_c = hp, hs_c = hs;
    double sc[NSR];
#pragma unroll
    for (int k = 0; k < NSR; ++k) sc[k] = sv[k];
    if (s > 0) fetch(s - 1);
    double carry = 0.0;
    if (t < H) {
      const double dht = go_c + dh;
      const size_t row = (size_t)s * B + b;
      double* gi = a.dgi + row * HG;
      double* gh = a.dgh + row * HG;
      if (CELL == 0) {
        const double r_c = sc[0], z_c = sc[1 % NSR], n_c = sc[2 % NSR], ghn_c = sc[3 % NSR];
        const double dn = dht * (1.0 - z_c);
        const double dz = dht * (hp_c - n_c);
        const double dnp = dn * (1.0 - n_c * n_c);
        const double dr = dnp * ghn_c;
        const double dzp = dz * z_c * (1.0 - z_c);
        const double drp = dr * r_c * (1.0 - r_c);
        carry = dht * z_c;
        gi[t] = drp; gi[(1 % G) * H + t] = dzp; gi[(2 % G) * H + t] = dnp;
        const double dghn = dnp * r_c;
        gh[t] = drp; gh[(1 % G) * H + t] = dzp; gh[(2 % G) * H + t] = dghn;
        dgs[t] = drp; dgs[(1 % G) * H + t] = dzp; dgs[(2 % G) * H + t] = dghn;
      } else if (CELL == 1) {
        const double dp = dht * (1.0 - hs_c * hs_c);
        gi[t] = dp; gh[t] = dp; dgs[t] = dp;
      } else {
        const double ig = sc[0], fg = sc[1 % NSR], gg = sc[2 % NSR], og = sc[3 % NSR], cs = sc[4 % NSR];
        const double tc = tanh64(cs);
        const double dcc = dc + dht * og * (1.0 - tc * tc);
        const double dip = dcc * gg * ig * (1.0 - ig);
        const double dfp = dcc * hp_c * fg * (1.0 - fg);           // hp_c holds c_{s-1}
        const double dgp = dcc * ig * (1.0 - gg * gg);
        const double dop = dht * tc * og * (1.0 - og);
        dc = dcc * fg;
        gi[t] = dip; gi[(1 % G) * H + t] = dfp; gi[(2 % G) * H + t] = dgp; gi[(3 % G) * H + t] = dop;
        gh[t] = dip; gh[(1 % G) * H + t] = dfp; gh[(2 % G) * H + t] = dgp; gh[(3 % G) * H + t] = dop;
        dgs[t] = dip; dgs[(1 % G) * H + t] = dfp; dgs[(2 % G) * H + t] = dgp; dgs[(3 % G) * H + t] = dop;
      }
    }
    __syncthreads();
    double acc = 0.0;
    if (live) acc = dot4(Wc, P * ld, dgs + part, P, cnt);
    acc = gsum(acc, P);
    if (live && part == 0) tmp[i] = acc;
    __syncthreads();
    if (t < H) dh = carry + tmp[t];
  }
}

static rgp_status rnn_plan(rgp_handle h, const rgp_rnn_scan* a, GruDev& d, size_t& lds) {
  if (!a || a->T < 1 || a->B < 1 || a->H < 1) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan: empty problem");
  if (a->cell < 0 || a->cell > 2) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan: cell must be RGP_CELL_GRU, RGP_CELL_RNN_TANH or RGP_CELL_LSTM");
  const int G = a->cell == 0 ? 3 : (a->cell == 1 ? 1 : 4);
  const int H = a->H, HG = G * H;
  if (HG > RNN_THREADS) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan: more than 256 gate rows (hidden width too large)");
  if (!a->W_hh || !a->b_hh || !a->hseq || (a->cell != 1 && !a->gates)) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan: null buffer");
  d.a = *a;
  int P = 1;
  while (P < 8 && HG * P * 2 <= RNN_THREADS && P * 2 <= H) P *= 2;
  int PT = 1;
  while (PT < 8 && H * PT * 2 <= RNN_THREADS && PT * 2 <= HG) PT *= 2;
  d.P = P; d.PT = PT; d.ld = H | 1;
  lds = ((size_t)HG * d.ld + 2 * (size_t)HG + 2 * (size_t)H) * sizeof(double);
  if (lds > 160 * 1024) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan: W_hh does not fit in LDS");
  return RGP_OK;
}

template <int CELL>
static hipError_t rnn_launch(bool fwd, const GruDev& d, size_t lds, hipStream_t s) {
  hipError_t e;
  if (fwd) {
    e = hipFuncSetAttribute((const void*)rnn_scan_fwd_kernel<CELL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL(rnn_scan_fwd_kernel<CELL>, dim3(d.a.B), dim3(RNN_THREADS), lds, s, d);
  } else {
    e = hipFuncSetAttribute((const void*)rnn_scan_bwd_kernel<CELL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL(rnn_scan_bwd_kernel<CELL>, dim3(d.a.B), dim3(RNN_THREADS), lds, s, d);
  }
  return e == hipSuccess ? hipGetLastError() : e;
}

static rgp_status rnn_run(rgp_handle h, const rgp_rnn_scan* a, bool fwd) {
  if (!h) return RGP_ERR_BAD_ARG;
  GruDev d{};
  size_t lds = 0;
  rgp_status st = rnn_plan(h, a, d, lds);
  if (st) return st;
  if (fwd && !a->gi) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan forward: gi is required");
  if (!fwd && (!a->g_out || !a->dgi || !a->dgh)) return rgp_fail(h, RGP_ERR_BAD_ARG, "rnn scan backward: g_out, dgi and dgh are required");
  hipError_t e = a->cell == 0 ? rnn_launch<0>(fwd, d, lds, h->stream)
               : a->cell == 1 ? rnn_launch<1>(fwd, d, lds, h->stream) : rnn_launch<2>(fwd, d, lds, h->stream);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("rnn scan: ") + hipGetErrorString(e));
  return RGP_OK;
}

extern "C" rgp_status rgp_rnn_scan_forward(rgp_handle h, const rgp_rnn_scan* a) { return rnn_run(h, a, true); }
extern "C" rgp_status rgp_rnn_scan_backward(rgp_handle h, const rgp_rnn_scan* a) { return rnn_run(h, a, false); }
