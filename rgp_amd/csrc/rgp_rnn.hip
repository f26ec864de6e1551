// rgp_rnn.hip -- the GRU recurrence of the sequence-encoder back-constraint (autoreg/rnn_encoder.py:21-83, PyTorch nn.GRU
// there) as a per-sequence persistent scan.  The input projections gi[t] = W_ih x_t + b_ih of ALL time steps are one GEMM and
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
  rgp_gru_scan a;
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

__global__ void __launch_bounds__(RNN_THREADS) gru_scan_fwd_kernel(GruDev d) {
  extern __shared__ double lds[];
  const rgp_gru_scan& a = d.a;
  const int t = threadIdx.x, b = blockIdx.x;
  const int H = a.H, H3 = 3 * H, T = a.T, B = a.B, ld = d.ld, P = d.P;
  double* Wl = lds;                     // [3H][ld]
  double* bl = Wl + (size_t)H3 * ld;    // [3H]
  double* hl = bl + H3;                 // [H]
  double* gh = hl + H;                  // [3H]
  for (int e = t; e < H3 * H; e += RNN_THREADS) Wl[(e / H) * ld + e % H] = a.W_hh[e];
  for (int e = t; e < H3; e += RNN_THREADS) bl[e] = a.b_hh[e];
  if (t < H) hl[t] = 0.0;
  const int o = t / P, part = t - o * P;
  const bool live = o < H3;
  const double* Wr = Wl + (size_t)(live ? o : 0) * ld + part;
  const int cnt = (H - part + P - 1) / P;
  double hreg = 0.0;                    // thread j < H keeps h[j]
  // gi of step t+1 is fetched while step t runs
  double gr = 0.0, gz = 0.0, gn = 0.0;
  if (t < H && T > 0) {
    const double* g0 = a.gi + (size_t)b * H3;
    gr = g0[t]; gz = g0[H + t]; gn = g0[2 * H + t];
  }
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    double nr = 0.0, nz = 0.0, nn = 0.0;
    if (t < H && s + 1 < T) {
      const double* g1 = a.gi + ((size_t)(s + 1) * B + b) * H3;
      nr = g1[t]; nz = g1[H + t]; nn = g1[2 * H + t];
    }
    double acc = 0.0;
    if (live) acc = dot4(Wr, P, hl + part, P, cnt);
    acc = gsum(acc, P);
    if (live && part == 0) gh[o] = acc + bl[o];
    __syncthreads();
    if (t < H) {
      const double r = sigmoid64(gr + gh[t]);
      const double z = sigmoid64(gz + gh[H + t]);
      const double ghn = gh[2 * H + t];
      const double n = tanh64(gn + r * ghn);
      hreg = (1.0 - z) * n + z * hreg;
      hl[t] = hreg;
      const size_t row = (size_t)s * B + b;
      a.hseq[row * H + t] = hreg;
      double* gt = a.gates + row * 4 * H;
      gt[t] = r; gt[H + t] = z; gt[2 * H + t] = n; gt[3 * H + t] = ghn;
    }
    gr = nr; gz = nz; gn = nn;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(RNN_THREADS) gru_scan_bwd_kernel(GruDev d) {
  extern __shared__ double lds[];
  const rgp_gru_scan& a = d.a;
  const int t = threadIdx.x, b = blockIdx.x;
  const int H = a.H, H3 = 3 * H, T = a.T, B = a.B, ld = d.ld, P = d.PT;
  double* Wl = lds;                     // [3H][ld]
  double* dgs = Wl + (size_t)H3 * ld;   // [3H]  dL/d gh of the step
  double* tmp = dgs + H3;               // [H]   W_hh^T dgh
  for (int e = t; e < H3 * H; e += RNN_THREADS) Wl[(e / H) * ld + e % H] = a.W_hh[e];
  const int i = t / P, part = t - i * P;
  const bool live = i < H;
  const double* Wc = Wl + (size_t)part * ld + (live ? i : 0);
  const int cnt = (H3 - part + P - 1) / P;
  double dh = 0.0;                      // carried dL/dh_{s} for thread j < H
  // operands of step s are fetched one step ahead
  double go = 0.0, r = 0.0, z = 0.0, n = 0.0, ghn = 0.0, hp = 0.0;
  auto fetch = [&](int s) {
    if (t < H) {
      const size_t row = (size_t)s * B + b;
      go = a.g_out[row * H + t];
      const double* gt = a.gates + row * 4 * H;
      r = gt[t]; z = gt[H + t]; n = gt[2 * H + t]; ghn = gt[3 * H + t];
      hp = s > 0 ? a.hseq[((size_t)(s - 1) * B + b) * H + t] : 0.0;
    }
  };
  if (T > 0) fetch(T - 1);
  __syncthreads();
  for (int s = T - 1; s >= 0; --s) {
    const double go_c = go, r_c = r, z_c = z, n_c = n, ghn_c = ghn, hp_c = hp;
    if (s > 0) fetch(s - 1);
    double carry = 0.0;
    if (t < H) {
      const double dht = go_c + dh;
      const double dn = dht * (1.0 - z_c);
      const double dz = dht * (hp_c - n_c);
      const double dnp = dn * (1.0 - n_c * n_c);
      const double dr = dnp * ghn_c;
      const double dzp = dz * z_c * (1.0 - z_c);
      const double drp = dr * r_c * (1.0 - r_c);
      carry = dht * z_c;
      const size_t row = (size_t)s * B + b;
      double* gi = a.dgi + row * H3;
      double* gh = a.dgh + row * H3;
      gi[t] = drp; gi[H + t] = dzp; gi[2 * H + t] = dnp;
      const double dghn = dnp * r_c;
      gh[t] = drp; gh[H + t] = dzp; gh[2 * H + t] = dghn;
      dgs[t] = drp; dgs[H + t] = dzp; dgs[2 * H + t] = dghn;
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

static rgp_status gru_plan(rgp_handle h, const rgp_gru_scan* a, GruDev& d, size_t& lds) {
  if (!a || a->T < 1 || a->B < 1 || a->H < 1) return rgp_fail(h, RGP_ERR_BAD_ARG, "gru scan: empty problem");
  if (3 * a->H > RNN_THREADS) return rgp_fail(h, RGP_ERR_BAD_ARG, "gru scan: hidden width > 85 (3H lanes per step) is not supported");
  if (!a->W_hh || !a->b_hh || !a->hseq || !a->gates) return rgp_fail(h, RGP_ERR_BAD_ARG, "gru scan: null buffer");
  d.a = *a;
  const int H = a->H, H3 = 3 * H;
  int P = 1;
  while (P < 8 && H3 * P * 2 <= RNN_THREADS && P * 2 <= H) P *= 2;
  int PT = 1;
  while (PT < 8 && H * PT * 2 <= RNN_THREADS && PT * 2 <= H3) PT *= 2;
  d.P = P; d.PT = PT; d.ld = H | 1;
  lds = ((size_t)H3 * d.ld + 2 * (size_t)H3 + 2 * (size_t)H) * sizeof(double);
  if (lds > 160 * 1024) return rgp_fail(h, RGP_ERR_BAD_ARG, "gru scan: W_hh does not fit in LDS");
  return RGP_OK;
}

extern "C" rgp_status rgp_gru_scan_forward(rgp_handle h, const rgp_gru_scan* a) {
  if (!h) return RGP_ERR_BAD_ARG;
  GruDev d{};
  size_t lds = 0;
  rgp_status st = gru_plan(h, a, d, lds);
  if (st) return st;
  if (!a->gi) return rgp_fail(h, RGP_ERR_BAD_ARG, "gru scan forward: gi is required");
  hipError_t e = hipFuncSetAttribute((const void*)gru_scan_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("gru scan LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(gru_scan_fwd_kernel, dim3(a->B), dim3(RNN_THREADS), lds, h->stream, d);
  e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("rgp_gru_scan_forward: ") + hipGetErrorString(e));
  return RGP_OK;
}

extern "C" rgp_status rgp_gru_scan_backward(rgp_handle h, const rgp_gru_scan* a) {
  if (!h) return RGP_ERR_BAD_ARG;
  GruDev d{};
  size_t lds = 0;
  rgp_status st = gru_plan(h, a, d, lds);
  if (st) return st;
  if (!a->g_out || !a->dgi || !a->dgh) return rgp_fail(h, RGP_ERR_BAD_ARG, "gru scan backward: g_out, dgi and dgh are required");
  hipError_t e = hipFuncSetAttribute((const void*)gru_scan_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("gru scan LDS attr: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(gru_scan_bwd_kernel, dim3(a->B), dim3(RNN_THREADS), lds, h->stream, d);
  e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("rgp_gru_scan_backward: ") + hipGetErrorString(e));
  return RGP_OK;
}
