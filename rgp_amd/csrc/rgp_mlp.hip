// rgp_mlp.hip -- the MLP back-constraint recursion of a hidden layer as a per-sequence persistent scan
// (Layer._encoder_freerun, autoreg/layers.py:623-666, and the hand-written reverse loop of Layer._encoder_update_gradient,
// layers.py:668-756; network autoreg/mlp.py:13-175: tanh hidden layers, linear output).
//
//   mean[X_win + n] = MLP([mean[n .. n+X_win) | up[u_off+n .. u_off+n+U_win)]),   n = 0..N-1,  u_off = up_len - N - U_win + 1
//
// The recursion is strictly sequential in n (4076 dependent network evaluations per sequence at config-4 lengths), so one
// 256-thread workgroup per sequence keeps the weights of all layers in LDS and walks the time loop; sequences run in
// parallel on different CUs.  A matrix-vector product uses P lanes per output (P = 2^k <= 8, summed with wave shuffles), so
// that every layer has ~160-256 active lanes and a step costs a few hundred cycles instead of one kernel launch per layer.
//
// Forward: writes the means and the hidden activations (the backward needs them).
// Backward: reverse scan with the reference's bookkeeping -- the gradient of row X_win+n is complete once the steps n' > n
// have added their window contributions (ring of X_win+1 rows in LDS), the gradient of the signal above is complete for
// row u_off+n+U_win-1 after step n (ring of U_win rows).  It writes delta_k[n] = dL/d(pre-activation of layer k at step n);
// the weight gradients are the GEMMs sum_n delta_k[n] in_k[n]^T, left to rocBLAS on the host side (rgp_amd/mlp_encoder.py).
#include <math.h>

#include "rgp_common.h"

#define MLP_THREADS 256

struct MlpDev {
  rgp_mlp_scan a;
  int P[RGP_MLP_MAX_LAYERS], PT[RGP_MLP_MAX_LAYERS];      // lanes per output: forward / transposed product
  int ld[RGP_MLP_MAX_LAYERS], woff[RGP_MLP_MAX_LAYERS], boff[RGP_MLP_MAX_LAYERS];
  int vmax;                                                // longest activation vector
  int ring_off, uring_off, v0_off, v1_off;                 // LDS offsets (doubles)
};

// sum over aligned groups of P = 1, 2, 4 or 8 adjacent lanes (P is uniform over the workgroup): DPP lane exchanges instead of
// __shfl_xor, which compiles to ds_bpermute -- an LDS round trip per exchange on the critical path of every layer of every step
static __device__ __forceinline__ double group_sum(double v, int P) {
  if (P >= 2) v += dpp_f64<0xB1>(v);            // quad_perm [1,0,3,2]
  if (P >= 4) v += dpp_f64<0x4E>(v);            // quad_perm [2,3,0,1]
  if (P >= 8) v += dpp_f64<0x141>(v);           // row_half_mirror
  return v;
}

// sum_j w[j * ws] * x[j * xs], j = 0..cnt-1: four independent loads in flight per trip (a plain loop waits one LDS round
// trip per term, which is what a step of the recursion costs otherwise)
static __device__ __forceinline__ double dot_lds(const double* w, int ws, const double* x, int xs, int cnt) {
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

// tanh(x) = em / (em + 2), em = expm1(2|x|) = 2^k (p + 1) - 1 with 2|x| = k ln2 + r, p = expm1(r) by its Taylor series up to
// r^14 (|r| <= 0.347: remainder 4e-18).  Branch free and ~50 dependent instructions; the library tanh/exp pair cost 0.86 us per
// call on the one active lane group here (both sides of its small-argument branch execute), more than the matrix-vector
// products of the step.  Relative error ~3e-16 including small |x| (no 1 - 2/(e+1) cancellation).
static __device__ __forceinline__ double tanh_f64(double x) {
  const double u = 2.0 * fmin(fabs(x), 40.0);
  const double kf = rint(u * 1.4426950408889634074);
  double r = fma(-kf, 6.93147180369123816490e-01, u);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double q = 1.0 / 87178291200.0;                   // 1/14!
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
  const double p = fma(r * r, q, r);                 // expm1(r)
  const double two_k = ldexp(1.0, (int)kf);
  const double em = fma(two_k, p, two_k - 1.0);
  return copysign(em / (em + 2.0), x);
}

static __device__ void load_weights(const MlpDev& d, double* lds) {
  for (int k = 0; k < d.a.n_layers; ++k) {
    const int out = d.a.dims[k + 1], in = d.a.dims[k];
    const double* W = d.a.W[k];
    for (int e = threadIdx.x; e < out * in; e += MLP_THREADS) lds[d.woff[k] + (e / in) * d.ld[k] + e % in] = W[e];
    for (int e = threadIdx.x; e < out; e += MLP_THREADS) lds[d.boff[k] + e] = d.a.b[k][e];
  }
}

// thread t owns element t of the input vector (dims[0] <= 256): which ring and which row offset / column it comes from
struct InputSlot { bool live, own; int j, dd; };
static __device__ __forceinline__ InputSlot input_slot(const rgp_mlp_scan& a, int t) {
  InputSlot s;
  const int XQ = a.X_win * a.X_dim;
  s.live = t < a.dims[0];
  s.own = t < XQ;
  if (s.own) { s.j = t / a.X_dim; s.dd = t - s.j * a.X_dim; }
  else { const int i2 = s.live ? t - XQ : 0; const int ud = a.U_dim > 0 ? a.U_dim : 1; s.j = i2 / ud; s.dd = i2 - s.j * ud; }
  return s;
}

// NL = number of linear layers: the layer loops are unrolled so that the per-layer constants (widths, LDS offsets, lane
// split, buffer pointers) are loop-invariant scalars instead of kernel-argument loads inside the time loop
template <int NL>
__global__ void __launch_bounds__(MLP_THREADS) mlp_scan_fwd_kernel(MlpDev d) {
  extern __shared__ double lds[];
  const rgp_mlp_scan& a = d.a;
  const int t = threadIdx.x, seq = blockIdx.x;
  const int Xw = a.X_win, Xd = a.X_dim, Uw = a.U_win, Ud = a.U_dim, N = a.N;
  constexpr int nl = NL;
  const int R = Xw + 1, XQ = Xw * Xd;
  double* ring = lds + d.ring_off;        // [R][Xd]   latent rows n .. n+X_win (row r in slot r % R)
  double* uring = lds + d.uring_off;      // [Uw][Ud]  rows of the signal above (row r in slot r % Uw)
  double* v0 = lds + d.v0_off;
  double* v1 = lds + d.v1_off;
  const int64_t u_off = (int64_t)a.up_len - N - Uw + 1;
  const double* up = a.up ? a.up + (size_t)seq * a.up_len * Ud : nullptr;
  double* mean = a.mean + (size_t)seq * (Xw + N) * Xd;

  load_weights(d, lds);
  for (int e = t; e < XQ; e += MLP_THREADS) {
    const double v = a.init_mean[(size_t)seq * XQ + e];
    ring[e] = v;                         // rows 0..Xw-1 sit in slots 0..Xw-1
    mean[e] = v;
  }
  for (int e = t; e < Uw * Ud; e += MLP_THREADS) {
    const int64_t row = u_off + e / Ud;
    uring[(row % Uw) * Ud + e % Ud] = up[row * Ud + e % Ud];
  }
  const InputSlot in0 = input_slot(a, t);
  // slot of this thread's input element at step n = 0, advanced by one row per step
  int slot = in0.own ? in0.j % R : (Uw > 0 ? (int)((u_off + in0.j) % Uw) : 0);
  const int nslots = in0.own ? R : (Uw > 0 ? Uw : 1);
  int new_slot = Xw % R;                  // where row X_win + n goes
  int u_new = Uw > 0 ? (int)(u_off % Uw) : 0;     // slot of row u_off + n (replaced by row u_off + n + U_win)
  __syncthreads();

  // rows of the signal above are fetched TWO steps ahead: memory operations retire in order, so a load consumed in the step
  // that issued it would also wait for the acknowledgement of the previous step's activation stores
  double unext = 0.0, unext2 = 0.0;
  if (Uw > 0 && t < Ud && 1 < N) unext = up[(u_off + Uw) * Ud + t];
  for (int n = 0; n < N; ++n) {
    const bool ufetch = Uw > 0 && t < Ud && n + 1 < N;
    if (Uw > 0 && t < Ud && n + 2 < N) unext2 = up[(u_off + n + 1 + Uw) * Ud + t];
    // input vector: own window, then the window of the signal above (layers.py:661-664)
    if (in0.live) v0[t] = in0.own ? ring[slot * Xd + in0.dd] : uring[slot * Ud + in0.dd];
    slot = slot + 1 == nslots ? 0 : slot + 1;
    __syncthreads();
    double* x = v0;
    double* y = v1;
#pragma unroll
    for (int k = 0; k < nl; ++k) {
      const int out = a.dims[k + 1], in = a.dims[k], P = d.P[k];
      const int o = t / P, part = t - o * P;
      const bool live = o < out;
      double s = 0.0;
      if (live) s = dot_lds(lds + d.woff[k] + o * d.ld[k] + part, P, x + part, P, (in - part + P - 1) / P);
      s = group_sum(s, P);
      if (live && part == 0) {
        s += lds[d.boff[k] + o];
        if (k < nl - 1) {
          s = tanh_f64(s);
          y[o] = s;
          a.act[k][((size_t)seq * N + n) * out + o] = s;
        } else {
          ring[new_slot * Xd + o] = s;                                // replaces row n-1
          mean[(size_t)(Xw + n) * Xd + o] = s;
        }
      }
      if (k == nl - 1 && ufetch) uring[u_new * Ud + t] = unext;      // row u_off+n is not needed again
      __syncthreads();
      double* tmp = x; x = y; y = tmp;
    }
    new_slot = new_slot + 1 == R ? 0 : new_slot + 1;
    if (Uw > 0) u_new = u_new + 1 == Uw ? 0 : u_new + 1;
    unext = unext2;
  }
}

template <int NL>
__global__ void __launch_bounds__(MLP_THREADS) mlp_scan_bwd_kernel(MlpDev d) {
  extern __shared__ double lds[];
  const rgp_mlp_scan& a = d.a;
  const int t = threadIdx.x, seq = blockIdx.x;
  const int Xw = a.X_win, Xd = a.X_dim, Uw = a.U_win, Ud = a.U_dim, N = a.N;
  constexpr int nl = NL;
  const int R = Xw + 1, XQ = Xw * Xd;
  double* gring = lds + d.ring_off;       // gradient accumulators of latent rows n .. n+X_win
  double* guring = lds + d.uring_off;     // gradient accumulators of rows u_off+n .. u_off+n+U_win-1 of the signal above
  double* v0 = lds + d.v0_off;
  double* v1 = lds + d.v1_off;
  const int64_t u_off = (int64_t)a.up_len - N - Uw + 1;
  const double* g_mean = a.g_mean + (size_t)seq * (Xw + N) * Xd;
  double* g_up = a.g_up ? a.g_up + (size_t)seq * a.up_len * Ud : nullptr;

  load_weights(d, lds);
  for (int e = t; e < R * Xd; e += MLP_THREADS) gring[e] = 0.0;
  for (int e = t; e < Uw * Ud; e += MLP_THREADS) guring[e] = 0.0;
  const InputSlot in0 = input_slot(a, t);
  // slots at step n = N-1, moved back by one row per step
  int slot = in0.own ? (N - 1 + in0.j) % R : (Uw > 0 ? (int)((u_off + N - 1 + in0.j) % Uw) : 0);
  const int nslots = in0.own ? R : (Uw > 0 ? Uw : 1);
  int out_slot = (Xw + N - 1) % R;                                   // slot of row X_win + n
  int utop_slot = Uw > 0 ? (int)((u_off + N - 1 + Uw - 1) % Uw) : 0;  // slot of row u_off + n + U_win - 1
  __syncthreads();

  // operands of step n held in registers, fetched one step ahead: g_mean[X_win+n][t] and, for the tanh derivative of layer
  // k-1, the activation of the unit this thread produces in the transposed product of layer k (unit t / PT[k])
  double gm_next = 0.0, h_next[RGP_MLP_MAX_LAYERS - 1];
#pragma unroll
  for (int k = 0; k < RGP_MLP_MAX_LAYERS - 1; ++k) h_next[k] = 0.0;
  auto fetch = [&](int n) {
    if (t < Xd) gm_next = g_mean[(size_t)(Xw + n) * Xd + t];
#pragma unroll
    for (int k = 0; k < RGP_MLP_MAX_LAYERS - 1; ++k)
      if (k < nl - 1) {
        const int u = t / d.PT[k + 1];
        if (u < a.dims[k + 1]) h_next[k] = a.act[k][((size_t)seq * N + n) * a.dims[k + 1] + u];
      }
  };
  if (N > 0) fetch(N - 1);

  for (int n = N - 1; n >= 0; --n) {
    const double gm = gm_next;
    double h[RGP_MLP_MAX_LAYERS - 1];
#pragma unroll
    for (int k = 0; k < RGP_MLP_MAX_LAYERS - 1; ++k) h[k] = h_next[k];
    if (n > 0) fetch(n - 1);
    // dL/d mean[X_win+n]: the direct term plus what the later steps put into the ring (layers.py:705-712); its slot becomes
    // the accumulator of row n-1
    if (t < Xd) {
      const double g = gring[out_slot * Xd + t] + gm;
      gring[out_slot * Xd + t] = 0.0;
      v0[t] = g;
      a.delta[nl - 1][((size_t)seq * N + n) * Xd + t] = g;
    }
    __syncthreads();
    double* dl = v0;
    double* gi = v1;
#pragma unroll
    for (int k = nl - 1; k >= 0; --k) {
      // gi[i] = sum_o W_k[o][i] dl[o], then through the tanh of layer k-1
      const int out = a.dims[k + 1], in = a.dims[k], P = d.PT[k];
      const int i = t / P, part = t - i * P;
      const bool live = i < in;
      double s = 0.0;
      if (live) s = dot_lds(lds + d.woff[k] + part * d.ld[k] + i, P * d.ld[k], dl + part, P, (out - part + P - 1) / P);
      s = group_sum(s, P);
      if (k > 0) {
        double hk = 0.0;
#pragma unroll
        for (int kk = 0; kk < RGP_MLP_MAX_LAYERS - 1; ++kk) hk = (kk == k - 1) ? h[kk] : hk;
        s *= 1.0 - hk * hk;
        if (live && part == 0) a.delta[k - 1][((size_t)seq * N + n) * in + i] = s;
      }
      if (live && part == 0) gi[i] = s;
      __syncthreads();
      double* tmp = dl; dl = gi; gi = tmp;
    }
    // dl = gradient of the input vector: overlap-add onto the window rows (element t -> row n + j)
    if (in0.live) {
      if (in0.own) gring[slot * Xd + in0.dd] += dl[t];
      else guring[slot * Ud + in0.dd] += dl[t];
    }
    slot = slot == 0 ? nslots - 1 : slot - 1;
    out_slot = out_slot == 0 ? R - 1 : out_slot - 1;
    __syncthreads();
    if (Uw > 0 && t < Ud) {                                   // row u_off+n+U_win-1 is complete; its slot serves row u_off+n-1 next
      if (g_up) g_up[(u_off + n + Uw - 1) * Ud + t] = guring[utop_slot * Ud + t];
      guring[utop_slot * Ud + t] = 0.0;
    }
    if (Uw > 0) utop_slot = utop_slot == 0 ? Uw - 1 : utop_slot - 1;
    // (the next step's first phase touches gring/v0 only; its barrier orders the guring reset before the next overlap-add)
  }
  __syncthreads();
  // the first X_win rows (init_Xs) and the first U_win-1 rows of the window range of the signal above
  for (int e = t; e < XQ; e += MLP_THREADS) a.g_init[(size_t)seq * XQ + e] = gring[e] + g_mean[e];
  if (g_up)
    for (int e = t; e < (Uw - 1) * Ud; e += MLP_THREADS) {
      const int64_t row = u_off + e / Ud;
      g_up[row * Ud + e % Ud] = guring[(row % Uw) * Ud + e % Ud];
    }
}

static rgp_status mlp_plan(rgp_handle h, const rgp_mlp_scan* a, MlpDev& d, size_t& lds_bytes) {
  if (!a || a->n_layers < 1 || a->n_layers > RGP_MLP_MAX_LAYERS) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: 1..4 linear layers");
  if (a->X_win < 1 || a->N < 0 || a->n_seq < 1) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: needs X_win >= 1 (layers.py:403)");
  if (a->dims[0] != a->X_win * a->X_dim + a->U_win * a->U_dim || a->dims[a->n_layers] != a->X_dim)
    return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: dims[0] must be the window width and dims[n_layers] the latent width");
  if (a->U_win > 0 && (!a->up || a->up_len < a->N + a->U_win - 1)) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: signal above too short");
  if (a->U_dim > MLP_THREADS || a->X_dim > MLP_THREADS) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: row width > 256");
  d.a = *a;
  int off = 0, vmax = 0;
  for (int k = 0; k <= a->n_layers; ++k) {
    if (a->dims[k] < 1 || a->dims[k] > MLP_THREADS) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: layer width must be 1..256");
    if (a->dims[k] > vmax) vmax = a->dims[k];
  }
  for (int k = 0; k < a->n_layers; ++k) {
    const int out = a->dims[k + 1], in = a->dims[k];
    int P = 1;
    while (P < 8 && out * P * 2 <= MLP_THREADS && P * 2 <= in) P *= 2;
    int PT = 1;
    while (PT < 8 && in * PT * 2 <= MLP_THREADS && PT * 2 <= out) PT *= 2;
    d.P[k] = P; d.PT[k] = PT;
    d.ld[k] = in | 1;                                        // odd leading dimension: both access patterns stay conflict free
    d.woff[k] = off; off += out * d.ld[k];
    d.boff[k] = off; off += out;
  }
  d.vmax = vmax;
  d.ring_off = off; off += (a->X_win + 1) * a->X_dim;
  d.uring_off = off; off += (a->U_win > 0 ? a->U_win * a->U_dim : 1);
  d.v0_off = off; off += vmax;
  d.v1_off = off; off += vmax;
  lds_bytes = (size_t)off * sizeof(double);
  if (lds_bytes > 160 * 1024) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan: the network does not fit in LDS (160 KB)");
  return RGP_OK;
}

extern "C" rgp_status rgp_mlp_scan_forward(rgp_handle h, const rgp_mlp_scan* a) {
  if (!h) return RGP_ERR_BAD_ARG;
  MlpDev d{};
  size_t lds = 0;
  rgp_status st = mlp_plan(h, a, d, lds);
  if (st) return st;
  if (!a->init_mean || !a->mean) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan forward: init_mean and mean are required");
  for (int k = 0; k < a->n_layers; ++k)
    if (!a->W[k] || !a->b[k] || (k < a->n_layers - 1 && !a->act[k])) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan forward: null weight/activation buffer");
  hipError_t e = hipSuccess;
#define MLP_FWD(NL)                                                                                                     \
  case NL:                                                                                                              \
    e = hipFuncSetAttribute((const void*)mlp_scan_fwd_kernel<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e == hipSuccess) hipLaunchKernelGGL(mlp_scan_fwd_kernel<NL>, dim3(a->n_seq), dim3(MLP_THREADS), lds, h->stream, d); \
    break;
  switch (a->n_layers) { MLP_FWD(1) MLP_FWD(2) MLP_FWD(3) MLP_FWD(4) }
#undef MLP_FWD
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("mlp scan LDS attr: ") + hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("rgp_mlp_scan_forward: ") + hipGetErrorString(e));
  return RGP_OK;
}

extern "C" rgp_status rgp_mlp_scan_backward(rgp_handle h, const rgp_mlp_scan* a) {
  if (!h) return RGP_ERR_BAD_ARG;
  MlpDev d{};
  size_t lds = 0;
  rgp_status st = mlp_plan(h, a, d, lds);
  if (st) return st;
  if (!a->g_mean || !a->g_init) return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan backward: g_mean and g_init are required");
  for (int k = 0; k < a->n_layers; ++k)
    if (!a->W[k] || !a->b[k] || !a->delta[k] || (k < a->n_layers - 1 && !a->act[k]))
      return rgp_fail(h, RGP_ERR_BAD_ARG, "mlp scan backward: null weight/activation/delta buffer");
  if (a->g_up) {
    hipError_t e0 = hipMemsetAsync(a->g_up, 0, (size_t)a->n_seq * a->up_len * a->U_dim * sizeof(double), h->stream);
    if (e0 != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("mlp scan memset: ") + hipGetErrorString(e0));
  }
  hipError_t e = hipSuccess;
#define MLP_BWD(NL)                                                                                                     \
  case NL:                                                                                                              \
    e = hipFuncSetAttribute((const void*)mlp_scan_bwd_kernel<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e == hipSuccess) hipLaunchKernelGGL(mlp_scan_bwd_kernel<NL>, dim3(a->n_seq), dim3(MLP_THREADS), lds, h->stream, d); \
    break;
  switch (a->n_layers) { MLP_BWD(1) MLP_BWD(2) MLP_BWD(3) MLP_BWD(4) }
#undef MLP_BWD
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("mlp scan LDS attr: ") + hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("rgp_mlp_scan_backward: ") + hipGetErrorString(e));
  return RGP_OK;
}
