// rgp_psi.h -- launchers of the psi-statistic kernels (rgp_psi.hip). Internal to librgp_hip.so.
#pragma once
#include "rgp_common.h"

#define RGP_PSI2_TEAM 128          // threads per workgroup of the psi2 kernel (one inducing row per thread)
#define RGP_MAX_M 128
#define RGP_MAX_QP 64

// exp(x) for the psi2 sweep (x <= ~700, any negative x including -inf): 2^k * T[j] * (1 + q(r)) with
// n = rint(x * 256/ln2) = 256 k + j, r = x - n ln2/256 (|r| <= ln2/512) and q the degree-4 Taylor polynomial of expm1 (next
// term r^5/120 < 4e-17).  T[j] = 2^(j/256) is a 256-entry LDS table (2 KB).  14 VALU instructions and one LDS read instead of
// the ~36 (27 fp64 + 9 register copies of polynomial constants) the library exp compiled to in the sweep; error <= 1 ulp
// (the rounding of T[j] plus one fused multiply-add).  The Horner steps are written as v_fma_f64 with three register
// sources: the compiler's own choice (copy the constant, then v_fmac) doubled their issue cost.
#define PSI2_EXP_TAB 256
static __device__ __forceinline__ double fma_rrr(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max(x, -800) that keeps a NaN: one unsigned minimum on the high word (x < -800 exactly when that word, read as unsigned,
// exceeds the high word of -800.0).  fmax would return -800 for a NaN and the exponential would turn a bad input into 0.
// The arithmetic NaNs of the device and numpy's nan have the sign bit clear, which is what the comparison relies on.
static __device__ __forceinline__ double clamp_exp_arg(double x) {
  const unsigned hi = (unsigned)__double2hiint(x);
  return __hiloint2double((int)(hi < 0xC0890000u ? hi : 0xC0890000u), __double2loint(x));
}
static __device__ __forceinline__ double exp_tab(double x, const double* __restrict__ T) {
  x = clamp_exp_arg(x);                                             // also maps -inf; the result underflows to 0
  const double t = fma(x, 0x1.71547652b82fep+8, 0x1.8p52);          // 256/ln2; integer n in the low mantissa bits
  const int n = __double2loint(t);
  const double nf = t - 0x1.8p52;
  double r = fma(nf, -0x1.62e42fee00000p-9, x);                     // ln2/256 split: the high part has 32 significant bits
  r = fma(nf, -0x1.a39ef35793c76p-41, r);
  const double tj = T[n & (PSI2_EXP_TAB - 1)];
  double q = fma_rrr(r, 0x1.5555555555555p-5, 0x1.5555555555555p-3);       // 1/24, 1/6
  q = fma_rrr(q, r, 0.5);
  q = fma_rrr(q, r, 1.0);
  q = q * r;
  return ldexp(fma(tj, q, tj), n >> 8);
}

// Z-derived, data-independent quantities of one layer.
struct ZTerms {
  int M, Q, QP;
  double* c;      // [QP]  column mean of Z (shift; the bound is translation invariant)
  double* lam;    // [QP]  inv_l padded with zeros
  double* Zc;     // [M x QP]  Z - c, zero padded
  double* ZcT;    // [QP x 128] transposed copy (per-lane rows read coalesced)
  double* zd;     // [M x M]   -1/4 sum_q lam_q (z_mq - z_m'q)^2
  double* Kuu;    // [M x M]   variance*exp(-1/2 r^2) + jitter*I
};

// Row-derived quantities (one row = one lagged window).
struct RowTerms {
  int64_t N; int QP;
  double* Xmu;    // [N x QP]  window means minus c, zero padded
  double* d2;     // [N x QP]  1/(2S + l^2) = lam/(1+2 S lam)
  double* e1;     // [N x QP]  1/(S + l^2)
  double* em;     // [N x QP]  e1 * mu
  double* g0;     // [N]  log(var^2) - 1/2 sum log(1+2 S lam) - sum d2 mu^2       (psi2 row constant)
  double* g1;     // [N]  log(var) - 1/2 sum log(1+S lam) - 1/2 sum e1 mu^2         (psi1 row constant, expanded form)
};

struct WindowSpec {     // implicit lagged windows (rgp_layer_in)
  int n_seq, X_win, X_dim, U_win, U_dim, D, y_uncertain;
  const int64_t *own_off, *own_len, *up_off, *up_len;
  const int64_t* row_off;             // [n_seq+1] prefix sums of N_s (device)
  const double *own_mean, *own_var, *up_mean, *up_var;
  double up_var_const;
};

size_t psi2_num_blocks(rgp_handle h, int64_t N);
bool psi2_use_rows4(int M, int QP);
int psi2_packed_teams(int M, int QP);     // > 4: packed-team kernel with that many rows in flight per workgroup
// grid size and number of per-workgroup partial slabs each psi2 launch produces
struct Psi2Plan { int grid, n_p2, n_dz, n_dl; };
Psi2Plan psi2_plan(rgp_handle h, int64_t N, int M, int QP, bool fwd, bool bwd);      // fused pass with four teams per workgroup and the dZ accumulator in LDS

void launch_prep_z(hipStream_t s, int M, int Q, int QP, const double* Z, const double* variance, const double* inv_l,
                   double jitter, const ZTerms& zt);
void launch_row_offsets(hipStream_t s, int n_seq, int X_win, const int64_t* own_len, int64_t* row_off);
void launch_rowprep_dense(hipStream_t s, int64_t N, int Q, const double* mu, const double* S, const double* variance,
                          const ZTerms& zt, const RowTerms& rt);
// windowed: also materialises Ym (N x D), Yv (N x D, if y_uncertain) and the partial sums of sum(Ym^2)+sum(Yv)
void launch_rowprep_window(hipStream_t s, int64_t N, int Q, const WindowSpec& w, const double* variance,
                           const ZTerms& zt, const RowTerms& rt, double* Ym, double* Yv, double* yry_out);
void launch_psi1_fwd(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, double* psi1);
// Lp = dL_dpsi1 * psi1; row gradients gmu/gS (=), partials of dinv_l [nblk x QP] and sum(Lp) [nblk]
void launch_psi1_bwd(hipStream_t s, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt, const double* dLdpsi1,
                     const double* psi1, double* Lp, double* gmu, double* gS, double* dlam_part, double* lsum_part,
                     int nblk);
// dZ partials [nblk x M x QP] of the psi1 term
void launch_psi1_dz(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, const double* Lp,
                    double* dz_part, int nblk);
// SVI psi1 side in three passes (forward + Lp + psi1Y partials, row gradients, dZ partials); D <= 16
#define PSI1_ROWS_MAXBLK 1024
bool psi1_svi_supported(int D);
int psi1_svi_rows_blocks(int64_t N);
void launch_psi1_svi(hipStream_t s, int64_t N, int M, int Q, int D, const ZTerms& zt, const RowTerms& rt, const double* Ym,
                     const double* Cpsi1, const double* noise, double* P, double* Lp, double* psi1Y_part, double* gmu,
                     double* gS, double* dYm, double* dlam_part, double* lsum_part, double* dz_part, int nblk);
// the same three passes for wide outputs (any D): Lp holds T = Ym Cpsi1^T on entry and Lp = psi1 o T on exit; P = psi1;
// psi1Y = Ym^T P and dL/dY = P Cpsi1 - beta Ym are products the caller runs on the GEMM kernel
void launch_psi1_svi_wide(hipStream_t s, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt, const double* noise,
                          double* P, double* Lp, double* gmu, double* gS, double* dlam_part, double* lsum_part, double* dz_part,
                          int nblk);
// fused psi2 pass: forward sum (tri-packed partials) and/or backward for given symmetric dLs (M x M)
struct Psi2Out {
  double* p2_part;     // [nblk x tri]           (forward)
  double* dz_part;     // [nblk x QP x 128]      (backward; layout [q][m])
  double* dlam_part;   // [nblk x QP]
  double* wsum_part;   // [nblk]                  sum of all weights (for dvar)
  double* gmu; double* gS; int accumulate;        // row gradients
  double* pair_ws;     // workspace of the pair-form pass (psi2_pair_ws_doubles); nullptr: the sweep kernels only
};
rgp_status launch_psi2(rgp_handle h, bool fwd, bool bwd, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt,
                       const double* dLs, const Psi2Out& out, int nblk);
// Pair-form fused pass on the fp64 matrix pipe (rgp_psi2_pair.hip): geometry of one launch, whether a shape is supported, and
// whether the handle's mode (rgp_set_psi2_pair / RGP_PSI2_PAIR) selects it for a fused forward + backward pass.
struct PairGeom {
  int M, QP, QT, TS, HS, OS, RS, PW, M2, P, G, Ppad, RR, NB;
  int64_t ntiles;
  size_t lds;
};
bool psi2_pair_geom(rgp_handle h, int64_t N, int M, int QP, PairGeom& g);
bool psi2_pair_selected(rgp_handle h, int64_t N, int M, int QP, PairGeom& g);
size_t psi2_pair_ws_doubles(const PairGeom& g, int64_t N);
rgp_status launch_psi2_pair(rgp_handle h, const PairGeom& g, int64_t N, int Q, const ZTerms& zt, const RowTerms& rt,
                            const double* dLs, const Psi2Out& out, double* ws);
// The pair form for 112 < M <= 1024 (blocked layer step): pair groups over 16 x 16 blocks of inducing rows, (group, row range)
// tasks on one workgroup per CU.  h: the (row, m) part of the exponent [N][M] as the blocked psi1 forward pass leaves it.
// Outputs: psi2 [M][M], dz3 [M][QP] (= sum_m' z[m',q] sum_n d2 W), and V2 = 2 v [N][Mld], T [N][QP] inside the workspace, from which
// the caller forms the row gradients and the rest of dZ with matrix products.
struct PairBlkGeom {
  int M, QP, Mld, nb, rem, NT, TG, ntl, G, RR, ntasks, grid, ND2;
  int64_t ntiles;
  size_t lds;
};
struct PairBlkOut {
  double *psi2, *dz3;            // in: where to write
  double *V2, *T;                // out: inside the workspace
};
bool psi2_pairblk_geom(rgp_handle h, int64_t N, int M, int QP, int Mld, PairBlkGeom& g);
bool psi2_pairblk_selected(rgp_handle h, int64_t N, int M, int QP, int Mld, PairBlkGeom& g);
size_t psi2_pairblk_ws_doubles(const PairBlkGeom& g, int64_t N);
// where the h blocks [row tile][block of 16 inducing rows][16][16] (= h + g0 / 2; -inf beyond N, 0 beyond M) live inside the
// workspace: a caller that fills them itself passes H = nullptr to launch_psi2_pairblk
double* psi2_pairblk_hq(const PairBlkGeom& g, double* ws);
rgp_status launch_psi2_pairblk(rgp_handle h, hipStream_t s, const PairBlkGeom& g, int64_t N, const ZTerms& zt, const RowTerms& rt,
                               const double* H, const double* dLs, double* ws, PairBlkOut& out);
// out[i] = scale * sum_g part[g*stride + i], i < len (fixed order)
void launch_reduce(hipStream_t s, int nblk, int64_t len, int64_t stride, const double* part, double* out, double scale,
                   int accumulate);
// up to REDUCE_BATCH_MAX reductions out[i] = sum_g part[g*stride + i] in one launch
#define REDUCE_BATCH_MAX 8
struct ReduceSeg { const double* part; double* out; int nblk; int64_t len, stride; };
struct ReduceBatch { ReduceSeg seg[REDUCE_BATCH_MAX]; int nseg; int blk_off[REDUCE_BATCH_MAX]; };
void launch_reduce_batch(hipStream_t s, int nseg, const ReduceSeg* segs);
void launch_unpack_tri(hipStream_t s, int M, const double* tri, double* full);
