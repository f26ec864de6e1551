// rgp_big_impl.h -- the SVI layer step for 128 < M <= 1024 inducing points (textually included by rgp_layer.hip, whose small
// kernels -- window tables, latent overlap-add, gradient finalisation -- it shares).
//
// What changes against the M <= 128 path (rgp_psi.hip) and why:
//   * psi2.  Neither the M x M forward sum (2 MB at M = 500) nor the [Q][M] dZ slab fits LDS next to the per-row state,
//     so the fused pass splits in two:
//       FORWARD  big_psi2_fwd_kernel: a wave owns 64 inducing rows m (lanes) x NP inducing columns m' and keeps
//                zz[q] = z[m,q] z[m',q] in registers; it streams over the data rows with the row's scaled inverse widths as
//                scalar operands, so psi2[m,m'] accumulates in a REGISTER (no LDS, no atomics, no barriers, fixed order).
//                Only column groups that reach the diagonal (m' >= m) are computed: 56 % of the pairs at M = 500.
//       BACKWARD big_psi2_bwd_kernel: as the fused kernel, thread = (data row, inducing row m of one block of <= 256), sweep
//                over ALL m' with z[m',:] as scalar operands; dZ slab [Q][<=256] in LDS; the per-row sums over m are
//                written per inducing block and combined by big_rowgrad_kernel.
//     h[n,m] (the (row, m) part of the exponent) is computed once by the psi1 forward pass and read by both.
//   * psi1 side.  Every contraction is a tall-skinny product and runs on the MFMA GEMM (rgp_gemm.hip):
//       Lp = psi1 o (Ym Cpsi1^T),  psi1Y = Ym^T psi1,  dYm = psi1 Cpsi1 - beta Ym,  [s1 | s2 | s0] = Lp [Z | Z^2 | 1],
//       dZ = Lp^T (e mu) - Z o (Lp^T e)        (svi_vardtc.py:164-167,192 and the psi1 gradients of GPy's rbf_psi_comp)
//   * M x M chain: blocked, multi-workgroup (rgp_chain.hip, svi_chain_pre_big / svi_chain_post_big).
// Everything runs on the handle's main stream: one layer step is >100 ms at these sizes.
#include "rgp_gemm.h"

#define RGP_BIG_MAX_M 1024
// forward psi2 kernel: column groups per wave task, rows per tile, inducing columns per lane (padded Q <= 48)
#ifndef BF_G
#define BF_G 8
#endif
#ifndef BF_RT
#define BF_RT 16
#endif
#ifndef BIG_BWD_SYNC_EVERY
#define BIG_BWD_SYNC_EVERY 0      // power of two (inducing columns between two meetings of the workgroup's waves), 0 = never;
                                  // M = 500, Q = 40: 35.0 TF never, 37.8 every 16, 38.6 every 64 (without the priority turns)
#endif
#ifndef BIG_BWD_PRIO_ALT
#define BIG_BWD_PRIO_ALT 16       // columns per turn of the alternating issue priority of a SIMD's two waves (0: off).  The
                                  // hardware serves the older wave first and the younger one in its gaps, so the two drift
                                  // apart by more z rows than the scalar cache holds; with turns they advance together:
                                  // 40.1 / 39.7 / 40.3 TF with turns of 4 / 8 / 16 columns
#endif
#ifndef BF_NP
#define BF_NP 2           // inducing columns per lane for padded Q <= 48 (1: half the registers, four waves per SIMD)
#endif

struct BigPlan {
  int QP, W, Mld, Mp;
  int nI, IB, TS, NT, bwd_grid;          // backward: inducing blocks, block size, team stride, rows in flight, row workgroups
  int NP, fwd_tasks, fwd_chunks;         // forward: columns per wave, wave tasks, row chunks
  int sk;                                // K splits of the products that sum over the data rows
  int rows_blk;                          // workgroups of the row-gradient kernels
  int num_cu;                            // compute units of the device the plan was made for
  bool pb; PairBlkGeom pbg;              // the psi2 term of the SVI bound runs in the pair form (rgp_psi2_pair.hip, blocked form)
};

static BigPlan big_plan(rgp_handle h, int64_t N, int M, int Q) {
  BigPlan p{};
  p.num_cu = h->num_cu;
  p.QP = rgp_qp(Q);
  p.W = 2 * p.QP + 8;
  p.Mld = round_up(M, 64);
  p.Mp = round_up(M, 8);
  p.nI = (M + 255) / 256;
  p.IB = (M + p.nI - 1) / p.nI;
  p.TS = round_up(p.IB, 64);
  p.NT = 512 / p.TS;
  int64_t g = h->num_cu / p.nI;
  if (g < 1) g = 1;
  const int64_t maxg = (N + p.NT - 1) / p.NT;
  if (g > maxg) g = maxg;
  p.bwd_grid = (int)(g < 1 ? 1 : g);
  p.NP = p.QP <= 48 ? BF_NP : 1;
  const int CG = p.NP * BF_G;                                          // inducing columns per wave task
  const int nIB = (M + 63) / 64, nSG = (M + CG - 1) / CG;
  int tasks = 0;
  for (int ib = 0; ib < nIB; ++ib) tasks += nSG - (64 * ib) / CG;
  p.fwd_tasks = tasks;
  const int fblocks = (tasks + 7) / 8;
  int64_t ch = ((int64_t)4 * h->num_cu + fblocks - 1) / fblocks;       // ~4 rounds of one 512-thread workgroup per CU
  if (ch > 128) ch = 128;
  if (ch > (N + 15) / 16) ch = (N + 15) / 16;
  p.fwd_chunks = (int)(ch < 1 ? 1 : ch);
  int64_t sk = (N + 511) / 512;
  p.sk = (int)(sk < 1 ? 1 : (sk > 256 ? 256 : sk));
  int64_t rb = (N + 3) / 4;
  p.rows_blk = (int)(rb < 1 ? 1 : (rb > 1024 ? 1024 : rb));
  p.pb = p.QP <= RGP_MAX_QP && psi2_pairblk_selected(h, N, M, p.QP, p.Mld, p.pbg);
  return p;
}

#define WQ_MAX_QP 1024        // widest padded window of the wide path (rgp_wideq_impl.h)
#define WQ_RB_MAX 24          // data rows per workgroup of its exponential pass (two M-wide LDS rows each, 60 KB at most)
#define WQ_E_BYTES (6ull << 30)   // the N x P weights are held for this many bytes of data rows at a time
struct WqWs {                 // what the wide-window psi2 term needs beyond the blocked path's buffers
  double *E, *ZZ2, *vrow, *T, *DM, *S22, *p2w_part, *p2pair, *C_part, *C, *dz3, *B_part, *B1, *B2, *ellq;
  int *pa, *pb;
  int P, rb, nsplit;
  int64_t nc;                 // data rows per chunk
};

static size_t wq_carve(WsAlloc& ws, const BigPlan& p, int64_t N, int M, WqWs& Q, int num_cu) {
  const int P = M * (M + 1) / 2;
  Q.P = P;
  Q.rb = (7680 - WQ_RB_MAX) / (2 * M);
  if (Q.rb > WQ_RB_MAX) Q.rb = WQ_RB_MAX;
  if (Q.rb < 1) Q.rb = 1;
  int64_t nc = (int64_t)(WQ_E_BYTES / ((size_t)P * sizeof(double)));
  if (nc < 1024) nc = 1024;
  if (const char* e = getenv("RGP_WQ_CHUNK_ROWS")) { const long v = atol(e); if (v > 0) nc = v; }     // tests: several chunks at small N
  if (nc > N) nc = N;
  Q.nc = nc;
  const int64_t nexp = (nc + Q.rb - 1) / Q.rb;
  const int tiles = ((P + 127) / 128) * ((p.QP + 127) / 128);  // 128 x 128 output tiles of the row-split product C
  // row splits of C: 1024 to 3072 workgroups of equal length, the count that leaves the fewest of the chip's workgroup
  // slots (two per CU) idle in the last round
  Q.nsplit = 1;
  const int slots = 2 * (num_cu > 0 ? num_cu : 256);
  if (tiles < 2048) {
    double best = 0.0;
    int lo = (1024 + tiles - 1) / tiles, hi = (3072 + tiles - 1) / tiles;
    if (lo > 32) lo = 32;
    if (hi > 32) hi = 32;
    for (int ns = lo; ns <= hi; ++ns) {
      const int w = tiles * ns;
      const double fill = (double)w / ((double)slots * ((w + slots - 1) / slots));
      if (fill > best + 1e-9) { best = fill; Q.nsplit = ns; }
    }
  }
  Q.E = ws.take<double>((size_t)nc * P);
  Q.ZZ2 = ws.take<double>((size_t)p.QP * P);
  Q.vrow = ws.take<double>((size_t)N * M);
  Q.T = ws.take<double>((size_t)N * p.QP);
  Q.DM = ws.take<double>((size_t)N * p.QP);
  Q.S22 = ws.take<double>((size_t)N * p.W);
  Q.p2w_part = ws.take<double>((size_t)nexp * P);
  Q.p2pair = ws.take<double>(P);
  Q.C_part = ws.take<double>((size_t)Q.nsplit * P * p.QP);
  Q.C = ws.take<double>((size_t)P * p.QP);
  Q.dz3 = ws.take<double>((size_t)M * p.QP);
  Q.B_part = ws.take<double>((size_t)p.sk * M * p.QP);
  Q.B1 = ws.take<double>((size_t)M * p.QP);
  Q.B2 = ws.take<double>((size_t)M * p.QP);
  Q.ellq = ws.take<double>(M);
  Q.pa = ws.take<int>(P);
  Q.pb = ws.take<int>(P);
  return ws.off;
}

struct BigWs {
  WqWs wq; bool wide;
  ZTerms zt; RowTerms rt;
  double *dm, *P, *Lp, *H, *gmu, *gS, *Ym, *Yv, *dYm, *S12, *ZZ;
  double *p2_part, *psi2, *dLs;
  double *RP, *dz2_part, *dz2, *dlam2_part, *dlam2, *wsum2_part, *wsum2;
  double *dzA_part, *dzA1, *dzA2, *dz1, *dlam1_part, *dlam1, *lsum1_part, *lsum1;
  double *yry_part, *yry, *psi1Y_part, *psi1Y, *psi0sum, *psi0coef, *elbo_part, *fin_dlrow, *fin_gsum;
  int64_t *row_off, *own_pre, *up_pre;
  double *r2, *T2, *PsiR_part, *PsiR, *dYvar_row;     // collapsed bound only
  double* fp2_part; int frs;                            // psi2 forward on the matrix pipe: partial sums per row split, row splits
  double *pb_ws, *pb_S22, *pb_DM, *pb_B_part, *pb_B1, *pb_B2, *pb_dz3;      // pair form of the psi2 term (bp_psi2)
};

static size_t big_carve(char* base, const BigPlan& p, int64_t N, int M, int Q, int D, int n_seq, BigWs& L, SviChain* ch,
                        VarDtcChain* vc = nullptr) {
  WsAlloc ws(base);
  const int QP = p.QP;
  const int Dd = D > 0 ? D : 1;
  const size_t nq = (size_t)N * QP, nm = (size_t)N * M, nd = (size_t)N * Dd, mm = (size_t)M * M;
  L.zt.M = M; L.zt.Q = Q; L.zt.QP = QP;
  L.zt.c = ws.take<double>(QP); L.zt.lam = ws.take<double>(QP); L.zt.Zc = ws.take<double>((size_t)M * QP);
  L.zt.ZcT = ws.take<double>((size_t)QP * RGP_PSI2_TEAM); L.zt.zd = ws.take<double>(mm); L.zt.Kuu = ws.take<double>(mm);
  if (ch) svi_chain_carve(*ch, ws);
  L.psi2 = ws.take<double>(mm); L.dLs = ws.take<double>(mm);
  L.ZZ = ws.take<double>((size_t)M * p.W);
  L.rt.N = N; L.rt.QP = QP;
  L.rt.Xmu = ws.take<double>(nq); L.rt.d2 = ws.take<double>(nq); L.rt.e1 = ws.take<double>(nq); L.rt.em = ws.take<double>(nq);
  L.rt.g0 = ws.take<double>(N); L.rt.g1 = ws.take<double>(N);
  L.dm = ws.take<double>(nq);
  L.P = ws.take<double>(nm); L.Lp = ws.take<double>(nm); L.H = ws.take<double>(nm);
  L.gmu = ws.take<double>(nq); L.gS = ws.take<double>(nq);
  L.Ym = ws.take<double>(nd); L.Yv = ws.take<double>(nd); L.dYm = ws.take<double>(nd);
  L.S12 = ws.take<double>((size_t)N * p.W);
  L.wide = QP > RGP_MAX_QP;
  L.p2_part = L.RP = L.dz2_part = nullptr;
  if (!L.wide) {                       // buffers of the blocked sweeps; the wide-window forms (wq_carve) never touch them
    L.p2_part = ws.take<double>((size_t)p.fwd_chunks * mm);
    L.RP = ws.take<double>((size_t)p.nI * N * (2 * QP + 2));
    L.dz2_part = ws.take<double>((size_t)p.bwd_grid * QP * p.Mld);
  }
  L.dz2 = ws.take<double>((size_t)QP * p.Mld);
  L.dlam2_part = ws.take<double>((size_t)p.rows_blk * QP); L.dlam2 = ws.take<double>(QP);
  L.wsum2_part = ws.take<double>(p.rows_blk); L.wsum2 = ws.take<double>(1);
  L.dzA_part = ws.take<double>((size_t)p.sk * M * QP); L.dzA1 = ws.take<double>((size_t)M * QP); L.dzA2 = ws.take<double>((size_t)M * QP);
  L.dz1 = ws.take<double>((size_t)M * QP);
  L.dlam1_part = ws.take<double>((size_t)p.rows_blk * QP); L.dlam1 = ws.take<double>(QP);
  L.lsum1_part = ws.take<double>(p.rows_blk); L.lsum1 = ws.take<double>(1);
  L.yry_part = ws.take<double>(2048); L.yry = ws.take<double>(1);
  L.psi1Y_part = ws.take<double>((size_t)p.sk * Dd * M); L.psi1Y = ws.take<double>((size_t)Dd * M);
  L.psi0sum = ws.take<double>(1); L.psi0coef = ws.take<double>(1); L.elbo_part = ws.take<double>(LAT_BLOCKS);
  L.fin_dlrow = ws.take<double>((size_t)M * QP); L.fin_gsum = ws.take<double>(M);
  L.row_off = ws.take<int64_t>(n_seq + 1); L.own_pre = ws.take<int64_t>(n_seq + 1); L.up_pre = ws.take<int64_t>(n_seq + 1);
  if (L.wide) wq_carve(ws, p, N, M, L.wq, p.num_cu);
  L.fp2_part = nullptr; L.frs = 1;
  if (!L.wide && (base == nullptr || getenv("RGP_PSI2_FWD_MFMA") != nullptr)) {     // sizing passes count it: an upper bound

    const int nb = (M + 31) / 32, blocks = nb * (nb + 1) / 2;      // 32 x 32 pair blocks of the matrix-pipe forward kernel
    int64_t rs = (1536 + blocks - 1) / blocks;
    if (rs > (N + 63) / 64) rs = (N + 63) / 64;
    L.frs = (int)(rs < 1 ? 1 : rs);
    L.fp2_part = ws.take<double>((size_t)L.frs * mm);
  }
  L.pb_ws = nullptr;
  if (p.pb) {
    L.pb_ws = ws.take<double>(psi2_pairblk_ws_doubles(p.pbg, N));
    L.pb_S22 = ws.take<double>((size_t)N * p.W); L.pb_DM = ws.take<double>(nq);
    L.pb_B_part = ws.take<double>((size_t)p.sk * M * QP);
    L.pb_B1 = ws.take<double>((size_t)M * QP); L.pb_B2 = ws.take<double>((size_t)M * QP); L.pb_dz3 = ws.take<double>((size_t)M * QP);
  }
  if (vc) {
    vardtc_chain_carve(*vc, ws);
    L.r2 = ws.take<double>(N); L.T2 = ws.take<double>(nm); L.dYvar_row = ws.take<double>(N);
    L.PsiR_part = ws.take<double>((size_t)p.sk * mm); L.PsiR = ws.take<double>(mm);
  }
  return ws.off;
}

static int64_t big_workspace_bytes(int64_t N, int M, int Q, int D) {
  rgp_ctx tmp;
  BigWs L;
  SviChain ch{}; ch.M = M; ch.D = D > 0 ? D : 1;
  VarDtcChain vc{}; vc.M = M; vc.D = D > 0 ? D : 1;
  const BigPlan p = big_plan(&tmp, N, M, Q);
  return (int64_t)big_carve(nullptr, p, N, M, Q, D, 4096, L, &ch, &vc);       // both bounds' buffers: an upper bound for either
}

static inline unsigned big_nb256(int64_t n) { return (unsigned)((n + 255) / 256); }

__global__ void __launch_bounds__(256) big_fill_kernel(int64_t n, double* __restrict__ x, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

#include "rgp_wideq_impl.h"

// ------------------------------------------------------------------------------------------------------------------
// small element-wise kernels
// ------------------------------------------------------------------------------------------------------------------
// ZZ[m] = [ z[m, 0..QP) | z[m, 0..QP)^2 | 1 | 0 ... ]       (right operand of the row-sum product)
__global__ void __launch_bounds__(256) big_zz_kernel(int M, int QP, int W, const double* __restrict__ Zc, double* __restrict__ ZZ) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * W) return;
  const int m = (int)(i / W), j = (int)(i % W);
  double v = 0.0;
  if (j < QP) v = Zc[(size_t)m * QP + j];
  else if (j < 2 * QP) { const double z = Zc[(size_t)m * QP + j - QP]; v = z * z; }
  else if (j == 2 * QP) v = 1.0;
  ZZ[i] = v;
}

// dm = -1/2 (d2 - lam): the per-row weights of the Gram term in the psi2 exponent
__global__ void __launch_bounds__(256) big_dm_kernel(int64_t n, int QP, const double* __restrict__ d2, const double* __restrict__ lam,
                                                      double* __restrict__ dm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dm[i] = -0.5 * (d2[i] - lam[i % QP]);
}

// dz1 = A1 - Zc o A2        (A1 = Lp^T (e mu), A2 = Lp^T e)
__global__ void __launch_bounds__(256) big_dz1_kernel(int64_t n, const double* __restrict__ A1, const double* __restrict__ A2,
                                                       const double* __restrict__ Zc, double* __restrict__ dz1) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz1[i] = A1[i] - Zc[i] * A2[i];
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 forward and the (row, m) part of the psi2 exponent.  Thread = inducing row m (z[m,:] in registers); the row
// operands are wave-uniform scalar loads.
//   P[n,m] = exp(g1[n] - 1/2 sum_q z (e z - 2 e mu));   H[n,m] = sum_q d2 z (mu - z/4) - 1/4 sum_q lam z^2
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ void __launch_bounds__(256) big_psi1_fwd_kernel(const int64_t N, const int M, const double* __restrict__ Zc,
                                                            const double* __restrict__ lam, const double* __restrict__ e1,
                                                            const double* __restrict__ em, const double* __restrict__ d2,
                                                            const double* __restrict__ Xmu, const double* __restrict__ g1,
                                                            double* __restrict__ P, double* __restrict__ H) {
  const int m = blockIdx.y * 256 + threadIdx.x;
  const bool act = m < M;
  const int mc = act ? m : M - 1;
  double z[QP];
  double ellq = 0.0;
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    z[q] = Zc[(size_t)mc * QP + q];
    ellq += 0.25 * lam[q] * z[q] * z[q];
  }
  for (int64_t r = blockIdx.x; r < N; r += gridDim.x) {
    const double* __restrict__ e = e1 + r * QP;
    const double* __restrict__ mm = em + r * QP;
    const double* __restrict__ d = d2 + r * QP;
    const double* __restrict__ mu = Xmu + r * QP;
    double a0 = 0.0, a1 = 0.0, h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int q = 0; q < QP; q += 2) {
      a0 += z[q] * (e[q] * z[q] - 2.0 * mm[q]);
      a1 += z[q + 1] * (e[q + 1] * z[q + 1] - 2.0 * mm[q + 1]);
      if (H) {
        h0 += z[q] * d[q] * (mu[q] - 0.25 * z[q]);
        h1 += z[q + 1] * d[q + 1] * (mu[q + 1] - 0.25 * z[q + 1]);
      }
    }
    if (act) {
      if (P) P[r * M + m] = exp(g1[r] - 0.5 * (a0 + a1));
      if (H) H[r * M + m] = (h0 + h1) - ellq;
    }
  }
}

// The same two arrays on the matrix pipe: per tile of 16 data rows and 16 inducing rows, two [16 x 2 QP] x [2 QP x 16] products
//   psi1 exponent = g1 + (e mu) Z^T + (2 e) (-Z^2 / 4)^T,      h = (d2 mu) Z^T + d2 (-Z^2 / 4)^T - 1/4 sum_q lam z^2
// (v_mfma_f64_16x16x4; wave = one tile of rows, left operands straight from global memory, right operands from LDS images of Z
// and -Z^2 / 4 for a chunk of up to 112 inducing rows = blockIdx.y).  h can also be written as the pair form's record wants it
// (Hq [row tile][block of 16 inducing rows][16][16] = h + g0 / 2, -inf for rows beyond N, 0 for columns beyond M).
#define BP1_MT 7              // tiles of 16 inducing rows per chunk: 2 x 112 x (QP + 1) doubles of LDS, two workgroups per CU
typedef double bp1_d4 __attribute__((ext_vector_type(4)));
template <int QP>
__global__ void __launch_bounds__(256, 2) big_psi1_fwd_mfma_kernel(const int64_t N, const int64_t ntiles, const int M, const int nb,
                                                                    const double* __restrict__ Zc, const double* __restrict__ lam,
                                                                    const double* __restrict__ e1, const double* __restrict__ em,
                                                                    const double* __restrict__ d2, const double* __restrict__ Xmu,
                                                                    const double* __restrict__ g1, const double* __restrict__ g0,
                                                                    double* __restrict__ P, double* __restrict__ H,
                                                                    double* __restrict__ Hq) {
  constexpr int ZQ = QP + 1, KQ = QP / 4;
  extern __shared__ double bp1_sm[];
  const int mt0 = blockIdx.y * BP1_MT;
  const int nmt = (nb - mt0 < BP1_MT) ? nb - mt0 : BP1_MT;
  double* const Zs = bp1_sm;                      // [16 BP1_MT][ZQ]   z
  double* const Z2s = Zs + 16 * BP1_MT * ZQ;      // [16 BP1_MT][ZQ]   -z^2 / 4
  double* const el = Z2s + 16 * BP1_MT * ZQ;      // [16 BP1_MT]       1/4 sum_q lam z^2
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 16 * nmt * ZQ; i += 256) {
    const int ml = i / ZQ, q = i - ml * ZQ, m = 16 * mt0 + ml;
    const double z = (m < M && q < QP) ? Zc[(size_t)m * QP + q] : 0.0;
    Zs[i] = z;
    Z2s[i] = -0.25 * z * z;
  }
  __syncthreads();
  for (int ml = t; ml < 16 * nmt; ml += 256) {
    double e = 0.0;
    for (int q = 0; q < QP; ++q) e -= lam[q] * Z2s[ml * ZQ + q];
    el[ml] = e;
  }
  __syncthreads();
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t n0 = tile * 16, n = n0 + c;
    const bool ok = n < N;
    const size_t o = (size_t)(ok ? n : N - 1) * QP + g;
    double Adm[KQ], Ad[KQ], Aem[KQ], Ae[KQ];
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      const double dv = ok ? d2[o + 4 * kk] : 0.0, ev = ok ? e1[o + 4 * kk] : 0.0;
      Ad[kk] = dv;
      Adm[kk] = dv * Xmu[o + 4 * kk];
      Ae[kk] = 2.0 * ev;
      Aem[kk] = ok ? em[o + 4 * kk] : 0.0;
    }
    double g1r[4], g0h[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t nr = n0 + 4 * rr + g;
      g1r[rr] = (P && nr < N) ? g1[nr] : 0.0;
      g0h[rr] = (nr < N) ? (Hq ? 0.5 * g0[nr] : 0.0) : -INFINITY;
    }
    for (int mt = 0; mt < nmt; ++mt) {
      bp1_d4 aH = (bp1_d4){0.0, 0.0, 0.0, 0.0}, aP = (bp1_d4){0.0, 0.0, 0.0, 0.0};
      const int zo = (16 * mt + c) * ZQ + g;
#pragma unroll
      for (int kk = 0; kk < KQ; ++kk) {
        const double zs = Zs[zo + 4 * kk], z2 = Z2s[zo + 4 * kk];
        if (H || Hq) {
          aH = __builtin_amdgcn_mfma_f64_16x16x4f64(Adm[kk], zs, aH, 0, 0, 0);
          aH = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad[kk], z2, aH, 0, 0, 0);
        }
        if (P) {
          aP = __builtin_amdgcn_mfma_f64_16x16x4f64(Aem[kk], zs, aP, 0, 0, 0);
          aP = __builtin_amdgcn_mfma_f64_16x16x4f64(Ae[kk], z2, aP, 0, 0, 0);
        }
      }
      const int m = 16 * (mt0 + mt) + c;
      const double elm = el[16 * mt + c];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int64_t nr = n0 + 4 * rr + g;
        const double h = aH[rr] - elm;
        if (nr < N && m < M) {
          if (P) P[(size_t)nr * M + m] = exp(g1r[rr] + aP[rr]);
          if (H) H[(size_t)nr * M + m] = h;
        }
        if (Hq) Hq[(((size_t)tile * nb + mt0 + mt) * 16 + 4 * rr + g) * 16 + c] = (nr < N) ? ((m < M) ? h + g0h[rr] : 0.0) : -INFINITY;
      }
    }
  }
}

template <int QP>
static hipError_t bp1_launch(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, double* P, double* H, double* Hq,
                             int num_cu) {
  const int nb = (M + 15) / 16, nch = (nb + BP1_MT - 1) / BP1_MT;
  const int64_t ntiles = (N + 15) / 16;
  const size_t lds = (2 * (size_t)16 * BP1_MT * (QP + 1) + 16 * BP1_MT) * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)big_psi1_fwd_mfma_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int64_t gx = (2 * (int64_t)(num_cu > 0 ? num_cu : 256) + nch - 1) / nch;       // about two workgroups per CU in all
  if (gx > (ntiles + 3) / 4) gx = (ntiles + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((big_psi1_fwd_mfma_kernel<QP>), dim3((unsigned)gx, (unsigned)nch), dim3(256), lds, s, N, ntiles, M, nb, zt.Zc, zt.lam,
                     rt.e1, rt.em, rt.d2, rt.Xmu, rt.g1, rt.g0, P, H, Hq);
  return hipSuccess;
}

// psi1 (P, may be null), h (H, row-major, may be null) and / or h in the pair form's block layout (Hq, may be null).
// RGP_PSI1_FWD_MFMA=0 in the environment keeps the vector-pipe kernel (which cannot write Hq).
static bool big_psi1_fwd_on_matrix_pipe() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RGP_PSI1_FWD_MFMA"); v = e ? (atoi(e) != 0) : 1; }
  return v != 0;
}
static void launch_big_psi1_fwd(hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt, double* P, double* H,
                                double* Hq = nullptr, int num_cu = 0) {
  if (Hq || big_psi1_fwd_on_matrix_pipe()) {
    switch (zt.QP) {
#define B1M(qp) case qp: (void)bp1_launch<qp>(s, N, M, zt, rt, P, H, Hq, num_cu); return;
      B1M(8) B1M(16) B1M(20) B1M(24) B1M(32) B1M(40) B1M(48) B1M(56) B1M(64)
#undef B1M
      default: break;
    }
  }
  const dim3 grid((unsigned)(N < 4096 ? N : 4096), (unsigned)((M + 255) / 256));
  switch (zt.QP) {
#define B1F(qp) case qp: hipLaunchKernelGGL((big_psi1_fwd_kernel<qp>), grid, dim3(256), 0, s, N, M, zt.Zc, zt.lam, rt.e1, rt.em, rt.d2, rt.Xmu, rt.g1, P, H); break;
    B1F(8) B1F(16) B1F(20) B1F(24) B1F(32) B1F(40) B1F(48) B1F(56) B1F(64)
#undef B1F
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// psi1 row gradients from the row sums S12[n] = [ s1[0..QP) | s2[0..QP) | s0 | 0.. ]  (s_k = sum_m Lp[n,m] z[m,q]^k):
//   dmu = -e (mu s0 - s1);  dS = 1/2 e^2 S2 - 1/2 e s0,  S2 = mu^2 s0 - 2 mu s1 + s2;
//   dinv_l[q] += -1/2 (e/lam)^2 S2 - 1/2 s0 (1 - e/lam)/lam;   lsum += s0
// Workgroup = 4 row slots x 64 q lanes; per-workgroup partials in a fixed order.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) big_rows1_kernel(int64_t N, int Q, int QP, int W, const double* __restrict__ S12,
                                                         const double* __restrict__ Xmu, const double* __restrict__ e1,
                                                         const double* __restrict__ lam, double* __restrict__ gmu,
                                                         double* __restrict__ gS, double* __restrict__ dlam_part,
                                                         double* __restrict__ lsum_part) {
  __shared__ double sm[4][64];
  __shared__ double sl[4];
  const int q = (int)blockIdx.y * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;     // blockIdx.y: 64 window columns each
  double dl = 0.0, ls = 0.0;
  const double il = (q < Q) ? 1.0 / lam[q] : 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 4 + g; n < N; n += (int64_t)gridDim.x * 4) {
    const double s0 = S12[n * W + 2 * QP];
    if (q == 0) ls += s0;
    if (q < Q) {
      const double s1 = S12[n * W + q], s2 = S12[n * W + QP + q];
      const double mu = Xmu[n * QP + q], e = e1[n * QP + q];
      const double S2 = mu * mu * s0 - 2.0 * mu * s1 + s2;
      gmu[n * QP + q] = -e * (mu * s0 - s1);
      gS[n * QP + q] = 0.5 * e * e * S2 - 0.5 * e * s0;
      const double el = e * il;
      dl += -0.5 * el * el * S2 - 0.5 * s0 * (1.0 - el) * il;
    } else if (q < QP) {
      gmu[n * QP + q] = 0.0;
      gS[n * QP + q] = 0.0;
    }
  }
  const int ql = threadIdx.x & 63;
  sm[g][ql] = dl;
  if (q == 0) sl[g] = ls;
  __syncthreads();
  if (g == 0) {
    if (q < QP) dlam_part[(size_t)blockIdx.x * QP + q] = (sm[0][ql] + sm[1][ql]) + (sm[2][ql] + sm[3][ql]);
    if (q == 0) lsum_part[blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// psi2 forward.  psi2[m,m'] = sum_n exp( g0[n] + H[n,m] + H[n,m'] + sum_q dm[n,q] z[m,q] z[m',q] ).
// Wave task = (block of 64 inducing rows m = lanes, BF_G groups of NP inducing columns m' that reach the block's diagonal).
// The products zz[q] = z[m,q] z[m',q] of the group in work sit in registers, dm[n,:], g0[n] and H[n,m'] are scalar operands,
// and psi2[m,m'] accumulates in a register: no LDS, no atomics, fixed order.
// The data rows are walked in tiles of BF_RT rows, and each tile is swept once per column group before the wave moves on:
// the tile's scalar operands (5 KB) are then served by the scalar cache for all but the first sweep; streaming every row
// once per group from L2/HBM left the VALU idle two thirds of the time (one row = 80 multiply-adds per 330 B of scalar loads
// that cannot be double-buffered in ~100 SGPRs).  The eight waves of a workgroup share the tile (barrier per tile).
// part: [chunk][M][M]; entry (m, m') is written only for m' >= 64 floor(m/64) (big_psi2_unpack_kernel reads m' >= m).
// ------------------------------------------------------------------------------------------------------------------
template <int QP, int NP>
__global__ void __launch_bounds__(512, (NP == 1 && QP <= 48) ? 4 : 2)
big_psi2_fwd_kernel(const int64_t N, const int M, const int ntasks, const double* __restrict__ Zc, const double* __restrict__ dm,
                    const double* __restrict__ H, const double* __restrict__ g0, double* part) {
  extern __shared__ double lds[];
  double* etab = lds;                                           // [256]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int CG4 = 16;                                       // row stride of hx (>= columns per task)
  static_assert(NP * BF_G <= CG4 && BF_RT % 4 == 0, "tile staging layout");
  double* hl = lds + PSI2_EXP_TAB + (size_t)wave * BF_RT * (64 + CG4);      // [BF_RT][64], private to the wave
  double* hx = hl + BF_RT * 64;                                              // [BF_RT][CG4]
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  int task = (int)blockIdx.x * 8 + wave;
  const bool has_task = task < ntasks;
  constexpr int CG = NP * BF_G;                                 // inducing columns per task
  const int nIB = (M + 63) / 64, nSG = (M + CG - 1) / CG;
  int ib = 0;
  for (; ib < nIB - 1; ++ib) {
    const int cnt = nSG - (64 * ib) / CG;
    if (task < cnt) break;
    task -= cnt;
  }
  const int col_base = ((64 * ib) / CG + task) * CG;            // wave-uniform
  const int m = 64 * ib + lane;
  const bool act = has_task && m < M;
  const int mc = m < M ? m : M - 1;
  // The lane's own z row is needed again for every column group of every row tile (the products zz are rebuilt per group).
  // As global loads that is 20 loads of 64 different cache lines each per group, and the wave waits for them; the rows of the
  // workgroup's inducing blocks (at most three: eight consecutive tasks) are therefore kept transposed in LDS, [q][lane].
  constexpr bool STRIP = QP <= 48;
  double* zstrip = lds + PSI2_EXP_TAB + (size_t)8 * BF_RT * (64 + CG4);       // [3][QP][64]   (STRIP)
  const double* zl = zstrip;
  if (STRIP) {
    int ib0 = 0, task0 = (int)blockIdx.x * 8;
    for (; ib0 < nIB - 1; ++ib0) {
      const int cnt = nSG - (64 * ib0) / CG;
      if (task0 < cnt) break;
      task0 -= cnt;
    }
    const int sidx = has_task ? ib - ib0 : 0;                    // 0..2
    double* zw = zstrip + (size_t)sidx * QP * 64 + lane;
    if (has_task) {
#pragma unroll
      for (int q = 0; q < QP; ++q) zw[q * 64] = Zc[(size_t)mc * QP + q];       // waves of one block write the same values
    }
    zl = zw;
  }
  const int64_t per = (N + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  double* __restrict__ prow = part + ((size_t)blockIdx.y * M + mc) * M;
  double acc[BF_G][NP];
#pragma unroll
  for (int g = 0; g < BF_G; ++g)
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[g][p] = 0.0;
  for (int64_t rt = r0; rt < r1; rt += BF_RT) {
    __syncthreads();                                            // the workgroup's waves stay on one tile (and etab is ready)
    if (!has_task) continue;
    const int64_t re = (rt + BF_RT < r1) ? rt + BF_RT : r1;
    // Everything of the tile that is not the dm rows goes through a wave-private LDS strip, loaded once per tile with all
    // loads in flight together: hl[r][lane] = g0[n] + H[n, m] (own inducing row) and hx[r][c] = H[n, col_base + c] (the task's
    // CG inducing columns, read back as broadcasts).  As scalar loads the H[n, m'] values took one 64-byte line per 16 bytes
    // used and, with eight waves per workgroup, pushed the dm rows out of the 16 KB scalar cache; a one-row-ahead register
    // prefetch of H[n, m] is folded back into load-then-wait by the compiler.
    {
      double hv[BF_RT], hc[BF_RT / 4];
      const int rr4 = lane / CG4, cc = lane - rr4 * CG4;         // CG4 = 16 lanes per row of hx
#pragma unroll
      for (int r = 0; r < BF_RT; ++r) {
        const int64_t n = (rt + r < re) ? rt + r : re - 1;
        hv[r] = H[n * M + mc] + g0[n];
      }
#pragma unroll
      for (int k = 0; k < BF_RT / 4; ++k) {
        const int64_t n = (rt + 4 * k + rr4 < re) ? rt + 4 * k + rr4 : re - 1;
        const int col = (col_base + cc < M) ? col_base + cc : M - 1;
        hc[k] = H[n * M + col];
      }
#pragma unroll
      for (int r = 0; r < BF_RT; ++r) hl[r * 64 + lane] = hv[r];
#pragma unroll
      for (int k = 0; k < BF_RT / 4; ++k) hx[(4 * k + rr4) * CG4 + cc] = hc[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int g = 0; g < BF_G; ++g) {                            // unrolled: the accumulators of all groups stay in registers
      const int col0 = col_base + NP * g;
      if (col0 >= M) break;
      int mp[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) mp[p] = (col0 + p < M) ? col0 + p : M - 1;
      double zz[NP][QP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int q = 0; q < QP; ++q) zz[p][q] = (STRIP ? zl[q * 64] : Zc[(size_t)mc * QP + q]) * Zc[(size_t)mp[p] * QP + q];
      }
      const int nr = (int)(re - rt);
      for (int r = 0; r < nr; ++r) {
        const int64_t n = rt + r;
        const double* __restrict__ dn = dm + n * QP;            // wave-uniform: scalar loads
        const double hcur = hl[r * 64 + lane];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          double c0 = hcur + hx[r * CG4 + NP * g + p], c1 = 0.0;
#pragma unroll
          for (int q = 0; q < QP; q += 2) {
            c0 += zz[p][q] * dn[q];
            c1 += zz[p][q + 1] * dn[q + 1];
          }
          acc[g][p] += exp_tab(c0 + c1, etab);
        }
      }
    }
  }
  if (act) {
#pragma unroll
    for (int g = 0; g < BF_G; ++g)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (col_base + NP * g + p < M) prow[col_base + NP * g + p] = acc[g][p];
  }
}

// psi2[i][j] = sum over chunks of part[chunk][min(i,j)][max(i,j)]   (fixed order)
__global__ void __launch_bounds__(256) big_psi2_unpack_kernel(int M, int nchunk, const double* __restrict__ part,
                                                               double* __restrict__ psi2) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)M * M) return;
  const int i = (int)(e / M), j = (int)(e % M);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += part[((size_t)c * M + lo) * M + hi];
  psi2[e] = s;
}

static rgp_status launch_big_psi2_fwd(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, const ZTerms& zt,
                                      const RowTerms& rt, const double* dm, const double* H, double* part, double* psi2) {
  const dim3 grid((unsigned)((p.fwd_tasks + 7) / 8), (unsigned)p.fwd_chunks);
  const size_t lds = (PSI2_EXP_TAB + (size_t)8 * BF_RT * (64 + 16) + (zt.QP <= 48 ? (size_t)3 * zt.QP * 64 : 0)) * sizeof(double);
  switch (zt.QP) {
#define B2F(qp, np)                                                                                                        \
  case qp: {                                                                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)big_psi2_fwd_kernel<qp, np>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 forward (blocked) LDS attr: ") + hipGetErrorString(e)); \
    hipLaunchKernelGGL((big_psi2_fwd_kernel<qp, np>), grid, dim3(512), lds, s, N, M, p.fwd_tasks, zt.Zc, dm, H, rt.g0, part);  \
  } break;
    B2F(8, BF_NP) B2F(16, BF_NP) B2F(20, BF_NP) B2F(24, BF_NP) B2F(32, BF_NP) B2F(40, BF_NP) B2F(48, BF_NP) B2F(56, 1) B2F(64, 1)
#undef B2F
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2 (blocked): unsupported padded Q");
  }
  hipLaunchKernelGGL(big_psi2_unpack_kernel, dim3(big_nb256((int64_t)M * M)), dim3(256), 0, s, M, p.fwd_chunks, part, psi2);
  return RGP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// psi2 backward for symmetric dLs (weights W_n = dLs o psi2n).  Workgroup (row chunk, inducing block I): NT teams of TS threads
// (whole waves), team k works on data row rr + k, thread = inducing row m of block I, sweep over all m' = 0..M-1:
//   v = sum_m' W,  y[q] = sum_m' W z[m',q]
// then   dZ[m,q] += d2 (v (2 mu - z) - y)            (LDS slab [QP][TS], teams one after the other: fixed order)
//        RP[I][n] = [ sum_m v z | 1/2 sum_m z (v z + y) | sum_m v ]       (sums over the block's rows; combined over blocks
//                                                                          by big_rowgrad_kernel)
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
static __device__ __forceinline__ void big_pair(const double* __restrict__ zs, const double (&dzl)[QP], double (&y)[QP], double& v,
                                                const double base, const double hbm, const double dlv,
                                                const double* __restrict__ etab) {
  double c0 = base + hbm, c1 = 0.0;
#pragma unroll
  for (int q = 0; q < QP; q += 2) {
    c0 += dzl[q] * zs[q];
    c1 += dzl[q + 1] * zs[q + 1];
  }
  const double p = exp_tab(c0 + c1, etab);
  const double w = p * dlv;
  v += w;
#pragma unroll
  for (int q = 0; q < QP; ++q) y[q] += w * zs[q];
}

#define BIG_T 512
template <int QP>
__global__ void __launch_bounds__(BIG_T, 2)
big_psi2_bwd_kernel(const int64_t N, const int M, const int IB, const int TS, const int NT, const int Mp, const int Mld,
                    const double* __restrict__ Zc, const double* __restrict__ lam_g, const double* __restrict__ dLs_g,
                    const double* __restrict__ Xmu_g, const double* __restrict__ d2_g, const double* __restrict__ g0_g,
                    const double* __restrict__ H_g, double* __restrict__ RP, double* __restrict__ dz_part) {
  extern __shared__ double lds[];
  constexpr int RD = 2 * QP + 2;                           // staged row: d2[QP] | mu[QP] | g0 | pad
  constexpr int RW = 2 * QP + 2;                           // row sums:   a1[QP] | a23[QP] | w | pad
  double* dzacc = lds;                                     // [QP][TS]
  double* hb = dzacc + (size_t)QP * TS;                    // [NT][Mp]   h[n, 0..M) of each team's row
  double* rowd = hb + (size_t)NT * Mp;                     // [NT][RD]
  double* res = rowd + (size_t)NT * RD;                    // [8 waves][RW]
  double* etab = res + 8 * RW;                             // [256]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wpt = TS >> 6;                                 // waves per team
  const int team_raw = wave / wpt;                         // wave-uniform
  const bool in_team = team_raw < NT;
  const int team = in_team ? team_raw : NT - 1;
  const int mloc = t - team_raw * TS;
  const int i0 = (int)blockIdx.y * IB;
  const bool act = in_team && mloc < IB && i0 + mloc < M;
  const int mc = act ? i0 + mloc : M - 1;
  const double* __restrict__ zrowp = Zc + (size_t)mc * QP;
  for (int i = t; i < QP * TS; i += BIG_T) dzacc[i] = 0.0;
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + NT - 1) / NT * NT;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  const double* __restrict__ dlrow = dLs_g + mc;           // dLs[m'][m]: coalesced over the lanes' rows
  const double* hbt = hb + (size_t)team * Mp;
  const double* rd = rowd + (size_t)team * RD;

  for (int64_t rr = r0; rr < r1; rr += NT) {
    __syncthreads();                                       // the previous iteration's readers of hb / rowd / res are done
    for (int e = t; e < NT * Mp; e += BIG_T) {
      const int tk = e / Mp, j = e - tk * Mp;
      const int64_t nk = (rr + tk < r1) ? rr + tk : r1 - 1;
      hb[e] = (j < M) ? H_g[nk * M + j] : 0.0;
    }
    for (int e = t; e < NT * RD; e += BIG_T) {
      const int tk = e / RD, j = e - tk * RD;
      const int64_t nk = (rr + tk < r1) ? rr + tk : r1 - 1;
      double x = 0.0;
      if (j < QP) x = d2_g[nk * QP + j];
      else if (j < 2 * QP) x = Xmu_g[nk * QP + (j - QP)];
      else if (j == 2 * QP) x = g0_g[nk];
      rowd[e] = x;
    }
    __syncthreads();
    asm volatile("" ::: "memory");                         // keep the per-row reloads of z[m,:] out of the registers
    const bool valid = in_team && rr + team < r1;
    const int64_t n = valid ? rr + team : r1 - 1;
    double dzl[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) dzl[q] = -0.5 * zrowp[q] * (rd[q] - lam_g[q]);
    const double base = (act && valid) ? (rd[2 * QP] + hbt[mc]) : -INFINITY;       // exp(-inf) = 0: idle lanes add nothing
    double y[QP];
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < QP; ++q) y[q] = 0.0;
    {
      const int last = M - 1;
      double dlv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) dlv[u] = dlrow[(size_t)((u < M) ? u : last) * M];
      for (int mp = 0; mp < M; mp += 4) {
#if BIG_BWD_SYNC_EVERY > 0
        // The eight waves of the workgroup read the same z[m',:] rows as scalar operands.  Left alone they drift apart by more
        // rows than the 16 KB scalar cache holds (Z is 160 KB at M = 500) and each wave fetches its rows from L2 again;
        // meeting every few rows keeps them within one cache-full of each other.
        if ((mp & (BIG_BWD_SYNC_EVERY - 1)) == 0) __syncthreads();
#endif
#if BIG_BWD_PRIO_ALT > 0
        // the two waves of a SIMD (w and w + 4) take the higher issue priority in turns of BIG_BWD_PRIO_ALT columns
        if ((mp & (BIG_BWD_PRIO_ALT - 1)) == 0) {
          if ((((mp / BIG_BWD_PRIO_ALT) ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
#endif
        double dlc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          dlc[u] = dlv[u];
          dlv[u] = dlrow[(size_t)((mp + 4 + u < M) ? mp + 4 + u : last) * M];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int mq = mp + u;
          if (u == 0 || mq < M) big_pair<QP>(Zc + (size_t)mq * QP, dzl, y, v, base, hbt[mq], dlc[u], etab);
        }
      }
    }
    asm volatile("" ::: "memory");
    // sums over the wave's inducing rows (fixed order), one slot per wave
    double* wres = res + (size_t)wave * RW;
    {
      const double ws = wave_sum_dpp(v);
      if (lane == 0) wres[2 * QP] = ws;
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const double z = zrowp[q];
        const double s1 = wave_sum_dpp(v * z);
        const double s2 = wave_sum_dpp(0.5 * z * (v * z + y[q]));
        if (lane == 0) { wres[q] = s1; wres[QP + q] = s2; }
      }
    }
    // dZ accumulation  d2 (v (2 mu - z) - y), one team after the other
#pragma unroll
    for (int q = 0; q < QP; ++q) y[q] = rd[q] * (v * (2.0 * rd[QP + q] - zrowp[q]) - y[q]);
    for (int k = 0; k < NT; ++k) {
      if (act && team_raw == k) {
#pragma unroll
        for (int q = 0; q < QP; ++q) dzacc[q * TS + mloc] += y[q];
      }
      __syncthreads();
    }
    // row sums of this inducing block: the team's waves in order
    if (valid) {
      for (int j = mloc; j < RW; j += TS) {
        double sacc = 0.0;
        for (int wv = team * wpt; wv < (team + 1) * wpt; ++wv) sacc += res[(size_t)wv * RW + j];
        RP[((size_t)blockIdx.y * N + n) * RW + j] = sacc;
      }
    }
  }
  __syncthreads();
  for (int i = t; i < QP * TS; i += BIG_T) {
    const int q = i / TS, ml = i - q * TS;
    if (ml < IB && i0 + ml < M) dz_part[((size_t)blockIdx.x * QP + q) * Mld + i0 + ml] = dzacc[i];
  }
}

static size_t big_bwd_lds(const BigPlan& p) {
  return ((size_t)p.QP * p.TS + (size_t)p.NT * p.Mp + (size_t)p.NT * (2 * p.QP + 2) + 8 * (size_t)(2 * p.QP + 2) + PSI2_EXP_TAB) * sizeof(double);
}

template <int QP>
static hipError_t big_bwd_launch(hipStream_t s, const BigPlan& p, int64_t N, int M, const ZTerms& zt, const RowTerms& rt,
                                 const double* dLs, const double* H, double* RP, double* dz_part) {
  const size_t lds = big_bwd_lds(p);
  hipError_t e = hipFuncSetAttribute((const void*)big_psi2_bwd_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((big_psi2_bwd_kernel<QP>), dim3(p.bwd_grid, p.nI), dim3(BIG_T), lds, s, N, M, p.IB, p.TS, p.NT, p.Mp, p.Mld,
                     zt.Zc, zt.lam, dLs, rt.Xmu, rt.d2, rt.g0, H, RP, dz_part);
  return hipSuccess;
}

// Row gradients of the psi2 term from the per-block row sums (same expressions as the fused kernel's epilogue):
//   dmu = 2 d (a1 - mu w);  dd = -mu^2 w + 2 mu a1 - a23 + w / (2 d);  dS = -2 d^2 dd;  dinv_l += dd (d/lam)^2 - w / (2 lam)
__global__ void __launch_bounds__(256) big_rowgrad_kernel(int64_t N, int Q, int QP, int nI, const double* __restrict__ RP,
                                                           const double* __restrict__ Xmu, const double* __restrict__ d2,
                                                           const double* __restrict__ lam, double* __restrict__ gmu,
                                                           double* __restrict__ gS, int accumulate,
                                                           double* __restrict__ dlam_part, double* __restrict__ wsum_part) {
  __shared__ double sm[4][64];
  __shared__ double sl[4];
  const int RW = 2 * QP + 2;
  const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
  double dl = 0.0, wsum = 0.0;
  const double il = (q < Q) ? 1.0 / lam[q] : 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 4 + g; n < N; n += (int64_t)gridDim.x * 4) {
    if (q < Q) {
      double w = 0.0, a1 = 0.0, a23 = 0.0;
      for (int b = 0; b < nI; ++b) {
        const double* rp = RP + ((size_t)b * N + n) * RW;
        w += rp[2 * QP];
        a1 += rp[q];
        a23 += rp[QP + q];
      }
      const double mu = Xmu[n * QP + q], dq = d2[n * QP + q];
      const double dmu = 2.0 * dq * (a1 - mu * w);
      const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
      const double dS = -2.0 * dq * dq * dd;
      if (accumulate) { gmu[n * QP + q] += dmu; gS[n * QP + q] += dS; }
      else { gmu[n * QP + q] = dmu; gS[n * QP + q] = dS; }
      dl += dd * (dq * il) * (dq * il) - 0.5 * w * il;
      if (q == 0) wsum += w;
    } else if (q < QP && !accumulate) {
      gmu[n * QP + q] = 0.0;
      gS[n * QP + q] = 0.0;
    }
  }
  sm[g][q] = dl;
  if (q == 0) sl[g] = wsum;
  __syncthreads();
  if (g == 0) {
    if (q < QP) dlam_part[(size_t)blockIdx.x * QP + q] = (sm[0][q] + sm[1][q]) + (sm[2][q] + sm[3][q]);
    if (q == 0) wsum_part[blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
  }
}

// backward pass + its reductions: row gradients into gmu / gS, dz2 [QP][Mld], dlam2 [QP], wsum2 [1]
static rgp_status launch_big_psi2_bwd(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, int Q, const ZTerms& zt,
                                      const RowTerms& rt, const double* dLs, const double* H, BigWs& L, int accumulate,
                                      bool profile = false) {
  hipError_t e = hipSuccess;
  if (profile) prof_begin(h, (double)N * M * M * Q);        // rgp_get_profile times the sweep kernel alone
  switch (zt.QP) {
#define B2B(qp) case qp: e = big_bwd_launch<qp>(s, p, N, M, zt, rt, dLs, H, L.RP, L.dz2_part); break;
    B2B(8) B2B(16) B2B(20) B2B(24) B2B(32) B2B(40) B2B(48) B2B(56) B2B(64)
#undef B2B
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2 (blocked): unsupported padded Q");
  }
  if (profile) prof_end(h);
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 backward (blocked): ") + hipGetErrorString(e));
  hipLaunchKernelGGL(big_rowgrad_kernel, dim3(p.rows_blk), dim3(256), 0, s, N, Q, p.QP, p.nI, L.RP, rt.Xmu, rt.d2, zt.lam, L.gmu,
                     L.gS, accumulate, L.dlam2_part, L.wsum2_part);
  const ReduceSeg segs[3] = {{L.dz2_part, L.dz2, p.bwd_grid, (int64_t)p.QP * p.Mld, (int64_t)p.QP * p.Mld},
                             {L.dlam2_part, L.dlam2, p.rows_blk, p.QP, p.QP},
                             {L.wsum2_part, L.wsum2, p.rows_blk, 1, 1}};
  launch_reduce_batch(s, 3, segs);
  return RGP_OK;
}

// psi1 backward for a given Lp = dL_dpsi1 o psi1 (N x M): row gradients (=), dz1, dlam1, lsum1
static void launch_big_psi1_bwd(hipStream_t s, const BigPlan& p, int64_t N, int M, int Q, const ZTerms& zt, const RowTerms& rt,
                                BigWs& L) {
  const int QP = p.QP;
  {                                                          // [s1 | s2 | s0] = Lp [Z | Z^2 | 1]
    GemmArgs g;
    g.m = (int)N; g.n = p.W; g.k = M; g.A = L.Lp; g.lda = M; g.B = L.ZZ; g.ldb = p.W; g.C = L.S12; g.ldc = p.W;
    launch_gemm(s, g);
  }
  hipLaunchKernelGGL(big_rows1_kernel, dim3(p.rows_blk, (QP + 63) / 64), dim3(256), 0, s, N, Q, QP, p.W, L.S12, rt.Xmu, rt.e1, zt.lam, L.gmu, L.gS,
                     L.dlam1_part, L.lsum1_part);
  for (int k = 0; k < 2; ++k) {                              // A1 = Lp^T (e mu), A2 = Lp^T e   (sums over the data rows: split K)
    GemmArgs g;
    g.m = M; g.n = QP; g.k = (int)N; g.A = L.Lp; g.lda = M; g.ta = 1; g.B = k == 0 ? rt.em : rt.e1; g.ldb = QP;
    g.C = L.dzA_part; g.ldc = QP; g.splits = p.sk; g.split_stride = (int64_t)M * QP;
    if (p.sk == 1) g.C = k == 0 ? L.dzA1 : L.dzA2;
    launch_gemm(s, g);
    if (p.sk > 1) launch_reduce(s, p.sk, (int64_t)M * QP, (int64_t)M * QP, L.dzA_part, k == 0 ? L.dzA1 : L.dzA2, 1.0, 0);
  }
  hipLaunchKernelGGL(big_dz1_kernel, dim3(big_nb256((int64_t)M * QP)), dim3(256), 0, s, (int64_t)M * QP, L.dzA1, L.dzA2, zt.Zc, L.dz1);
  const ReduceSeg segs[2] = {{L.dlam1_part, L.dlam1, p.rows_blk, QP, QP}, {L.lsum1_part, L.lsum1, p.rows_blk, 1, 1}};
  launch_reduce_batch(s, 2, segs);
}

// ------------------------------------------------------------------------------------------------------------------
// psi2 forward sum on the matrix pipe, over the unordered pairs (a <= b) of inducing rows (psi2 is symmetric):
//   psi2[a,b] = sum_n exp(g0[n] + h[n,a] + h[n,b] + sum_q dm[n,q] z[a,q] z[b,q])
// Workgroup = a 32 x 32 block of (a, b) with block(b) >= block(a), times a range of data rows.  Only the 64 inducing rows of
// the block are touched, so per block of 64 data rows the operands in LDS are small (dm rows, g0, 2 x 32 columns of h) against
// 64 x 1024 exponents.  A wave owns 8 sub-tiles of 16 pairs (one a, 16 consecutive b); for a sub-tile the products
// z[a,q] z[b,q] are the B operand of v_mfma_f64_16x16x4 (built once per row block from the z rows in LDS), the A operand is the
// dm tile, and QP / 4 matrix instructions turn 16 rows x 16 pairs into exponents.  The exponentials (table form) and the sums
// over the rows stay in the lane (row = (lane >> 4) + 4 r, b = lane & 15).  Partial sums per (row split, a, b) are summed in
// fixed order and mirrored by bfm_finish_kernel.
// ------------------------------------------------------------------------------------------------------------------
#define BFM_R 64            // data rows per block
#define BFM_B 32            // inducing rows per side of a pair block
template <int KS>
__global__ void __launch_bounds__(512, 2) big_psi2_fwd_mfma_kernel(const int64_t N, const int M, const double* __restrict__ Zc,
                                                                    const double* __restrict__ dm, const double* __restrict__ H,
                                                                    const double* __restrict__ g0, double* __restrict__ part) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int QP = 4 * KS, LD = QP + 1, HLD = BFM_B + 1;
  __shared__ double etab[PSI2_EXP_TAB];
  __shared__ double Za[BFM_B * LD], Zb[BFM_B * LD];
  __shared__ double dmL[BFM_R * LD];
  __shared__ double g0L[BFM_R];
  __shared__ double HA[BFM_R * HLD], HB[BFM_R * HLD];
  const int ab = blockIdx.x, bb = blockIdx.y;
  if (bb < ab) return;                                 // the lower blocks mirror the upper ones
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, c = lane & 15;
  const int a0 = ab * BFM_B, b0 = bb * BFM_B;
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  for (int i = t; i < BFM_B * QP; i += 512) {
    const int r = i / QP, q = i - r * QP;
    Za[r * LD + q] = (a0 + r < M) ? Zc[(size_t)(a0 + r) * QP + q] : 0.0;
    Zb[r * LD + q] = (b0 + r < M) ? Zc[(size_t)(b0 + r) * QP + q] : 0.0;
  }
  const int64_t per = ((N + gridDim.z - 1) / gridDim.z + BFM_R - 1) / BFM_R * BFM_R;
  const int64_t r0 = (int64_t)blockIdx.z * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  double psum[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) psum[u] = 0.0;
  for (int64_t rb = r0; rb < r1; rb += BFM_R) {
    __syncthreads();                                   // the previous block's operand reads are done (first pass: Za, Zb, etab written)
    for (int i = t; i < BFM_R * QP; i += 512) {
      const int rr = i / QP, q = i - rr * QP;
      dmL[rr * LD + q] = (rb + rr < r1) ? dm[(size_t)(rb + rr) * QP + q] : 0.0;
    }
    for (int i = t; i < BFM_R * BFM_B; i += 512) {
      const int rr = i >> 5, jx = i & 31;
      const bool in = rb + rr < r1;
      HA[rr * HLD + jx] = (in && a0 + jx < M) ? H[(size_t)(rb + rr) * M + a0 + jx] : 0.0;
      HB[rr * HLD + jx] = (in && b0 + jx < M) ? H[(size_t)(rb + rr) * M + b0 + jx] : 0.0;
    }
    if (t < BFM_R) g0L[t] = (rb + t < r1) ? g0[rb + t] : 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {                      // the wave's sub-tiles: a = a0 + 4 wave + (u >> 1), b = b0 + 16 (u & 1) + c
      asm volatile("" ::: "memory");                   // keep each sub-tile's operand reads with its own instructions
      const int al = 4 * wave + (u >> 1), bl = 16 * (u & 1) + c;
      const bool valid = (a0 + al < M) && (b0 + bl < M) && (a0 + al <= b0 + bl);
      double zb[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) zb[k] = Za[al * LD + 4 * k + g] * Zb[bl * LD + 4 * k + g];
#pragma unroll
      for (int rs = 0; rs < BFM_R / 16; ++rs) {
        asm volatile("" ::: "memory");
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KS; ++k)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dmL[(rs * 16 + c) * LD + 4 * k + g], zb[k], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = rs * 16 + g + 4 * r;
          const double e = exp_tab(acc[r] + g0L[nl] + HA[nl * HLD + al] + HB[nl * HLD + bl], etab);
          psum[u] += (valid && rb + nl < r1) ? e : 0.0;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    double v = psum[u];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const int al = 4 * wave + (u >> 1), bl = 16 * (u & 1) + c;
    if (g == 0 && a0 + al < M && b0 + bl < M && a0 + al <= b0 + bl)
      part[(size_t)blockIdx.z * M * M + (size_t)(a0 + al) * M + (b0 + bl)] = v;
  }
}

// psi2[a,b] = psi2[b,a] = sum over the row splits of part[split][a,b]   (a <= b; fixed order)
__global__ void __launch_bounds__(256) bfm_finish_kernel(const int M, const int nsplit, const double* __restrict__ part, double* __restrict__ psi2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * M) return;
  const int a = i / M, b = i % M;
  if (b < a) return;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * M * M + i];
  psi2[i] = s;
  psi2[(size_t)b * M + a] = s;
}

static rgp_status launch_big_psi2_fwd_mfma(rgp_handle h, hipStream_t s, int64_t N, int M, const ZTerms& zt, const RowTerms& rt,
                                           const double* dm, const double* H, double* part, int rsplit, double* psi2) {
  const int nb = (M + BFM_B - 1) / BFM_B;
  const dim3 grid((unsigned)nb, (unsigned)nb, (unsigned)rsplit);
  switch (zt.QP) {
#define BFM(qp) case qp: hipLaunchKernelGGL((big_psi2_fwd_mfma_kernel<qp / 4>), grid, dim3(512), 0, s, N, M, zt.Zc, dm, H, rt.g0, part); break;
    BFM(8) BFM(16) BFM(20) BFM(24) BFM(32) BFM(40) BFM(48) BFM(56) BFM(64)
#undef BFM
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2 forward (blocked): unsupported padded Q");
  }
  hipLaunchKernelGGL(bfm_finish_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, rsplit, part, psi2);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 forward (matrix pipe): ") + hipGetErrorString(e));
  return RGP_OK;
}

// psi2 forward sum of the blocked path: the register-accumulator VALU kernel over the column groups that reach the diagonal
// (default), or the matrix-pipe kernel over 32 x 32 blocks of unordered pairs (RGP_PSI2_FWD_MFMA=1, read per call).  Both visit
// about the same number of (row, pair) elements and fp64 MFMA has the VALU's multiply-add rate on this chip, so they take the
// same time (243 against 255 ms per Q = 40 launch at M = 500, N = 524 128); the VALU kernel stays the default.
static rgp_status big_psi2_forward(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, BigWs& L, double* psi2) {
  if (getenv("RGP_PSI2_FWD_MFMA") == nullptr) return launch_big_psi2_fwd(h, s, p, N, M, L.zt, L.rt, L.dm, L.H, L.p2_part, psi2);
  return launch_big_psi2_fwd_mfma(h, s, N, M, L.zt, L.rt, L.dm, L.H, L.fp2_part, L.frs, psi2);
}

// The psi2 term of the SVI bound in the pair form (rgp_psi2_pair.hip, blocked form): forward sum and backward in one pass over
// the unordered pairs of inducing rows on the matrix pipe, then, from v = V2 / 2 and T alone (as for wide windows,
// rgp_wideq_impl.h):  [a1 | vz2 | w] = v [Z | Z^2 | 1],  B1 = v^T (d2 o mu),  B2 = v^T d2,  dZ2 = 2 B1 - Z o B2 - dZ3 and the row
// gradients.  Outputs as launch_big_psi2_fwd + launch_big_psi2_bwd: L.psi2, row gradients added to L.gmu / L.gS, L.dz2, L.dlam2,
// L.wsum2.
static rgp_status bp_psi2(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, int Q, BigWs& L, bool hq_ready) {
  const int QP = p.QP;
  PairBlkOut o{};
  o.psi2 = L.psi2; o.dz3 = L.pb_dz3;
  rgp_status st = launch_psi2_pairblk(h, s, p.pbg, N, L.zt, L.rt, hq_ready ? nullptr : L.H, L.dLs, L.pb_ws, o);
  if (st) return st;
  hipLaunchKernelGGL(wq_dmu_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, L.rt.d2, L.rt.Xmu, L.pb_DM);
  {
    GemmArgs g;
    g.m = (int)N; g.n = p.W; g.k = M; g.A = o.V2; g.lda = p.Mld; g.B = L.ZZ; g.ldb = p.W; g.C = L.pb_S22; g.ldc = p.W; g.alpha = 0.5;
    launch_gemm(s, g);
  }
  for (int k = 0; k < 2; ++k) {
    GemmArgs g;
    g.m = M; g.n = QP; g.k = (int)N; g.A = o.V2; g.lda = p.Mld; g.ta = 1; g.B = k == 0 ? L.pb_DM : L.rt.d2; g.ldb = QP; g.alpha = 0.5;
    g.C = L.pb_B_part; g.ldc = QP; g.splits = p.sk; g.split_stride = (int64_t)M * QP;
    if (p.sk == 1) g.C = k == 0 ? L.pb_B1 : L.pb_B2;
    launch_gemm(s, g);
    if (p.sk > 1) launch_reduce(s, p.sk, (int64_t)M * QP, (int64_t)M * QP, L.pb_B_part, k == 0 ? L.pb_B1 : L.pb_B2, 1.0, 0);
  }
  hipLaunchKernelGGL(wq_dz2_kernel, dim3(big_nb256((int64_t)M * QP)), dim3(256), 0, s, M, QP, p.Mld, L.pb_B1, L.pb_B2, L.pb_dz3, L.zt.Zc, L.dz2);
  hipLaunchKernelGGL(wq_rowgrad_kernel, dim3(p.rows_blk, (QP + 63) / 64), dim3(256), 0, s, N, Q, QP, p.W, L.pb_S22, o.T, L.rt.Xmu,
                     L.rt.d2, L.zt.lam, L.gmu, L.gS, L.dlam2_part, L.wsum2_part);
  const ReduceSeg segs[2] = {{L.dlam2_part, L.dlam2, p.rows_blk, QP, QP}, {L.wsum2_part, L.wsum2, p.rows_blk, 1, 1}};
  launch_reduce_batch(s, 2, segs);
  return RGP_OK;
}

static rgp_status big_check(rgp_handle h, int64_t N, int M, int Q, bool wide_ok = false) {
  if (N <= 0 || M <= 0 || Q <= 0) return rgp_fail(h, RGP_ERR_BAD_ARG, "N, M, Q must be positive");
  if (M > RGP_BIG_MAX_M) return rgp_fail(h, RGP_ERR_BAD_ARG, "M > 1024 is not supported by this build");
  if (rgp_qp(Q) > RGP_MAX_QP) {
    if (!wide_ok) return rgp_fail(h, RGP_ERR_BAD_ARG, "Q > 64 is supported by the layer step and free-run only");
    if (rgp_qp(Q) > WQ_MAX_QP) return rgp_fail(h, RGP_ERR_BAD_ARG, "Q > 1024 is not supported by this build");
  }
  if (N > 0x7fffffffLL / 2) return rgp_fail(h, RGP_ERR_BAD_ARG, "more than 2^30 data rows in one call");
  return RGP_OK;
}

// ==================================================================================================================
// unit entry points (rgp_psi_forward / rgp_psi_backward for M > 128)
// ==================================================================================================================
static rgp_status big_psi_forward(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S, const double* Z,
                                  const double* variance, const double* inv_l, double* psi1, double* psi2) {
  rgp_status st = big_check(h, N, M, Q, true);
  if (st) return st;
  const BigPlan p = big_plan(h, N, M, Q);
  BigWs L;
  const size_t need = big_carve(nullptr, p, N, M, Q, 1, 1, L, nullptr);
  if ((st = rgp_ensure_ws(h, need))) return st;
  big_carve(h->ws, p, N, M, Q, 1, 1, L, nullptr);
  hipStream_t s = h->stream;
  launch_prep_z(s, M, Q, p.QP, Z, variance, inv_l, RGP_JITTER, L.zt);
  launch_rowprep_dense(s, N, Q, mu, S, variance, L.zt, L.rt);
  if (L.wide) {                                              // windows wider than 64 columns: the GEMM-shaped forms
    hipLaunchKernelGGL(big_zz_kernel, dim3(big_nb256((int64_t)M * p.W)), dim3(256), 0, s, M, p.QP, p.W, L.zt.Zc, L.ZZ);
    wq_psi1_fwd(s, p, N, M, L, L.wq);
    if (psi1) RGP_HIP_CHECK(h, hipMemcpyAsync(psi1, L.P, (size_t)N * M * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (psi2) {
      hipLaunchKernelGGL(big_dm_kernel, dim3(big_nb256(N * p.QP)), dim3(256), 0, s, N * p.QP, p.QP, L.rt.d2, L.zt.lam, L.dm);
      if ((st = wq_psi2_forward(h, s, p, N, M, L, L.wq))) return st;
      RGP_HIP_CHECK(h, hipMemcpyAsync(psi2, L.psi2, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    return last_launch(h, "rgp_psi_forward (wide windows)");
  }
  launch_big_psi1_fwd(s, N, M, L.zt, L.rt, psi1, psi2 ? L.H : nullptr);
  if (psi2) {
    hipLaunchKernelGGL(big_dm_kernel, dim3(big_nb256(N * p.QP)), dim3(256), 0, s, N * p.QP, p.QP, L.rt.d2, L.zt.lam, L.dm);
    if ((st = big_psi2_forward(h, s, p, N, M, L, psi2))) return st;
  }
  return last_launch(h, "rgp_psi_forward (blocked)");
}

__global__ void __launch_bounds__(256) big_mul_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                       double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

static rgp_status big_psi_backward(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S, const double* Z,
                                   const double* variance, const double* inv_l, double dL_dpsi0_sum, const double* dL_dpsi1,
                                   const double* dL_dpsi2, double* dvar, double* dinv_l, double* dZ, double* dmu, double* dS) {
  rgp_status st = big_check(h, N, M, Q, true);
  if (st) return st;
  const BigPlan p = big_plan(h, N, M, Q);
  BigWs L;
  const size_t need = big_carve(nullptr, p, N, M, Q, 1, 1, L, nullptr);
  if ((st = rgp_ensure_ws(h, need))) return st;
  big_carve(h->ws, p, N, M, Q, 1, 1, L, nullptr);
  hipStream_t s = h->stream;
  const int QP = p.QP;
  launch_prep_z(s, M, Q, QP, Z, variance, inv_l, RGP_JITTER, L.zt);
  launch_rowprep_dense(s, N, Q, mu, S, variance, L.zt, L.rt);
  hipLaunchKernelGGL(big_zz_kernel, dim3(big_nb256((int64_t)M * p.W)), dim3(256), 0, s, M, QP, p.W, L.zt.Zc, L.ZZ);
  if (L.wide) wq_psi1_fwd(s, p, N, M, L, L.wq);
  const bool hq = !L.wide && p.pb && dL_dpsi2 && big_psi1_fwd_on_matrix_pipe();    // h goes straight into the pair form's block layout
  if (!L.wide) launch_big_psi1_fwd(s, N, M, L.zt, L.rt, dL_dpsi1 ? L.P : nullptr, (dL_dpsi2 && !hq) ? L.H : nullptr,
                                   hq ? psi2_pairblk_hq(p.pbg, L.pb_ws) : nullptr, p.num_cu);
  if (dL_dpsi1) {
    hipLaunchKernelGGL(big_mul_kernel, dim3(big_nb256(N * M)), dim3(256), 0, s, N * M, dL_dpsi1, L.P, L.Lp);
    launch_big_psi1_bwd(s, p, N, M, Q, L.zt, L.rt, L);
  } else {
    hipLaunchKernelGGL(big_fill_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, L.gmu, 0.0);
    hipLaunchKernelGGL(big_fill_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, L.gS, 0.0);
  }
  if (dL_dpsi2) {
    hipLaunchKernelGGL(symmetrise_kernel, dim3(big_nb256((int64_t)M * M)), dim3(256), 0, s, M, dL_dpsi2, L.dLs);
    hipLaunchKernelGGL(big_dm_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, QP, L.rt.d2, L.zt.lam, L.dm);
    if (L.wide) st = wq_psi2(h, s, p, N, M, Q, L, L.wq);
    else if (p.pb) st = bp_psi2(h, s, p, N, M, Q, L, hq);
    else if (!(st = big_psi2_forward(h, s, p, N, M, L, L.psi2)))
      st = launch_big_psi2_bwd(h, s, p, N, M, Q, L.zt, L.rt, L.dLs, L.H, L, 1);
    if (st) return st;
  }
  FinalizeArgs f{};
  f.M = M; f.Q = Q; f.QP = QP; f.Zc = L.zt.Zc; f.lam = L.zt.lam; f.variance = variance;
  f.Kj = nullptr; f.dKmm = nullptr;
  f.dLs = dL_dpsi2 ? L.dLs : nullptr; f.psi2 = dL_dpsi2 ? L.psi2 : nullptr;
  f.dz1 = dL_dpsi1 ? L.dz1 : nullptr; f.dz2 = dL_dpsi2 ? L.dz2 : nullptr; f.dz2_ld = p.Mld;
  f.dlam1 = dL_dpsi1 ? L.dlam1 : nullptr; f.dlam2 = dL_dpsi2 ? L.dlam2 : nullptr;
  f.lsum1 = dL_dpsi1 ? L.lsum1 : nullptr; f.wsum2 = dL_dpsi2 ? L.wsum2 : nullptr;
  f.dk_scale = 1.0; f.psi0_coef = dL_dpsi0_sum; f.psi0_dev = nullptr; f.psi0_dev_scale = 0.0;
  f.gZ = dZ; f.ginv_l = dinv_l; f.gvariance = dvar;
  launch_finalize(s, f, L.fin_dlrow, L.fin_gsum);
  const int64_t nq = N * Q;
  hipLaunchKernelGGL(unpad_kernel, dim3(big_nb256(nq)), dim3(256), 0, s, N, Q, QP, L.gmu, dmu);
  hipLaunchKernelGGL(unpad_kernel, dim3(big_nb256(nq)), dim3(256), 0, s, N, Q, QP, L.gS, dS);
  return last_launch(h, "rgp_psi_backward (blocked)");
}

// ==================================================================================================================
// the SVI layer step (rgp_layer_step, M > 128): same inputs, outputs and arithmetic as the M <= 128 step
// ==================================================================================================================
static rgp_status big_svi_layer_step(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out) {
  const int M = in->M, D = in->D;
  const int Q = in->X_win * in->X_dim + in->U_win * in->U_dim;
  const int64_t N = in->N;
  rgp_status st = big_check(h, N, M, Q, true);
  if (st) return st;
  if (in->y_uncertain && (in->X_dim != D || !in->own_var)) return rgp_fail(h, RGP_ERR_BAD_ARG, "hidden layer needs X_dim == D and own_var");
  if (!in->y_uncertain && in->X_win != 0) return rgp_fail(h, RGP_ERR_BAD_ARG, "observed layer must have X_win == 0 (as in the reference)");
  if (D < 1 || D > 2048) return rgp_fail(h, RGP_ERR_BAD_ARG, "D must be between 1 and 2048");
  const BigPlan p = big_plan(h, N, M, Q);
  const int QP = p.QP;
  BigWs L;
  SviChain c{};
  c.N = N; c.M = M; c.D = D; c.qU_ratio = in->qU_ratio;
  const size_t need = big_carve(nullptr, p, N, M, Q, D, in->n_seq, L, &c);
  if ((st = rgp_ensure_ws(h, need))) return st;                 // also joins the side streams (parameter updates issued there)
  big_carve(h->ws, p, N, M, Q, D, in->n_seq, L, &c);
  hipStream_t s = h->stream;

  WindowSpec w{};
  w.n_seq = in->n_seq; w.X_win = in->X_win; w.X_dim = in->X_dim; w.U_win = in->U_win; w.U_dim = in->U_dim; w.D = D;
  w.y_uncertain = in->y_uncertain; w.own_off = in->own_off; w.own_len = in->own_len; w.up_off = in->up_off;
  w.up_len = in->up_len; w.row_off = L.row_off; w.own_mean = in->own_mean; w.own_var = in->own_var;
  w.up_mean = in->up_mean; w.up_var = in->up_var; w.up_var_const = in->up_var_const;

  // tables, Z terms, row terms
  hipLaunchKernelGGL(svi_tables_kernel, dim3(1), dim3(64), 0, s, in->n_seq, in->X_win, in->own_len,
                     in->U_win > 0 ? in->up_len : nullptr, L.row_off, L.own_pre, L.up_pre, N, D, in->variance, in->noise,
                     L.psi0sum, L.psi0coef);
  launch_prep_z(s, M, Q, QP, in->Z, in->variance, in->inv_l, RGP_JITTER, L.zt);
  hipLaunchKernelGGL(big_zz_kernel, dim3(big_nb256((int64_t)M * p.W)), dim3(256), 0, s, M, QP, p.W, L.zt.Zc, L.ZZ);
  launch_rowprep_window(s, N, Q, w, in->variance, L.zt, L.rt, L.Ym, in->y_uncertain ? L.Yv : nullptr, L.yry_part);
  launch_reduce(s, 2048, 1, 1, L.yry_part, L.yry, 1.0, 0);
  hipLaunchKernelGGL(big_dm_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, QP, L.rt.d2, L.zt.lam, L.dm);

  // data-independent algebra: K_uu factor, Cpsi1, dL/dpsi2, KL terms
  c.Kj = L.zt.Kuu; c.noise = in->noise; c.qU_m = in->qU_m; c.qU_W = in->qU_W; c.qU_a = in->qU_a; c.status = h->status;
  c.dLs_sym = L.dLs;
  c.psi0sum = L.psi0sum; c.psi1Y = L.psi1Y; c.psi2 = L.psi2; c.yry = L.yry;
  c.gqU_m = out->gqU_m; c.gqU_W = out->gqU_W; c.gqU_a = out->gqU_a; c.winv = out->woodbury_inv;
  if ((st = launch_svi_chain_pre(h, c, s))) return st;
  if (M <= RGP_MAX_M)       // the single-workgroup chain (M <= 128) does not leave the device scalars the products below use
    hipLaunchKernelGGL(wq_scal_kernel, dim3(1), dim3(64), 0, s, in->noise, D, c.scal);

  // psi1 side
  const bool hq = !L.wide && p.pb && big_psi1_fwd_on_matrix_pipe();      // h goes straight into the pair form's block layout
  if (L.wide)
    wq_psi1_fwd(s, p, N, M, L, L.wq);
  else
    launch_big_psi1_fwd(s, N, M, L.zt, L.rt, L.P, hq ? nullptr : L.H, hq ? psi2_pairblk_hq(p.pbg, L.pb_ws) : nullptr, p.num_cu);
  {                                                          // Lp = psi1 o (Ym Cpsi1^T)              (svi_vardtc.py:164-167)
    GemmArgs g;
    g.m = (int)N; g.n = M; g.k = D; g.A = L.Ym; g.lda = D; g.B = c.Cpsi1; g.ldb = D; g.tb = 1; g.C = L.Lp; g.ldc = M;
    g.E = L.P; g.lde = M;
    launch_gemm(s, g);
  }
  {                                                          // psi1Y = Ym^T psi1  (D x M)            (svi_vardtc.py:57)
    GemmArgs g;
    g.m = D; g.n = M; g.k = (int)N; g.A = L.Ym; g.lda = D; g.ta = 1; g.B = L.P; g.ldb = M;
    g.C = p.sk > 1 ? L.psi1Y_part : L.psi1Y; g.ldc = M; g.splits = p.sk; g.split_stride = (int64_t)D * M;
    launch_gemm(s, g);
    if (p.sk > 1) launch_reduce(s, p.sk, (int64_t)D * M, (int64_t)D * M, L.psi1Y_part, L.psi1Y, 1.0, 0);
  }
  if (in->y_uncertain) {                                     // dL/dYmean = psi1 Cpsi1 - beta Ym           (svi_vardtc.py:192)
    GemmArgs g;
    g.m = (int)N; g.n = D; g.k = M; g.A = L.P; g.lda = M; g.B = c.Cpsi1; g.ldb = D; g.C = L.dYm; g.ldc = D;
    g.gamma = 1.0; g.g_dev = c.scal + 5; g.Cin = L.Ym; g.ldcin = D;
    launch_gemm(s, g);
  }
  launch_big_psi1_bwd(s, p, N, M, Q, L.zt, L.rt, L);

  // psi2: forward sum, then backward with the symmetrised dL/dpsi2 of the chain
  if (L.wide) {
    prof_begin(h, (double)N * M * M * Q);                     // rgp_get_profile times the whole GEMM-shaped psi2 term
    st = wq_psi2(h, s, p, N, M, Q, L, L.wq);
    prof_end(h);
  } else if (p.pb) {
    prof_begin(h, (double)N * M * M * Q);                     // rgp_get_profile times the whole fused psi2 term
    st = bp_psi2(h, s, p, N, M, Q, L, hq);
    prof_end(h);
  } else {
    st = big_psi2_forward(h, s, p, N, M, L, L.psi2);
    if (!st) st = launch_big_psi2_bwd(h, s, p, N, M, Q, L.zt, L.rt, L.dLs, L.H, L, 1, true);
  }
  if (st) return st;

  // latent gradients
  int npart = 0;
  if (in->y_uncertain) {
    LatentArgs a{};
    a.w = w; a.QP = QP; a.own_rows = in->own_rows; a.own_pre = L.own_pre; a.gmu = L.gmu; a.gS = L.gS; a.dYm = L.dYm;
    a.noise = in->noise; a.dYvar_row = nullptr; a.g_own_mean = out->g_own_mean; a.g_own_var = out->g_own_var;
    a.elbo_part = L.elbo_part;
    hipLaunchKernelGGL(own_latent_kernel, dim3(LAT_BLOCKS), dim3(256), 0, s, a);
    npart = LAT_BLOCKS;
  }
  if (out->g_up_mean && in->U_win > 0) {
    UpArgs a{};
    a.w = w; a.QP = QP; a.up_rows = in->up_rows; a.up_pre = L.up_pre; a.gmu = L.gmu; a.gS = L.gS;
    a.g_up_mean = out->g_up_mean; a.g_up_var = out->g_up_var;
    hipLaunchKernelGGL(up_latent_kernel, dim3(LAT_BLOCKS), dim3(256), 0, s, a);
  }
  // data-dependent algebra + kernel-hyperparameter gradients
  if ((st = launch_svi_chain_post(h, c, s))) return st;
  {
    FinalizeArgs f{};
    f.M = M; f.Q = Q; f.QP = QP; f.Zc = L.zt.Zc; f.lam = L.zt.lam; f.variance = in->variance;
    f.Kj = L.zt.Kuu; f.dKmm = c.dKmm; f.dLs = L.dLs; f.psi2 = L.psi2; f.dz1 = L.dz1; f.dz2 = L.dz2; f.dz2_ld = p.Mld;
    f.dlam1 = L.dlam1; f.dlam2 = L.dlam2; f.lsum1 = L.lsum1; f.wsum2 = L.wsum2;
    f.dk_scale = 1.0; f.psi0_coef = 0.0; f.psi0_dev = L.psi0coef; f.psi0_dev_scale = 1.0;
    f.gZ = out->gZ; f.ginv_l = out->ginv_l; f.gvariance = out->gvariance;
    launch_finalize(s, f, L.fin_dlrow, L.fin_gsum);
  }
  hipLaunchKernelGGL(finish_elbo_kernel, dim3(1), dim3(64), 0, s, c.bound, L.elbo_part, npart, c.dthetaL, out->elbo,
                     out->gnoise, 1.0);
  if (out->woodbury_vector)
    RGP_HIP_CHECK(h, hipMemcpyAsync(out->woodbury_vector, c.Kinv_mu, (size_t)M * D * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (h->overlap) {
    // rgp_adam_step_post / the caller's all-reduce run on the side stream: order them behind this step
    RGP_HIP_CHECK(h, hipEventRecord(h->ev_stats, s));
    RGP_HIP_CHECK(h, hipStreamWaitEvent(h->side_post, h->ev_stats, 0));
  }
  return last_launch(h, "rgp_layer_step (blocked)");
}

// ==================================================================================================================
// collapsed-bound layer step for M > 128 (VarDTC.inference, vardtc.py:88-208): the blocked kernels around the blocked chain.
// dL/dpsi2 depends on psi2, so the forward kernel runs before the chain and the backward kernel after it.
// ==================================================================================================================
// T = r2[n] P[n,m]   (left operand of PsiR = P^T diag(r2) P)
__global__ void __launch_bounds__(256) big_vd_scale_rows_kernel(int64_t NM, int M, const double* __restrict__ r2,
                                                                 const double* __restrict__ P, double* __restrict__ T) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NM) T[i] = r2[i / M] * P[i];
}
// Row part of the bound (vardtc.py:177-205) for uncertain outputs, one wave per row, T2 = psi1 G2:
//   Lp[n,m] (= beta psi1 o (Ym v), already there) += beta^2 r2[n] psi1[n,m] T2[n,m];  dL_dYvar[n] = -beta/2 + beta^2/2 sum_m psi1 T2
__global__ void __launch_bounds__(256) big_vd_rows_kernel(int64_t N, int M, const double* __restrict__ r2, const double* __restrict__ P,
                                                           const double* __restrict__ T2, const double* __restrict__ beta_p,
                                                           double* __restrict__ Lp, double* __restrict__ dYvar) {
  const double beta = beta_p[0];
  const int lane = threadIdx.x & 63;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < N; r += (int64_t)gridDim.x * wpb) {
    const double rr = r2[r];
    double tp = 0.0;
    for (int m = lane; m < M; m += 64) {
      const double pt = P[r * M + m] * T2[r * M + m];
      tp += pt;
      Lp[r * M + m] += beta * beta * rr * pt;
    }
    tp = wave_sum(tp);
    if (lane == 0) dYvar[r] = -0.5 * beta + 0.5 * beta * beta * tp;
  }
}

// phase 0: whole step with local statistics; 1: statistics only, packed into stats_io as [psi0sum, yry, psi2 (M*M), psi1Y (D*M),
// PsiR (M*M)]; 2: finish with the (all-reduced) statistics in stats_io, N_global rows in total, replicated terms weighted by rep_w.
// Phase 2 relies on the row terms, psi1 and h that phase 1 of the same layer left in the handle's workspace.
static rgp_status big_vardtc_layer_step(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out, int phase, double* stats_io,
                                        int64_t N_global, double rep_w) {
  const int M = in->M, D = in->D;
  const int Q = in->X_win * in->X_dim + in->U_win * in->U_dim;
  const int64_t N = in->N;
  rgp_status st = big_check(h, N, M, Q, true);
  if (st) return st;
  if (in->y_uncertain && (in->X_dim != D || !in->own_var)) return rgp_fail(h, RGP_ERR_BAD_ARG, "hidden layer needs X_dim == D and own_var");
  if (!in->y_uncertain && in->X_win != 0) return rgp_fail(h, RGP_ERR_BAD_ARG, "observed layer must have X_win == 0 (as in the reference)");
  if (D < 1 || D > 2048) return rgp_fail(h, RGP_ERR_BAD_ARG, "D must be between 1 and 2048");
  const BigPlan p = big_plan(h, N, M, Q);
  const int QP = p.QP;
  BigWs L;
  VarDtcChain c{};
  c.N = N; c.M = M; c.D = D; c.y_uncertain = in->y_uncertain;
  const size_t need = big_carve(nullptr, p, N, M, Q, D, in->n_seq, L, nullptr, &c);
  if ((st = rgp_ensure_ws(h, need))) return st;
  big_carve(h->ws, p, N, M, Q, D, in->n_seq, L, nullptr, &c);
  hipStream_t s = h->stream;

  WindowSpec w{};
  w.n_seq = in->n_seq; w.X_win = in->X_win; w.X_dim = in->X_dim; w.U_win = in->U_win; w.U_dim = in->U_dim; w.D = D;
  w.y_uncertain = in->y_uncertain; w.own_off = in->own_off; w.own_len = in->own_len; w.up_off = in->up_off;
  w.up_len = in->up_len; w.row_off = L.row_off; w.own_mean = in->own_mean; w.own_var = in->own_var;
  w.up_mean = in->up_mean; w.up_var = in->up_var; w.up_var_const = in->up_var_const;

  const size_t mm = (size_t)M * M, dm_ = (size_t)D * M;
  if (phase != 2) {
  // tables, Z terms, row terms
  hipLaunchKernelGGL(svi_tables_kernel, dim3(1), dim3(64), 0, s, in->n_seq, in->X_win, in->own_len,
                     in->U_win > 0 ? in->up_len : nullptr, L.row_off, L.own_pre, L.up_pre, N, D, in->variance, in->noise,
                     L.psi0sum, L.psi0coef);
  launch_prep_z(s, M, Q, QP, in->Z, in->variance, in->inv_l, RGP_JITTER, L.zt);
  hipLaunchKernelGGL(big_zz_kernel, dim3(big_nb256((int64_t)M * p.W)), dim3(256), 0, s, M, QP, p.W, L.zt.Zc, L.ZZ);
  launch_rowprep_window(s, N, Q, w, in->variance, L.zt, L.rt, L.Ym, in->y_uncertain ? L.Yv : nullptr, L.yry_part);
  launch_reduce(s, 2048, 1, 1, L.yry_part, L.yry, 1.0, 0);
  hipLaunchKernelGGL(big_dm_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, QP, L.rt.d2, L.zt.lam, L.dm);

  // statistics (vardtc.py:54-86): psi1, psi1^T Y, psi2, and for uncertain outputs PsiR = psi1^T diag(r^2) psi1
  if (L.wide) wq_psi1_fwd(s, p, N, M, L, L.wq);
  else launch_big_psi1_fwd(s, N, M, L.zt, L.rt, L.P, L.H);
  {
    GemmArgs g;
    g.m = D; g.n = M; g.k = (int)N; g.A = L.Ym; g.lda = D; g.ta = 1; g.B = L.P; g.ldb = M;
    g.C = p.sk > 1 ? L.psi1Y_part : L.psi1Y; g.ldc = M; g.splits = p.sk; g.split_stride = (int64_t)D * M;
    launch_gemm(s, g);
    if (p.sk > 1) launch_reduce(s, p.sk, (int64_t)D * M, (int64_t)D * M, L.psi1Y_part, L.psi1Y, 1.0, 0);
  }
  if (in->y_uncertain) {
    hipLaunchKernelGGL(rowsum_kernel, dim3(big_nb256(N)), dim3(256), 0, s, N, D, L.Yv, L.r2);
    hipLaunchKernelGGL(big_vd_scale_rows_kernel, dim3(big_nb256(N * M)), dim3(256), 0, s, N * M, M, L.r2, L.P, L.T2);
    GemmArgs g;
    g.m = M; g.n = M; g.k = (int)N; g.A = L.T2; g.lda = M; g.ta = 1; g.B = L.P; g.ldb = M;
    g.C = p.sk > 1 ? L.PsiR_part : L.PsiR; g.ldc = M; g.splits = p.sk; g.split_stride = (int64_t)M * M;
    launch_gemm(s, g);
    if (p.sk > 1) launch_reduce(s, p.sk, (int64_t)M * M, (int64_t)M * M, L.PsiR_part, L.PsiR, 1.0, 0);
  }
  st = L.wide ? wq_psi2_forward(h, s, p, N, M, L, L.wq)
              : big_psi2_forward(h, s, p, N, M, L, L.psi2);
  if (st) return st;
  }
  if (phase == 1) {                                  // hand the local statistics to the caller (vardtc.py:65-67 stub)
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io, L.psi0sum, sizeof(double), hipMemcpyDeviceToDevice, s));
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 1, L.yry, sizeof(double), hipMemcpyDeviceToDevice, s));
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 2, L.psi2, mm * sizeof(double), hipMemcpyDeviceToDevice, s));
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 2 + mm, L.psi1Y, dm_ * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (in->y_uncertain) RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 2 + mm + dm_, L.PsiR, mm * sizeof(double), hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL(big_fill_kernel, dim3(big_nb256((int64_t)mm)), dim3(256), 0, s, (int64_t)mm, stats_io + 2 + mm + dm_, 0.0);
    return last_launch(h, "rgp_vardtc_stats (blocked)");
  }

  // M x M algebra
  c.Kj = L.zt.Kuu; c.noise = in->noise; c.psi0sum = L.psi0sum; c.psi1Y = L.psi1Y; c.psi2 = L.psi2; c.yry = L.yry;
  c.PsiR = L.PsiR; c.winv = out->woodbury_inv; c.status = h->status;
  if (phase == 2) {                                  // global statistics supplied by the caller
    c.N = N_global; c.psi0sum = stats_io; c.yry = stats_io + 1; c.psi2 = stats_io + 2; c.psi1Y = stats_io + 2 + mm;
    c.PsiR = stats_io + 2 + mm + dm_;
  }
  if ((st = launch_vardtc_chain(h, c))) return st;
  hipLaunchKernelGGL(psi0_coef_beta_kernel, dim3(1), dim3(64), 0, s, N, D, c.beta, L.psi0coef);

  // row gradients: Lp = psi1 o dL/dpsi1,  dL/dYmean = beta (psi1 v^T - Ym),  dL/dYvar
  {                                                          // Lp = beta psi1 o (Ym v)                     (vardtc.py:179-181)
    GemmArgs g;
    g.m = (int)N; g.n = M; g.k = D; g.A = L.Ym; g.lda = D; g.B = c.vT; g.ldb = D; g.tb = 1; g.C = L.Lp; g.ldc = M;
    g.a_dev = c.beta; g.E = L.P; g.lde = M;
    launch_gemm(s, g);
  }
  if (in->y_uncertain) {
    {                                                        // T2 = psi1 G2
      GemmArgs g;
      g.m = (int)N; g.n = M; g.k = M; g.A = L.P; g.lda = M; g.B = c.G2; g.ldb = M; g.C = L.T2; g.ldc = M;
      launch_gemm(s, g);
    }
    hipLaunchKernelGGL(big_vd_rows_kernel, dim3((unsigned)(N < 4096 ? (N + 3) / 4 : 1024)), dim3(256), 0, s, N, M, L.r2, L.P, L.T2,
                       c.beta, L.Lp, L.dYvar_row);
    {                                                        // dL/dYmean = beta psi1 v^T - beta Ym           (vardtc.py:204)
      GemmArgs g;
      g.m = (int)N; g.n = D; g.k = M; g.A = L.P; g.lda = M; g.B = c.vT; g.ldb = D; g.C = L.dYm; g.ldc = D;
      g.a_dev = c.beta; g.gamma = -1.0; g.g_dev = c.beta; g.Cin = L.Ym; g.ldcin = D;
      launch_gemm(s, g);
    }
  }
  launch_big_psi1_bwd(s, p, N, M, Q, L.zt, L.rt, L);
  hipLaunchKernelGGL(symmetrise_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, c.dLdpsi2, L.dLs);
  st = L.wide ? wq_psi2(h, s, p, N, M, Q, L, L.wq) : launch_big_psi2_bwd(h, s, p, N, M, Q, L.zt, L.rt, L.dLs, L.H, L, 1);
  if (st) return st;

  // latent gradients
  int npart = 0;
  if (in->y_uncertain) {
    LatentArgs a{};
    a.w = w; a.QP = QP; a.own_rows = in->own_rows; a.own_pre = L.own_pre; a.gmu = L.gmu; a.gS = L.gS; a.dYm = L.dYm;
    a.noise = in->noise; a.dYvar_row = L.dYvar_row; a.g_own_mean = out->g_own_mean; a.g_own_var = out->g_own_var;
    a.elbo_part = L.elbo_part;
    hipLaunchKernelGGL(own_latent_kernel, dim3(LAT_BLOCKS), dim3(256), 0, s, a);
    npart = LAT_BLOCKS;
  }
  if (out->g_up_mean && in->U_win > 0) {
    UpArgs a{};
    a.w = w; a.QP = QP; a.up_rows = in->up_rows; a.up_pre = L.up_pre; a.gmu = L.gmu; a.gS = L.gS;
    a.g_up_mean = out->g_up_mean; a.g_up_var = out->g_up_var;
    hipLaunchKernelGGL(up_latent_kernel, dim3(LAT_BLOCKS), dim3(256), 0, s, a);
  }
  {
    FinalizeArgs f{};
    f.M = M; f.Q = Q; f.QP = QP; f.Zc = L.zt.Zc; f.lam = L.zt.lam; f.variance = in->variance;
    f.Kj = L.zt.Kuu; f.dKmm = c.dKmm; f.dLs = L.dLs; f.psi2 = L.psi2; f.dz1 = L.dz1; f.dz2 = L.dz2; f.dz2_ld = p.Mld;
    f.dlam1 = L.dlam1; f.dlam2 = L.dlam2; f.lsum1 = L.lsum1; f.wsum2 = L.wsum2;
    f.dk_scale = rep_w; f.psi0_coef = 0.0; f.psi0_dev = L.psi0coef; f.psi0_dev_scale = 1.0;
    f.gZ = out->gZ; f.ginv_l = out->ginv_l; f.gvariance = out->gvariance;
    launch_finalize(s, f, L.fin_dlrow, L.fin_gsum);
  }
  hipLaunchKernelGGL(finish_elbo_kernel, dim3(1), dim3(64), 0, s, c.bound, L.elbo_part, npart, c.dthetaL, out->elbo,
                     out->gnoise, rep_w);
  if (out->woodbury_vector)
    RGP_HIP_CHECK(h, hipMemcpyAsync(out->woodbury_vector, c.vT, (size_t)M * D * sizeof(double), hipMemcpyDeviceToDevice, s));
  return last_launch(h, "rgp_layer_step (blocked, collapsed bound)");
}
