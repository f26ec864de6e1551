// rgp_psi2_pair.hip -- the fused psi2 pass (forward sum + backward, SVI bound) on the fp64 matrix pipe of gfx950, written over
// the P = M (M + 1) / 2 UNORDERED pairs of inducing rows (psi2n[n, m, m'] and its weights are symmetric in m, m').
//
// What it computes (GPy's rbf_psi_comp closed forms as restated in rgp_psi.hip; called for kern.psi2 at
// autoreg/inference/svi_vardtc.py:48-50 and for the three backward contractions at autoreg/layers.py:98-132,1452-1458):
//   E[n,p]  = h'[n,a] + h'[n,b] + k_p - 1/2 sum_q d2[n,q] z[a,q] z[b,q]        p = (a, b)
//             h'[n,m] = sum_q d2 z (mu - z/4) - 1/4 sum_q lam z^2 + g0[n]/2,    k_p = 1/2 sum_q lam_q z[a,q] z[b,q]
//   e       = exp(E);   psi2[a,b] = psi2[b,a] = sum_n e;   w = c_p dLs[a,b] e   (c_p = 2 off the diagonal, 1 on it)
//   V2[n,m] = sum_{p contains m} w  (a diagonal pair adds twice)  = 2 v[n,m],   v[n,m] = sum_m' dLs[m,m'] psi2n[n,m,m']
//   T[n,q]  = sum_p w[n,p] z[a,q] z[b,q]                                        (= sum_{m,m'} W z z')
//   C[p,q]  = sum_n w[n,p] d2[n,q]
// and from those, outside the main kernel: [a1 | vz2 | ws] = v [Z | Z^2 | 1], the row gradients (same expressions as the sweep
// kernels' epilogue), B1 = v^T (d2 o mu), B2 = v^T d2, dZ3[m,q] = sum_m' z[m',q] C[(m,m'),q] / c, dZ2 = 2 B1 - Z o B2 - dZ3.
//
// Why pairs: a weight computed once per unordered pair feeds both inducing rows through matrix products (the sum over lanes is
// free on the matrix pipe), so the exponent, the exponential and the two back products are evaluated on M (M + 1) / 2 instead of
// M^2 entries per data row: 6 N P Q = 3 N M (M + 1) Q multiply-add flop against 4 N M^2 Q of the sweep kernels.
//
// Ownership.  A workgroup of eight waves owns 512 pairs (4 tiles of 16 pairs per wave) and a range of data rows, walked in steps
// of 16 rows.  Per step a wave runs, for each of its pair tiles, v_mfma_f64_16x16x4 products
//   Gram  [16 rows x 16 pairs], K = Q     A = d2 row tile (LDS), B = z[a,q] z[b,q] formed on the fly from the Z image in LDS
//   C     [16 pairs x 16 q] x QT, K = 16 rows   A = the weights as they stand in the accumulator layout, B = d2 row tile
//   T     [16 rows x 16 q] x QT, K = 16 pairs   A = the weights transposed through a wave-private LDS tile, B = z z on the fly
// C stays in accumulator registers for the whole row range (flushed once per workgroup), the pair sums of psi2 in one register
// per tile; T and V2 are sums over pairs: the eight waves add theirs into an LDS tile [16 rows][V2 | T] (fp64 LDS atomics) that
// leaves the chip once per step as this pair group's partial.  The partials of the G pair groups are summed by the row kernel.
// Pair order: circular diagonals p = d M + a, b = (a + d) mod M, d = 0 .. M/2: every unordered pair once, no padding inside the
// list, and the 16 pairs of a tile touch 16 different a and 16 different b (conflict-free h reads and V2 adds).
#include "rgp_psi.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define PAIR_NT 4                 // pair tiles per wave (their C accumulators: 4 x QT x 4 doubles per lane)
#define PAIR_WAVES 8
#define PAIR_GROUP (PAIR_WAVES * PAIR_NT * 16)       // pairs per workgroup

static __device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exp_tab with a NaN-preserving clamp
static __device__ __forceinline__ double exp_tab_n(double x, const double* __restrict__ T) {
  x = (x < -800.0) ? -800.0 : x;
  const double t = fma(x, 0x1.71547652b82fep+8, 0x1.8p52);
  const int n = __double2loint(t);
  const double nf = t - 0x1.8p52;
  double r = fma(nf, -0x1.62e42fee00000p-9, x);
  r = fma(nf, -0x1.a39ef35793c76p-41, r);
  const double tj = T[n & (PSI2_EXP_TAB - 1)];
  double q = fma_rrr(r, 0x1.5555555555555p-5, 0x1.5555555555555p-3);
  q = fma_rrr(q, r, 0.5);
  q = fma_rrr(q, r, 1.0);
  q = q * r;
  return ldexp(fma(tj, q, tj), n >> 8);
}

// pair p -> (a, b) on the circular diagonals
static __device__ __forceinline__ void pair_ab(int p, int M, int P, int& a, int& b) {
  if (p >= P) { a = 0; b = 0; return; }
  const int d = p / M;
  a = p - d * M;
  b = a + d;
  if (b >= M) b -= M;
}

struct PairArgs {
  int64_t N, ntiles;
  int M, P, G, RR, HS, OS, PW, M2, tri, Ppad;
  const double *rec, *Zc, *lam, *dLs;
  double *PV, *Cpart, *p2_part;
};

// LDS image (doubles): Z [M][ZS] | exp table [256] | 2 x input record [d2 16 x TS | h' 16 x HS] | 2 x output tile [16][OS] |
// 8 x transposition tile [16][17]
template <int QP>
__global__ void __launch_bounds__(512, 2) psi2_pair_kernel(const PairArgs a) {
  constexpr int KQ = QP / 4, QT = (QP + 15) / 16, ZS = 16 * QT + 1, TS = 16 * QT + 1, NT = PAIR_NT;
  extern __shared__ double lds[];
  const int M = a.M, HS = a.HS, OS = a.OS;
  const int RS = 16 * (TS + HS);
  double* const Zl = lds;
  double* const etab = Zl + ((M * ZS + 1) & ~1);
  double* const inb = etab + PSI2_EXP_TAB;
  double* const outb = inb + 2 * RS;
  double* const tball = outb + 2 * 16 * OS;

  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int gi = blockIdx.x % a.G, rr = blockIdx.x / a.G;
  const int64_t t0 = a.ntiles * rr / a.RR, t1 = a.ntiles * (rr + 1) / a.RR;
  double* const tb = tball + wave * (16 * 17);

  // ---- one-time set-up: Z image, exp table, zeroed output tiles, first record, per-lane pair constants
  for (int i = t; i < M * ZS; i += 512) {
    const int m = i / ZS, q = i - m * ZS;
    Zl[i] = (q < QP) ? a.Zc[(size_t)m * QP + q] : 0.0;
  }
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  for (int i = t; i < 2 * 16 * OS; i += 512) outb[i] = 0.0;
  if (t0 < t1) {
    const double2* src = reinterpret_cast<const double2*>(a.rec + (size_t)t0 * RS);
    double2* dst = reinterpret_cast<double2*>(inb);
    for (int i = t; i < RS / 2; i += 512) dst[i] = src[i];
  }
  int pkn[NT], pkt[NT][2];
  double dl2[NT], kp[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tau = (gi * PAIR_WAVES + wave) * NT + k;
    int pa, pb;
    const int p = 16 * tau + c;
    pair_ab(p, M, a.P, pa, pb);
    pkn[k] = pa | (pb << 8) | ((p < a.P ? 1 : 0) << 16);
    dl2[k] = (p < a.P) ? a.dLs[(size_t)pa * M + pb] * (pa == pb ? 1.0 : 2.0) : 0.0;
    double s = 0.0;
    for (int q = 0; q < QP; ++q) s += a.lam[q] * a.Zc[(size_t)pa * QP + q] * a.Zc[(size_t)pb * QP + q];
    kp[k] = 0.5 * s;
    int w4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      pair_ab(16 * tau + 4 * kk + g, M, a.P, pa, pb);
      w4[kk] = pa | (pb << 8);
    }
    pkt[k][0] = w4[0] | (w4[1] << 16);
    pkt[k][1] = w4[2] | (w4[3] << 16);
  }
  d4 Cacc[NT][QT];
  double psum[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    psum[k] = 0.0;
#pragma unroll
    for (int j = 0; j < QT; ++j) Cacc[k][j] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();

  constexpr int NPRE = (16 * (TS + 112) / 2 + 511) / 512;       // double2 per thread of the largest record (HS <= 112)
  for (int64_t s = t0; s < t1; ++s) {
    const int buf = (int)((s - t0) & 1);
    const double* const d2t = inb + buf * RS;                    // [16][TS]
    const double* const hq = d2t + 16 * TS;                      // [16][HS]
    double* const out = outb + buf * (16 * OS);                  // [16][OS]: V2 in columns [0, HS), T in [HS, HS + 16 QT)
    const bool more = s + 1 < t1;
    // next record: global loads now, LDS stores at the end of the step
    double2 pre[NPRE];
    if (more) {
      const double2* src = reinterpret_cast<const double2*>(a.rec + (size_t)(s + 1) * RS);
#pragma unroll
      for (int j = 0; j < NPRE; ++j) {
        const int i = t + 512 * j;
        pre[j] = src[i < RS / 2 ? i : RS / 2 - 1];
      }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      asm volatile("" : "+v"(pkn[k]), "+v"(pkt[k][0]), "+v"(pkt[k][1]));      // keep the derived addresses out of loop-invariant registers
    }

    // ---- Gram: acc[k][16 rows x 16 pairs] = sum_q d2[n,q] z[a,q] z[b,q]
    d4 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      const double A = d2t[c * TS + 4 * kq + g];
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int pa = pkn[k] & 0xff, pb = (pkn[k] >> 8) & 0xff;
        const double zz = Zl[pa * ZS + 4 * kq + g] * Zl[pb * ZS + 4 * kq + g];
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, zz, acc[k], 0, 0, 0);
      }
    }
    // ---- exponent, exp, weights, V2
    double w[NT][4];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int pa = pkn[k] & 0xff, pb = (pkn[k] >> 8) & 0xff;
      double ps = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * r + g;
        const double E = fma(-0.5, acc[k][r], (hq[row * HS + pa] + hq[row * HS + pb]) + kp[k]);
        const double e = exp_tab_n(E, etab);
        ps += e;
        w[k][r] = e * dl2[k];
        atomicAdd(&out[row * OS + pa], w[k][r]);
        atomicAdd(&out[row * OS + pb], w[k][r]);
      }
      psum[k] += ps;
    }
    // ---- C[k][16 pairs x 16 q] += w^T d2   (K = the step's 16 rows)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        const double B = d2t[(4 * r + g) * TS + 16 * j + c];
#pragma unroll
        for (int k = 0; k < NT; ++k) Cacc[k][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[k][r], B, Cacc[k][j], 0, 0, 0);
      }
    }
    // ---- T[16 rows x 16 q] += w (z z)^T   (K = the tile's 16 pairs; the weights transposed through the wave's LDS tile)
    d4 Tacc[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) Tacc[j] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NT; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) tb[(4 * r + g) * 17 + c] = w[k][r];
      wave_lds_fence();
      double at[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) at[kk] = tb[c * 17 + 4 * kk + g];
      wave_lds_fence();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int pk = (pkt[k][kk >> 1] >> (16 * (kk & 1))) & 0xffff;
        const int pa = pk & 0xff, pb = pk >> 8;
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const double zz = Zl[pa * ZS + 16 * j + c] * Zl[pb * ZS + 16 * j + c];
          Tacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[kk], zz, Tacc[j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&out[(4 * r + g) * OS + HS + 16 * j + c], Tacc[j][r]);
    // ---- stage the next record, hand the finished output tile over
    if (more) {
      double2* dst = reinterpret_cast<double2*>(inb + (buf ^ 1) * RS);
#pragma unroll
      for (int j = 0; j < NPRE; ++j) {
        const int i = t + 512 * j;
        if (i < RS / 2) dst[i] = pre[j];
      }
    }
    __syncthreads();
    {
      const int row = t >> 5;
      const int64_t n = s * 16 + row;
      double* const orow = out + row * OS;
      double* const prow = a.PV + ((size_t)gi * a.N + (n < a.N ? n : 0)) * a.PW;
      for (int col = t & 31; col < a.PW; col += 32) {
        const int src = col < a.M2 ? col : HS + (col - a.M2);
        const double v = orow[src];
        orow[src] = 0.0;
        if (n < a.N) prow[col] = v;
      }
    }
  }

  // ---- flush: C partial of this (row range, pair group), pair sums of psi2
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tau = (gi * PAIR_WAVES + wave) * NT + k;
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      const int q = 16 * j + c;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (q < QP) a.Cpart[((size_t)rr * a.Ppad + 16 * tau + 4 * r + g) * QP + q] = Cacc[k][j][r];
    }
    double ps = psum[k];
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    if (g == 0 && ((pkn[k] >> 16) & 1)) {
      const int pa = pkn[k] & 0xff, pb = (pkn[k] >> 8) & 0xff;
      const int lo = pa < pb ? pa : pb, hi = pa < pb ? pb : pa;
      a.p2_part[(size_t)rr * a.tri + hi * (hi + 1) / 2 + lo] = ps;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Input records of the main kernel, one per tile of 16 data rows: [d2 16 x TS (zero padded) | h' 16 x HS].
// Thread = (inducing row m, half of the tile's rows); the tile's d2 and d2 mu rows are staged in LDS and broadcast.
// Rows beyond N get d2 = 0 and h' = -inf (their exponentials are 0).
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ void __launch_bounds__(256) psi2_pair_prep_kernel(const int64_t N, const int64_t ntiles, const int M, const int HS,
                                                              const double* __restrict__ Zc, const double* __restrict__ lam,
                                                              const double* __restrict__ Xmu, const double* __restrict__ d2,
                                                              const double* __restrict__ g0, double* __restrict__ rec) {
  constexpr int QT = (QP + 15) / 16, TS = 16 * QT + 1;
  __shared__ __attribute__((aligned(16))) double2 dd[16 * QP];          // (d2, d2 mu)
  __shared__ double gs[16];
  const int t = threadIdx.x, m = t & 127, rh = t >> 7;
  const int RS = 16 * (TS + HS);
  const bool act = m < M;
  double z[QP];
  double ellq = 0.0;
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    z[q] = act ? Zc[(size_t)m * QP + q] : 0.0;
    ellq += 0.25 * lam[q] * z[q] * z[q];
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * 16;
    double* const r = rec + (size_t)tile * RS;
    __syncthreads();
    for (int i = t; i < 16 * QP; i += 256) {
      const int row = i / QP;
      const bool ok = n0 + row < N;
      const double dv = ok ? d2[n0 * QP + i] : 0.0, mv = ok ? Xmu[n0 * QP + i] : 0.0;
      dd[i] = make_double2(dv, dv * mv);
    }
    if (t < 16) gs[t] = (n0 + t < N) ? 0.5 * g0[n0 + t] : -INFINITY;
    __syncthreads();
    for (int i = t; i < 16 * TS; i += 256) {
      const int row = i / TS, q = i - row * TS;
      r[i] = (q < QP) ? dd[row * QP + q].x : 0.0;
    }
#pragma unroll 1
    for (int rw = rh * 8; rw < rh * 8 + 8; ++rw) {
      const double2* dr = dd + rw * QP;
      double h0 = 0.0, h1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; q += 2) {
        const double2 x0 = dr[q], x1 = dr[q + 1];
        h0 += z[q] * (x0.y - 0.25 * x0.x * z[q]);
        h1 += z[q + 1] * (x1.y - 0.25 * x1.x * z[q + 1]);
      }
      if (m < HS) r[16 * TS + rw * HS + m] = act ? ((h0 + h1) - ellq) + gs[rw] : 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Row kernel: sums the pair groups' partials of V2 and T per tile of 16 rows, then
//   job A  (thread = (row, 16-column stripe)): [a1 | vz2 | ws] = v [Z | Z^2 | 1] and the row gradients, dinv_l and sum-of-weights
//           partials;   job B (thread = (inducing row, half of the columns)): B1 += v^T (d2 mu), B2 += v^T d2 in registers over
//           the workgroup's tiles; at the end the slab dz[q][m] = 2 B1 - z B2.
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ void __launch_bounds__(256) psi2_pair_rows_kernel(const int64_t N, const int64_t ntiles, const int M, const int Q,
                                                              const int G, const int PW, const int M2,
                                                              const double* __restrict__ PV, const double* __restrict__ Zc,
                                                              const double* __restrict__ lam, const double* __restrict__ Xmu,
                                                              const double* __restrict__ d2, double* __restrict__ gmu,
                                                              double* __restrict__ gS, const int accumulate,
                                                              double* __restrict__ dz_part, double* __restrict__ dlam_part,
                                                              double* __restrict__ wsum_part) {
  constexpr int QT = (QP + 15) / 16, QH = QP / 2, ZQ = QP + 1;
  extern __shared__ double sm[];
  const int VS = M2 + 1;
  double2* const dd = reinterpret_cast<double2*>(sm);     // [16][QP] (d2 mu, d2)
  double* const mus = sm + 2 * 16 * QP;                   // [16][QP]
  double* const Tt = mus + 16 * QP;                       // [16][QP]
  double* const red = Tt + 16 * QP;                       // [16][QP] + [16]
  double* const Zs = red + 16 * QP + 16;                  // [M][ZQ]
  double* const vt = Zs + M * ZQ;                         // [16][VS]
  const int t = threadIdx.x;
  for (int i = t; i < M * QP; i += 256) { const int m = i / QP, q = i - m * QP; Zs[m * ZQ + q] = Zc[i]; }
  // job A identity
  const int ra = t >> 4, cq = t & 15;
  // job B identity
  const int mb = t & 127, qh = t >> 7;
  const bool actb = mb < M;
  double b1[QH], b2[QH];
#pragma unroll
  for (int j = 0; j < QH; ++j) { b1[j] = 0.0; b2[j] = 0.0; }
  double dl_acc[QT], ws_acc = 0.0;
#pragma unroll
  for (int j = 0; j < QT; ++j) dl_acc[j] = 0.0;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * 16;
    __syncthreads();
    // sum of the partials: V2 / 2 and T
    for (int i = t; i < 16 * PW; i += 256) {
      const int row = i / PW, col = i - row * PW;
      const int64_t n = n0 + row;
      double s = 0.0;
      if (n < N)
        for (int gg = 0; gg < G; ++gg) s += PV[((size_t)gg * N + n) * PW + col];
      if (col < M2) vt[row * VS + col] = 0.5 * s;
      else Tt[row * QP + (col - M2)] = s;
    }
    for (int i = t; i < 16 * QP; i += 256) {
      const int row = i / QP;
      const bool ok = n0 + row < N;
      const double dv = ok ? d2[n0 * QP + i] : 0.0, mv = ok ? Xmu[n0 * QP + i] : 0.0;
      dd[i] = make_double2(dv * mv, dv);
      mus[i] = mv;
    }
    __syncthreads();
    // job A
    {
      double a1[QT], vz2[QT], ws = 0.0;
#pragma unroll
      for (int j = 0; j < QT; ++j) { a1[j] = 0.0; vz2[j] = 0.0; }
      const double* vr = vt + ra * VS;
      for (int m = 0; m < M; ++m) {
        const double vv = vr[m];
        ws += vv;
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const int q = cq + 16 * j;
          const double z = (q < QP) ? Zs[m * ZQ + q] : 0.0;
          const double lz = vv * z;
          a1[j] += lz;
          vz2[j] += lz * z;
        }
      }
      const int64_t n = n0 + ra;
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        const int q = cq + 16 * j;
        if (q < QP && n < N) {
          if (q < Q) {
            const double mu = mus[ra * QP + q], dq = dd[ra * QP + q].y;
            const double a23 = 0.5 * (vz2[j] + Tt[ra * QP + q]);
            const double dmu = 2.0 * dq * (a1[j] - mu * ws);
            const double ddq = -mu * mu * ws + 2.0 * mu * a1[j] - a23 + 0.5 * ws / dq;
            const double dS = -2.0 * dq * dq * ddq;
            if (accumulate) { gmu[n * QP + q] += dmu; gS[n * QP + q] += dS; }
            else { gmu[n * QP + q] = dmu; gS[n * QP + q] = dS; }
            const double il = 1.0 / lam[q];
            dl_acc[j] += ddq * (dq * il) * (dq * il) - 0.5 * ws * il;
          } else if (!accumulate) {
            gmu[n * QP + q] = 0.0; gS[n * QP + q] = 0.0;
          }
        }
      }
      if (cq == 0 && n < N) ws_acc += ws;
    }
    // job B
    if (actb) {
#pragma unroll 2
      for (int r = 0; r < 16; ++r) {
        const double vv = vt[r * VS + mb];
        const double2* dr = dd + r * QP + qh * QH;
#pragma unroll
        for (int j = 0; j < QH; ++j) {
          const double2 x = dr[j];
          b1[j] += vv * x.x;
          b2[j] += vv * x.y;
        }
      }
    }
  }
  // slab of this workgroup: dz[q][m] = 2 B1[m,q] - z[m,q] B2[m,q]   (layout [QP][128])
  {
    double* slab = dz_part + (size_t)blockIdx.x * QP * RGP_PSI2_TEAM;
#pragma unroll
    for (int j = 0; j < QH; ++j) {
      const int q = qh * QH + j;
      slab[q * RGP_PSI2_TEAM + mb] = actb ? 2.0 * b1[j] - Zs[mb * ZQ + q] * b2[j] : 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    const int q = cq + 16 * j;
    if (q < QP) red[ra * QP + q] = dl_acc[j];
  }
  if (cq == 0) red[16 * QP + ra] = ws_acc;
  __syncthreads();
  if (t < QP) {
    double s = 0.0;
    for (int r = 0; r < 16; ++r) s += red[r * QP + t];
    dlam_part[(size_t)blockIdx.x * QP + t] = (t < Q) ? s : 0.0;
  }
  if (t == 0) {
    double s = 0.0;
    for (int r = 0; r < 16; ++r) s += red[16 * QP + r];
    wsum_part[blockIdx.x] = s;
  }
}

// slab [q][m] = -dZ3[m,q] = -sum_m' z[m',q] Csum[(m,m'),q] / c   (Csum carries c_p = 2 off the diagonal)
__global__ void __launch_bounds__(256) psi2_pair_dz3_kernel(const int M, const int QP, const int P, const double* __restrict__ Csum,
                                                             const double* __restrict__ Zc, double* __restrict__ slab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= QP * RGP_PSI2_TEAM) return;
  const int q = i % QP, m = i / QP;           // consecutive threads walk q: coalesced reads of C
  if (m >= M) { slab[q * RGP_PSI2_TEAM + m] = 0.0; return; }
  double s = 0.0;
  for (int mp = 0; mp < M; ++mp) {
    int d = mp - m, a = m;
    if (d < 0) d += M;
    if (2 * d > M || (2 * d == M && mp < m)) { d = M - d; a = mp; }
    const int p = d * M + a;
    const double cv = Csum[(size_t)p * QP + q];
    s += Zc[(size_t)mp * QP + q] * (mp == m ? cv : 0.5 * cv);
  }
  slab[q * RGP_PSI2_TEAM + m] = -s;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
static int pair_hs(int M) { return M <= 16 ? 16 : ((M - 16 + 31) / 32) * 32 + 16; }      // >= M, = 16 mod 32 (bank spread of the four row groups)

bool psi2_pair_geom(rgp_handle h, int64_t N, int M, int QP, PairGeom& g) {
  if (M < 2 || M > 112 || QP < 8 || QP > RGP_MAX_QP || (QP & 3) || N < 1) return false;
  g.M = M; g.QP = QP;
  g.QT = (QP + 15) / 16;
  g.TS = 16 * g.QT + 1;
  g.HS = pair_hs(M);
  g.OS = (g.HS + 16 * g.QT + 31) / 32 * 32 + 16;
  g.RS = 16 * (g.TS + g.HS);
  g.M2 = (M + 1) & ~1;
  g.PW = g.M2 + QP;
  g.P = M * (M + 1) / 2;
  g.G = (g.P + PAIR_GROUP - 1) / PAIR_GROUP;
  g.Ppad = g.G * PAIR_GROUP;
  g.ntiles = (N + 15) / 16;
  int64_t rr = (h->num_cu - (h->overlap && h->num_cu > 64 ? 1 : 0)) / g.G;
  if (rr < 1) rr = 1;
  if (rr > g.ntiles) rr = g.ntiles;
  g.RR = (int)rr;
  int64_t nb = 2 * (int64_t)h->num_cu;
  if (nb > g.ntiles) nb = g.ntiles;
  g.NB = (int)nb;
  const size_t zs = (size_t)(M * g.TS + 1) & ~(size_t)1;
  g.lds = (zs + PSI2_EXP_TAB + 2 * (size_t)g.RS + 2 * 16 * (size_t)g.OS + PAIR_WAVES * 16 * 17) * sizeof(double);
  return g.lds <= 160 * 1024;
}

// The pair form pays once a workgroup's 512 pairs are mostly real and every workgroup has a few steps of 16 rows to walk
bool psi2_pair_selected(rgp_handle h, int64_t N, int M, int QP, PairGeom& g) {
  int mode = h->psi2_pair;
  if (mode < 0) {
    static int env_mode = -2;
    if (env_mode == -2) { const char* e = getenv("RGP_PSI2_PAIR"); env_mode = e ? atoi(e) : 1; }
    mode = env_mode;
  }
  if (mode <= 0 || h->deterministic) return false;
  if (!psi2_pair_geom(h, N, M, QP, g)) return false;
  if (mode >= 2) return true;
  return M >= 48 && N >= 2048;
}

size_t psi2_pair_ws_doubles(const PairGeom& g, int64_t N) {
  auto r = [](size_t n) { return (n + 31) & ~(size_t)31; };
  return r((size_t)g.ntiles * g.RS) + r((size_t)g.G * N * g.PW) + r((size_t)g.RR * g.Ppad * g.QP) + r((size_t)g.Ppad * g.QP) + 64;
}

template <int QP>
static hipError_t pair_launch(hipStream_t s, const PairGeom& g, int64_t N, int Q, const ZTerms& zt, const RowTerms& rt,
                              const double* dLs, const Psi2Out& o, double* ws) {
  auto r = [](size_t n) { return (n + 31) & ~(size_t)31; };
  double* rec = ws;
  double* PV = rec + r((size_t)g.ntiles * g.RS);
  double* Cpart = PV + r((size_t)g.G * N * g.PW);
  double* Csum = Cpart + r((size_t)g.RR * g.Ppad * g.QP);
  const int M = g.M;
  int prep_grid = (int)(g.ntiles < 1024 ? g.ntiles : 1024);
  hipLaunchKernelGGL((psi2_pair_prep_kernel<QP>), dim3(prep_grid), dim3(256), 0, s, N, g.ntiles, M, g.HS, zt.Zc, zt.lam, rt.Xmu,
                     rt.d2, rt.g0, rec);
  PairArgs a{};
  a.N = N; a.ntiles = g.ntiles; a.M = M; a.P = g.P; a.G = g.G; a.RR = g.RR; a.HS = g.HS; a.OS = g.OS; a.PW = g.PW; a.M2 = g.M2;
  a.tri = M * (M + 1) / 2; a.Ppad = g.Ppad;
  a.rec = rec; a.Zc = zt.Zc; a.lam = zt.lam; a.dLs = dLs; a.PV = PV; a.Cpart = Cpart; a.p2_part = o.p2_part;
  hipError_t e = hipFuncSetAttribute((const void*)psi2_pair_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((psi2_pair_kernel<QP>), dim3(g.G * g.RR), dim3(512), g.lds, s, a);
  // row gradients, dinv_l / weight-sum partials, B slabs
  const size_t lr = (5 * 16 * (size_t)QP + 16 + (size_t)M * (QP + 1) + 16 * (size_t)(g.M2 + 1)) * sizeof(double);
  e = hipFuncSetAttribute((const void*)psi2_pair_rows_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((psi2_pair_rows_kernel<QP>), dim3(g.NB), dim3(256), lr, s, N, g.ntiles, M, Q, g.G, g.PW, g.M2, PV, zt.Zc, zt.lam,
                     rt.Xmu, rt.d2, o.gmu, o.gS, o.accumulate, o.dz_part, o.dlam_part, o.wsum_part);
  // dZ3 from the C partials: one more slab behind the row kernel's
  launch_reduce(s, g.RR, (int64_t)g.Ppad * QP, (int64_t)g.Ppad * QP, Cpart, Csum, 1.0, 0);
  hipLaunchKernelGGL(psi2_pair_dz3_kernel, dim3((QP * RGP_PSI2_TEAM + 255) / 256), dim3(256), 0, s, M, QP, g.P, Csum, zt.Zc,
                     o.dz_part + (size_t)g.NB * QP * RGP_PSI2_TEAM);
  return hipGetLastError();
}

rgp_status launch_psi2_pair(rgp_handle h, const PairGeom& g, int64_t N, int Q, const ZTerms& zt, const RowTerms& rt,
                            const double* dLs, const Psi2Out& out, double* ws) {
  hipStream_t s = h->stream;
  hipError_t e = hipSuccess;
  switch (g.QP) {
    case 8:  e = pair_launch<8>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 16: e = pair_launch<16>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 20: e = pair_launch<20>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 24: e = pair_launch<24>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 32: e = pair_launch<32>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 40: e = pair_launch<40>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 48: e = pair_launch<48>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 56: e = pair_launch<56>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    case 64: e = pair_launch<64>(s, g, N, Q, zt, rt, dLs, out, ws); break;
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2 (pair form): unsupported padded Q");
  }
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 (pair form) launch: ") + hipGetErrorString(e));
  return RGP_OK;
}
