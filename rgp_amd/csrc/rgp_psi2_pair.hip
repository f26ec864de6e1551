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
// Ownership.  A workgroup of eight waves owns 512 pairs (4 tiles of 16 pairs per wave; 640 = 5 tiles per wave for windows of up
// to 20 columns, whose C accumulators are smaller) and a range of data rows, walked in steps
// of 16 rows.  Per step a wave runs, for each of its pair tiles, v_mfma_f64_16x16x4 products
//   Gram  [16 rows x 16 pairs], K = Q     A = d2 row tile (LDS), B = z[a,q] z[b,q] formed on the fly from the Z image in LDS
//   C     [16 pairs x 16 q] x QT, K = 16 rows   A = the weights as they stand in the accumulator layout, B = d2 row tile
//   T     [16 rows x 16 q] x QT, K = 16 pairs   A = the weights transposed through a wave-private LDS tile, B = z z on the fly
// C stays in accumulator registers for the whole row range (flushed once per workgroup), the pair sums of psi2 in one register
// per tile; T and V2 are sums over pairs: the eight waves add theirs into an LDS tile [16 rows][V2 | T] (fp64 LDS atomics) that
// leaves the chip once per step as this pair group's partial.  The partials of the G pair groups are summed by the row kernel.
// Pair order: circular diagonals p = d M + a, b = (a + d) mod M, d = 0 .. M/2: every unordered pair once, no padding inside the
// list, and the 16 pairs of a tile touch 16 different a and 16 different b (conflict-free h reads and V2 adds).
// For M > 112 (the blocked layer step) the same body runs on pair groups over 16 x 16 blocks of inducing rows and on local row
// indices (template flag BLK; see "Blocked form" at the end of the file).
#include "rgp_psi.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define PAIR_DEFAULT_MODE 1         // the pair form where it pays (psi2_pair_selected)
// Pair tiles per wave (their C accumulators: NT x (8 NF + 2 NR) registers per lane): four, five where the window is narrow
// enough for the registers (the step's fixed work -- barrier, record, write-out -- is then shared by more pairs: 67.7 against
// 65.9 steps/s at the benchmark size from the Q = 20 pass alone; six spill).
// (The hardware favours the older wave of a SIMD, which with equal work reaches the step's barrier early and leaves the younger
// one to finish alone; giving the younger waves fewer tiles (4 + 2) was measured and does not pay: 10.7 M against 9.6 M cycles per
// launch at the benchmark size.)
#ifndef PAIR_NT_NARROW
#define PAIR_NT_NARROW 5
#endif
static constexpr int pair_nt(int QP) { return QP <= 20 ? PAIR_NT_NARROW : 4; }
#define PAIR_WAVES 8
#ifndef PAIR_WO0
#define PAIR_WO0 1                // 1: waves 0-3 alone write the finished V2 / T tiles out and request the records
#endif
#define PAIR_GROUP_MAX (128 * 6)                     // pairs per workgroup at most (8 waves x NT tiles x 16)
#define PAIR_LB 6                 // blocked form: column blocks of 16 inducing rows a pair group may touch

static __device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exp_tab with a NaN-preserving clamp
static __device__ __forceinline__ double exp_tab_n(double x, const double* __restrict__ T) {
  x = clamp_exp_arg(x);
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

// pair p -> (a, b) on the circular diagonals.  The slots behind the last pair (their weights are zero) stand for diagonal pairs
// (a, a) with 16 different a per tile: all on one inducing row they would send every lane's h' read and V2 add of the tile to
// one LDS address, and the workgroups of the last pair group, 16-way serialised there, took 20 % longer than all others.
static __device__ __forceinline__ void pair_ab(int p, int M, int P, int& a, int& b) {
  if (p >= P) { a = p % M; b = a; return; }
  const int d = p / M;
  a = p - d * M;
  b = a + d;
  if (b >= M) b -= M;
}

struct PairArgs {
  int64_t N, ntiles;
  int M, P, G, RR, PW, M2, tri, Ppad;
  const double *rec, *Zc, *lam, *dLs;
  double *PV, *Cpart, *p2_part;
  double* dbg;          // per-phase cycle stamps of a -DPAIR_STAMPS build (nullptr otherwise)
  // blocked form (M > 112): h' in 16 x 16 blocks [tile][column block], outputs V2 [N][Mld], T [N][QP], C and pair sums in tile order
  const double* Hq;
  double *V2, *Tg, *Csum, *p2sum;
  int nb, Mld, ntl, ntasks, rem;          // rem: inducing rows in the last block of 16 (1 .. 16)
};
#ifdef PAIR_STAMPS
// the counter reads are issued where the phases end and collected once per step: a read that was waited for on the spot would
// drain the wave's LDS queue at every stamp
#define PSTAMP(i) asm volatile("s_memtime %0" : "=s"(sq_[i]))
#else
#define PSTAMP(i) do { } while (0)
#endif

// LDS image (doubles; every offset but the length of the Z image is a compile-time constant, so that most addresses are a
// per-lane register plus an immediate): exp table [256] | 2 x input record [d2 16 x TS | h' 16 x HS | pad to 1 KiB] |
// 2 x V2 tile [16][HS] | 3 x T tile [16][TT] | 8 x transposition tile [16][17] | -2 k_p [512] | pair sums [512] | c_p dLs [512] |
// Z [M][ZS]
//
// Two roles.  The eight waves of a workgroup sit two to a SIMD (waves w and w + 4), and the two waves of a SIMD share its fp64
// pipe: in step with each other they would wait for their LDS reads at the same moments and want the pipe at the same moments.
// Every wave runs [GE(k): Gram, exponentials, V2 adds] [C(k)] [T(k)] per step k, with ONE workgroup barrier per step: waves
// 0-3 (role 0) pass it after T(k), waves 4-7 (role 1) between C(k) and T(k).  Between two barriers role 0 therefore runs
// GE(k+1), C(k+1), T(k+1) while role 1 runs T(k), GE(k+1), C(k+1).  What the barrier B_k guarantees: V2 tile k complete (all
// waves add it in GE(k)), T tile k - 1 complete (role 1 adds T(k-1) after B_(k-1)).  After B_k every thread adds its slice of V2
// tile k and T tile k - 1 to the global sums and clears it; V2 tiles rotate over two buffers, T tiles over three (k - 1 being
// written out, k receiving role 1's adds, k + 1 role 0's).  The record of step k + 2 is requested right after B_k (LDS-DMA into
// the buffer of record k, which nobody reads any more) and has landed before B_(k+1).
//
// Software pipeline.  The register allocator works at the 256-register limit of two waves per SIMD and, left alone, waits for
// every LDS read right after issuing it.  The step is therefore written in the order it should issue -- operands of the next
// group of matrix instructions are requested before the current group is issued -- with scheduling barriers between the groups.
#define PAIR_HS 112               // row stride of the h' and V2 tiles (= 16 mod 32: the four row groups of a wave on disjoint banks); M <= 112
#define PAIR_SB() __builtin_amdgcn_sched_barrier(0)
template <int QP, int NT, bool role1, bool BLK>
static __device__ __forceinline__ void psi2_pair_body(const PairArgs& a, const int gi, const int rr) {
  constexpr int KQ = QP / 4, QT = (QP + 15) / 16, ZS = 16 * QT + 1, TS = 16 * QT + 1, HS = PAIR_HS;
  // Blocked form (BLK, M > 112).  The pair group's pairs live in at most PAIR_LB blocks of 16 inducing rows; the workgroup sees
  // only those: "local" inducing row l = 16 * (local block) + (row in block), the Z image holds the local rows, and the h' and V2
  // tiles are stored block by block, [local block][16 data rows][16], which is the order in which the h' blocks lie in memory
  // (one LDS-DMA piece = 8 rows of a block).  Everything between the set-up and the write-out is the same code on local rows.
  constexpr int ND2 = (16 * TS + 127) / 128;                                 // 1-KiB pieces of the d2 tile
  constexpr int HR = BLK ? 16 : HS;                                          // row stride of the h' and V2 tiles
  constexpr int VT = BLK ? PAIR_LB * 256 : 16 * HS;                          // length of a V2 tile
  constexpr int HOFF = BLK ? ND2 * 128 : 16 * TS;                            // h' tile within the record
  constexpr int TT = ((16 * QT) % 32 == 16) ? 16 * QT : 16 * QT + 16;       // T tile row stride: = 16 mod 32
  // Window columns of the C and T products: NF full tiles of 16 on v_mfma_f64_16x16x4 and NR groups of 4 on
  // v_mfma_f64_4x4x4_4b (four 4x4x4 blocks, a quarter of the cycles): no matrix instruction is spent on the padding of Q to 16.
  // In the small instruction lane (g, c) supplies the same left operands as in the large one (C: w[row 4 kk + g][pair c], T:
  // w[row c][pair 4 kk + g]) and, as right operand, column c & 3 of the group at the row / pair g; its result lane holds
  // C[pair 4 (c >> 2) + g][column c & 3] resp. T[row 4 (c >> 2) + g][column c & 3].
  constexpr int NF = QP / 16, NR = (QP % 16) / 4, NF1 = NF > 0 ? NF : 1, NR1 = NR > 0 ? NR : 1, NE = NF + NR;
  constexpr int RSP = BLK ? ND2 * 128 + PAIR_LB * 256 : (16 * (TS + HS) + 127) / 128 * 128;    // record length: whole 1-KiB pieces
  constexpr int O_ETAB = 0, O_INB = PSI2_EXP_TAB, O_V2 = O_INB + 2 * RSP, O_TT = O_V2 + 2 * VT, O_TB = O_TT + 3 * 16 * TT,
                O_KP = O_TB + PAIR_WAVES * 16 * 17, O_PS = O_KP + 128 * NT, O_DL = O_PS + 128 * NT, O_Z = O_DL + 128 * NT,
                O_TI = O_Z + PAIR_LB * 16 * ZS;                               // blocked form: tile table [8 NT] and block list [PAIR_LB] (ints)
  extern __shared__ double lds[];
  // LDS operands are addressed by their byte offset in the workgroup's LDS image: the kernel has no static LDS, so the dynamic
  // image starts at offset 0 and an offset is the whole address (no base register, no add in front of every access)
  typedef __attribute__((address_space(3))) double lds_f64;
#define LDSD(off) (*(lds_f64*)(unsigned)(off))
#define LDSP(off) ((double*)(lds_f64*)(unsigned)(off))
  const int M = a.M;
  double* const Zl = lds + O_Z;
  double* const etab = lds + O_ETAB;
  double* const inb = lds + O_INB;
  double* const v2b = lds + O_V2;
  double* const ttb = lds + O_TT;

  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tb = wave * NT;                                      // the wave's first tile within the pair group
  const int64_t t0 = a.ntiles * rr / a.RR, t1 = a.ntiles * (rr + 1) / a.RR;
  const int nsteps = (int)(t1 - t0);

  auto hcol = [](int p8) { return BLK ? (((p8 >> 7) << 11) | (p8 & 127)) : p8; };      // 8 l -> byte offset of column l in a tile row
  int* const tinfo = (int*)(lds + O_TI);
  int* const gbl = tinfo + 8 * NT;
  // record r of this workgroup -> LDS buffer r & 1, one 1-KiB piece per wave instruction
  // (inline asm: the compiler, unable to tell the DMA's LDS destination from the tiles the waves read meanwhile, would wait for
  // the transfer before the next LDS read; the waves wait for it themselves, with s_waitcnt vmcnt(0) in front of the barrier
  // that precedes the record's first use)
  auto request_record = [&](int r, int nw) {           // by waves 0 .. nw - 1
    const double* src = a.rec + (size_t)(t0 + r) * (BLK ? ND2 * 128 : RSP) + lane * 2;
    const unsigned dst = (unsigned)(size_t)(inb + (r & 1) * RSP);
    for (int ch = wave; ch * 128 < RSP; ch += nw) {
      const double* gsrc = src + ch * 128;
      if (BLK && ch >= ND2) {                            // h' block (ch - ND2) / 2 of the group's list, rows 0-7 or 8-15
        int gbv = __builtin_amdgcn_readfirstlane(gbl[(ch - ND2) >> 1]);
        if (gbv < 0) gbv = 0;                            // unused slot of the list: any block will do, nobody reads it
        gsrc = a.Hq + ((size_t)(t0 + r) * a.nb + gbv) * 256 + ((ch - ND2) & 1) * 128 + lane * 2;
      }
      const unsigned ldst = __builtin_amdgcn_readfirstlane(dst + ch * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
    }
  };
#define PAIR_BARRIER() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

  // ---- one-time set-up: Z image, exp table, zeroed output tiles, first two records, per-lane pair constants
  if (BLK) {
    // Tile table of the group.  Tiles are numbered row by row over the upper triangle of the nb x nb grid of 16 x 16 blocks:
    // block pair (I, I) has 9 tiles (diagonals d = 0 .. 8 of the block, the last one half full), (I, J > I) has 16; pair c of
    // tile (I, J, d) is (16 I + c, 16 J + (c + d) mod 16): 16 different a and 16 different b, every unordered pair once.
    // The last block column may hold fewer than 16 inducing rows (a.rem): (I, nb - 1) then has a.rem tiles, pair c of tile d being
    // (16 I + c, 16 (nb - 1) + (c + d) mod rem) -- no tile is spent on rows that do not exist.
    if (t == 0) {
      const int nb = a.nb;
      for (int i = 0; i < PAIR_LB; ++i) gbl[i] = -1;
      int I = 0, rem = gi * (8 * NT);
      while (I < nb) { const int cnt = I < nb - 1 ? 9 + 16 * (nb - 2 - I) + a.rem : 9; if (rem < cnt) break; rem -= cnt; ++I; }
      int J = I, d = rem;
      if (I < nb && rem >= 9) {
        const int x = rem - 9, full = 16 * (nb - 2 - I);
        if (x < full) { J = I + 1 + x / 16; d = x % 16; } else { J = nb - 1; d = x - full; }
      }
      int nl = 0;
      for (int i = 0; i < 8 * NT; ++i) {
        if (I >= nb) { tinfo[i] = 0; continue; }
        int la = -1, lb = -1;
        for (int j = 0; j < nl; ++j) { if (gbl[j] == I) la = j; if (gbl[j] == J) lb = j; }
        if (la < 0) { la = nl < PAIR_LB ? nl++ : PAIR_LB - 1; gbl[la] = I; }     // (the host has checked that the list never overflows)
        if (lb < 0) { if (J == I) lb = la; else { lb = nl < PAIR_LB ? nl++ : PAIR_LB - 1; gbl[lb] = J; } }
        tinfo[i] = la | (lb << 3) | (d << 6) | ((I == J ? 1 : 0) << 10) | (1 << 11) | ((J == nb - 1 && I != J ? 1 : 0) << 12);
        if (++d == (I == J ? 9 : (J == nb - 1 ? a.rem : 16))) { d = 0; if (++J == nb) { ++I; J = I; } }
      }
    }
    __syncthreads();
    for (int i = t; i < PAIR_LB * 16 * ZS; i += 512) {
      const int l = i / ZS, q = i - l * ZS;
      const int gbv = gbl[l >> 4], m = gbv * 16 + (l & 15);
      Zl[i] = (q < QP && gbv >= 0 && m < M) ? a.Zc[(size_t)m * QP + q] : 0.0;
    }
  } else {
    for (int i = t; i < M * ZS; i += 512) {
      const int m = i / ZS, q = i - m * ZS;
      Zl[i] = (q < QP) ? a.Zc[(size_t)m * QP + q] : 0.0;
    }
  }
  if (t < PSI2_EXP_TAB) etab[t] = exp2((double)t * (1.0 / PSI2_EXP_TAB));
  for (int i = t; i < 2 * VT + 3 * 16 * TT; i += 512) v2b[i] = 0.0;
  if (nsteps > 0) request_record(0, PAIR_WAVES);
  if (nsteps > 1) request_record(1, PAIR_WAVES);
  // per-lane constants: LDS byte offsets of this lane's operands (pair of tile k in the accumulator layout: column c), the
  // pairs 4 kk + g of the tile packed for the T product's right operand, c_p dLs
  int pvalid = 0, pkt[NT][2];
  int pa8[NT], pb8[NT];                                          // 8 a, 8 b of the lane's pair of tile k: every operand address derives from them
  const int kpo = (O_KP + tb * 16 + c) * 8;
  constexpr int pso_d = (O_PS - O_KP) * 8, dlo_d = (O_DL - O_KP) * 8;     // pair sums and c_p dLs: the same slot of the next tables
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tau = gi * (8 * NT) + tb + k;
    int pa, pb;                                                  // the lane's pair of the tile: rows of the Z image (local rows in the blocked form)
    int ga, gb;                                                  // ... as inducing rows
    bool ok;
    int w4[4];
    if (BLK) {
      const int ti = tinfo[tb + k];
      const int la = ti & 7, lb = (ti >> 3) & 7, d = (ti >> 6) & 15;
      const int md = ((ti >> 12) & 1) ? a.rem : 16;             // rows of the b block
      pa = 16 * la + c; pb = 16 * lb + (c + d) % md;
      ga = 16 * gbl[la] + c; gb = 16 * gbl[lb] + (c + d) % md;
      ok = ((ti >> 11) & 1) && ga < M && gb < M && (!((ti >> 10) & 1) || d < 8 || c < 8);
      if (!ok) { ga = 0; gb = 0; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w4[kk] = (16 * la + 4 * kk + g) | ((16 * lb + (4 * kk + g + d) % md) << 8);
    } else {
      const int p = 16 * tau + c;
      pair_ab(p, M, a.P, pa, pb);
      ga = pa; gb = pb;
      ok = p < a.P;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        int xa, xb;
        pair_ab(16 * tau + 4 * kk + g, M, a.P, xa, xb);
        w4[kk] = xa | (xb << 8);
      }
    }
    pvalid |= (ok ? 1 : 0) << k;
    pa8[k] = pa * 8; pb8[k] = pb * 8;
    const double dlv = ok ? a.dLs[(size_t)ga * M + gb] * (ga == gb ? 1.0 : 2.0) : 0.0;
    double s = 0.0;
    for (int q = 0; q < QP; ++q) s += a.lam[q] * a.Zc[(size_t)ga * QP + q] * a.Zc[(size_t)gb * QP + q];
    if (g == 0) { LDSD(kpo + 128 * k) = 0.5 * s; LDSD(kpo + pso_d + 128 * k) = 0.0; LDSD(kpo + dlo_d + 128 * k) = dlv; }       // k_p
    pkt[k][0] = w4[0] | (w4[1] << 16);
    pkt[k][1] = w4[2] | (w4[3] << 16);
  }
  // The lane's operand offsets (8 c and 8 g times a stride) are recomputed from the thread index where a phase needs them instead
  // of occupying registers through the whole step.
  auto lane8 = [&](int& c8, int& g8) {
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    c8 = (tl & 15) << 3;
    g8 = (tl >> 1) & 0x18;
  };
  const int tbase = (O_TB + wave * (16 * 17)) * 8;                // this wave's transposition tile
  d4 Cacc[NT][NF1];
  double CaccR[NT][NR1];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int j = 0; j < NF1; ++j) Cacc[k][j] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int j = 0; j < NR1; ++j) CaccR[k][j] = 0.0;
  PAIR_BARRIER();

  // write-out slice of this thread: row t / 32 of the tile, columns t % 32 + 32 j.  Few loop invariants on purpose: two LDS
  // offsets, one running pointer into the sums.
  double* pvp = BLK ? nullptr : a.PV + ((size_t)t0 * 16 + (t >> 5)) * a.PW + (t & 31);     // row of tile k, advanced after B_k
  int t3 = 2;                                                    // (k - 1) mod 3 at B_k
  // After barrier B_k: V2 tile k and T tile k - 1 are added to the global sums and cleared, record k + 2 is requested.
  // The write-out is the older waves' job (PAIR_WO0): they reach the barrier early whatever they are given (the hardware favours
  // the older wave of a SIMD), so the work every thread would share after the barrier is worth more taken off the younger waves,
  // which are the ones the step waits for.  Thread t < 256 then handles rows t / 32 and t / 32 + 8 of the tiles.
  auto after_barrier = [&](int k) {
    if (PAIR_WO0 && role1) return;
    constexpr int NH = PAIR_WO0 ? 2 : 1;
    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));                                 // the slice's offsets are recomputed here, not kept across the step
    if (BLK) {
      // blocked form (threads 0-255 only, PAIR_WO0): V2 tile k is [local block][16 rows][16]; thread tq takes element tq of every
      // block (row tq / 16, column tq % 16) to the block's columns of V2[N][Mld]; T tile k - 1 as below, to T[N][QP]
      static_assert(!BLK || PAIR_WO0, "the blocked form's write-out is written for waves 0-3");
      const int64_t row0 = (t0 + k) * 16;
      const int64_t left = a.N - row0;
      if (k < nsteps) {
        double* const vt = v2b + (k & 1) * VT + tq;
        const int row = tq >> 4;
        double v[PAIR_LB];
#pragma unroll
        for (int j = 0; j < PAIR_LB; ++j) v[j] = vt[256 * j];
        double* const dst = a.V2 + (row0 + row) * a.Mld + (tq & 15);
#pragma unroll
        for (int j = 0; j < PAIR_LB; ++j) {
          vt[256 * j] = 0.0;
          const int gbv = __builtin_amdgcn_readfirstlane(gbl[j]);
          if (gbv >= 0 && row < left) atomicAdd(&dst[16 * gbv], v[j]);
        }
      }
      if (k >= 1) {
        const int wrow0 = tq >> 5, wcol = tq & 31;
        const int tslots = (QP - wcol + 31) >> 5;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int wrow = wrow0 + 8 * h;
          double* const trow = ttb + t3 * (16 * TT) + wrow * TT + wcol;
          double* const prow = a.Tg + (row0 - 16 + wrow) * QP + wcol;
          double v[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) v[j] = (j < tslots) ? trow[32 * j] : 0.0;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j < tslots) {
              trow[32 * j] = 0.0;
              if (wrow < left + 16) atomicAdd(&prow[32 * j], v[j]);
            }
        }
      }
      t3 = (t3 == 2) ? 0 : t3 + 1;
      if (k + 2 < nsteps) request_record(k + 2, 4);
      return;
    }
    const int wrow0 = tq >> 5, wcol = tq & 31;
    const int vslots = (a.M2 - wcol + 31) >> 5;                  // column slots of this thread in a V2 row (0..4)
    const int tslots = (QP - wcol + 31) >> 5;                    // ... in a T row (0..2)
    const int64_t left = a.N - (t0 + k) * 16;                    // valid rows of tile k
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int wrow = wrow0 + 8 * h;
      const int woff = wrow * HS + wcol, toff = wrow * TT + wcol;
      double* const pv = pvp + 8 * h * (int64_t)a.PW;
      if (k < nsteps) {
        double* const vrow = v2b + (k & 1) * (16 * HS) + woff;
        const bool rok = wrow < left;
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (j < vslots) ? vrow[32 * j] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < vslots) {
            vrow[32 * j] = 0.0;
            if (rok) atomicAdd(&pv[32 * j], v[j]);
          }
      }
      if (k >= 1) {
        double* const trow = ttb + t3 * (16 * TT) + toff;
        double* const prow = pv - 16 * (int64_t)a.PW + a.M2;
        double v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) v[j] = (j < tslots) ? trow[32 * j] : 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < tslots) {
            trow[32 * j] = 0.0;
            if (wrow < left + 16) atomicAdd(&prow[32 * j], v[j]);  // rows of tile k - 1 that exist
          }
      }
    }
    pvp += 16 * (int64_t)a.PW;
    t3 = (t3 == 2) ? 0 : t3 + 1;
    if (k + 2 < nsteps) request_record(k + 2, PAIR_WO0 ? 4 : PAIR_WAVES);     // last: no register reload may follow it before the next barrier
  };

#ifdef PAIR_STAMPS
  long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long sq_[13], sprev_ = __builtin_readcyclecounter();
  const unsigned long long clk0_ = sprev_, wall0_ = wall_clock64();       // core clock against the constant 100 MHz counter
#endif
  int tt3 = 0;                                                   // s mod 3
  for (int s = 0; s < nsteps; ++s) {
    const int inB = (O_INB + (s & 1) * RSP) * 8;                 // d2 tile [16][TS] of this step's record
    const int hB = inB + HOFF * 8;                               // h' tile [16][HS]   (blocked form: [block][16][16])
    const int vB = (O_V2 + (s & 1) * VT) * 8;                    // V2 tile, same shape
    const int tBs = (O_TT + tt3 * (16 * TT)) * 8;                // T tile [16][TT]
    tt3 = (tt3 == 2) ? 0 : tt3 + 1;
    // ======== GE(s) ========
    // group 0: the reads of the step's head in one batch: -2 k_p and the 32 h' values; the first Gram operands follow once half of
    // them have been consumed.  The accumulators start at h'[n,a] + h'[n,b] + k_p and the record holds -d2 / 2: E = acc after the Gram product, no
    // LDS round trip between the product and the exponential
    double za[NT], zb[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { asm volatile("" : "+v"(pa8[k]), "+v"(pb8[k])); }      // the derived addresses are recomputed per step, not kept
    int c8, g8;
    lane8(c8, g8);
    const int gH = g8 * HR, zc = O_Z * 8 + g8, aoff = c8 * TS + g8;
    d4 acc[NT];
    double A0, A1;
    int zoA[NT], zoB[NT];
    // two pair tiles at a time (the 36 reads of all four at once would not fit the registers next to the accumulators)
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += 2) {
      double kpv[2], hA[2][4], hBv[2][4];
#pragma unroll
      for (int k = k0; k < k0 + 2 && k < NT; ++k) {
        kpv[k - k0] = LDSD(kpo + 128 * k);
        int oa = hcol(pa8[k]) + gH + hB, ob = hcol(pb8[k]) + gH + hB;
        asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
        for (int r = 0; r < 4; ++r) { hA[k - k0][r] = LDSD(oa + r * (4 * HR * 8)); hBv[k - k0][r] = LDSD(ob + r * (4 * HR * 8)); }
      }
      if (k0 + 2 >= NT) {                    // behind the last h' reads: the first Gram operands
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          zoA[k] = pa8[k] * ZS + zc; zoB[k] = pb8[k] * ZS + zc;
          asm volatile("" : "+v"(zoA[k]), "+v"(zoB[k]));         // one register per tile side; the k-step goes into the instruction's offset
        }
        A0 = LDSD(inB + aoff); A1 = (KQ > 1) ? LDSD(inB + aoff + 32) : 0.0;
#pragma unroll
        for (int k = 0; k < NT; ++k) { za[k] = LDSD(zoA[k]); zb[k] = LDSD(zoB[k]); }
      }
      PAIR_SB();
#pragma unroll
      for (int k = k0; k < k0 + 2 && k < NT; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[k][r] = (hA[k - k0][r] + hBv[k - k0][r]) + kpv[k - k0];
      PAIR_SB();
    }
    PSTAMP(0);
    // Gram: acc[k][16 rows x 16 pairs] += sum_q d2[n,q] z[a,q] z[b,q]; the operands of k-step kq + 1 are requested before the
    // matrix instructions of k-step kq are issued
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      double zan[NT], zbn[NT], A2 = 0.0;
      if (kq + 1 < KQ) {
#pragma unroll
        for (int k = 0; k < NT; ++k) { zan[k] = LDSD(zoA[k] + 32 * (kq + 1)); zbn[k] = LDSD(zoB[k] + 32 * (kq + 1)); }
      }
      if (kq + 2 < KQ) A2 = LDSD(inB + aoff + 32 * (kq + 2));
      double zz[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) zz[k] = za[k] * zb[k];
      PAIR_SB();
#pragma unroll
      for (int k = 0; k < NT; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, zz[k], acc[k], 0, 0, 0);
      PAIR_SB();
      if (kq + 1 < KQ) {
#pragma unroll
        for (int k = 0; k < NT; ++k) { za[k] = zan[k]; zb[k] = zbn[k]; }
      }
      A0 = A1; A1 = A2;
    }
    PSTAMP(1);
    // the 16 exponentials: range reduction and table read of both halves first, then the polynomials
    double w[NT][4], tj[NT][4], dl2[NT];
    int sh[NT][4];
#pragma unroll
    for (int k = 0; k < NT; ++k) dl2[k] = LDSD(kpo + dlo_d + 128 * k);
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += 2) {
#pragma unroll
      for (int k = k0; k < k0 + 2 && k < NT; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double x = acc[k][r];
          x = clamp_exp_arg(x);                                              // NaN stays NaN
          const double tt = fma(x, 0x1.71547652b82fep+8, 0x1.8p52);
          const int n = __double2loint(tt);
          const double nf = tt - 0x1.8p52;
          double rr_ = fma(nf, -0x1.62e42fee00000p-9, x);
          rr_ = fma(nf, -0x1.a39ef35793c76p-41, rr_);
          w[k][r] = rr_;
          sh[k][r] = n >> 8;
          tj[k][r] = LDSD(O_ETAB * 8 + ((n & (PSI2_EXP_TAB - 1)) << 3));
        }
      PAIR_SB();
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      double ps = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double rr_ = w[k][r];
        double q = fma_rrr(rr_, 0x1.5555555555555p-5, 0x1.5555555555555p-3);
        q = fma_rrr(q, rr_, 0.5);
        q = fma_rrr(q, rr_, 1.0);
        q = q * rr_;
        const double e = ldexp(fma(tj[k][r], q, tj[k][r]), sh[k][r]);
        ps += e;
        w[k][r] = e * dl2[k];
      }
      atomicAdd(LDSP(kpo + pso_d + 128 * k), ps);
    }
    PAIR_SB();
    // right operand of the C product: requested now, it arrives under the V2 adds
    double Bc[4][NF1], BcR[4][NR1];
    {
      int c8b, g8b;
      lane8(c8b, g8b);
      int boff = g8b * TS + c8b + inB;                           // C product: right operand d2[row 4 r + g][16 j + c]
      asm volatile("" : "+v"(boff));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NF; ++j) Bc[r][j] = LDSD(boff + (4 * r * TS + 16 * j) * 8);
      if (NR > 0) {
        int boffR = g8b * TS + (c8b & 0x18) + inB;               // ... of the groups of 4 columns: d2[row 4 r + g][16 NF + 4 j + (c & 3)]
        asm volatile("" : "+v"(boffR));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NR; ++j) BcR[r][j] = LDSD(boffR + (4 * r * TS + 16 * NF + 4 * j) * 8);
      }
    }
    // V2 adds in one batch (fire and forget: they drain under the C product)
    int gHv;
    { int c8v, g8v; lane8(c8v, g8v); gHv = g8v * HR + vB; }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      int oa = hcol(pa8[k]) + gHv, ob = hcol(pb8[k]) + gHv;
      asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomicAdd(LDSP(oa + r * (4 * HR * 8)), w[k][r]);
        atomicAdd(LDSP(ob + r * (4 * HR * 8)), w[k][r]);
      }
    }
    PAIR_SB();
    PSTAMP(2);
    // ======== C(s): C[k][16 pairs x 16 q] += w^T d2 (K = the step's 16 rows).  Behind the first quarter of its matrix
    // instructions the first pair tile's weights go through the transposition tile and the first z values of the T product are
    // requested ========
    auto c_quarter = [&](int r) {
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int k = 0; k < NT; ++k) Cacc[k][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[k][r], Bc[r][j], Cacc[k][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int k = 0; k < NT; ++k) CaccR[k][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(w[k][r], BcR[r][j], CaccR[k][j], 0, 0, 0);
    };
    // z values of k-step kk of pair tile k for the T product: rows of the pair 4 kk + g, columns 16 j + c
    // (columns 16 j + c for the full tiles j < NF, then 16 NF + 4 j' + (c & 3) for the groups of 4: entries NF + j')
    auto t_reads = [&](int zto, int ztoR, int k, int kk, double (&zra)[NE], double (&zrb)[NE]) {
      const int pk = (pkt[k][kk >> 1] >> (16 * (kk & 1))) & 0xffff;
      const int ra = (pk & 0xff) * (ZS * 8), rb = (pk >> 8) * (ZS * 8);
      if (NF > 0) {
        int oa = ra + zto, ob = rb + zto;
        asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
        for (int j = 0; j < NF; ++j) { zra[j] = LDSD(oa + 128 * j); zrb[j] = LDSD(ob + 128 * j); }
      }
      if (NR > 0) {
        int oa = ra + ztoR, ob = rb + ztoR;
        asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
        for (int j = 0; j < NR; ++j) { zra[NF + j] = LDSD(oa + (16 * NF + 4 * j) * 8); zrb[NF + j] = LDSD(ob + (16 * NF + 4 * j) * 8); }
      }
    };
    int tbw, tbr, zto, ztoR;
    {
      int c8t, g8t;
      lane8(c8t, g8t);
      tbw = tbase + g8t * 17 + c8t;                              // transposition tile: write [4 r + g][c]
      tbr = tbase + c8t * 17 + g8t;                              //                     read  [c][4 kk + g]
      zto = O_Z * 8 + c8t;                                       // Z image, T product's right operand: [pair's row][16 j + c]
      ztoR = O_Z * 8 + (c8t & 0x18);                             //                     ... of the groups of 4: [pair's row][.. + (c & 3)]
    }
    c_quarter(0);
    PAIR_SB();
    PSTAMP(11);
    double at[4], zra[NE], zrb[NE];
#pragma unroll
    for (int r = 0; r < 4; ++r) LDSD(tbw + r * (4 * 17 * 8)) = w[0][r];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) at[kk] = LDSD(tbr + kk * 32);
    t_reads(zto, ztoR, 0, 0, zra, zrb);
    PAIR_SB();
    PSTAMP(12);
    c_quarter(1);
    c_quarter(2);
    c_quarter(3);
    PAIR_SB();
    PSTAMP(3);
    if (role1) {
      PAIR_BARRIER();                        // B_s for waves 4-7: between the C and the T product
      PSTAMP(5);
      after_barrier(s);
      PSTAMP(6);
    }
    // ======== T(s): T[16 rows x 16 q] += w (z z)^T, K = a tile's 16 pairs, one k-step (four pairs) at a time; the operands of the
    // next k-step (and, three k-steps into a tile, the next tile's transposed weights) are requested before the current one is
    // issued ========
    d4 Tacc[NF1];
    double TaccR[NR1];
#pragma unroll
    for (int j = 0; j < NF1; ++j) Tacc[j] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < NR1; ++j) TaccR[j] = 0.0;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      double atn[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double zna[NE], znb[NE];
        const bool last = (k == NT - 1 && kk == 3);
        if (kk == 2 && k + 1 < NT) {           // the tile's own transposed weights are in registers: the tile may be overwritten
#pragma unroll
          for (int r = 0; r < 4; ++r) LDSD(tbw + r * (4 * 17 * 8)) = w[k + 1][r];
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) atn[k2] = LDSD(tbr + k2 * 32);
        }
        if (kk < 3) t_reads(zto, ztoR, k, kk + 1, zna, znb);
        else if (!last) t_reads(zto, ztoR, k + 1, 0, zna, znb);
        double zz[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) zz[j] = zra[j] * zrb[j];
        PAIR_SB();
#pragma unroll
        for (int j = 0; j < NF; ++j) Tacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[kk], zz[j], Tacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NR; ++j) TaccR[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(at[kk], zz[NF + j], TaccR[j], 0, 0, 0);
        PAIR_SB();
        if (!last) {
#pragma unroll
          for (int j = 0; j < NE; ++j) { zra[j] = zna[j]; zrb[j] = znb[j]; }
        }
      }
      if (k + 1 < NT) {
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) at[k2] = atn[k2];
      }
      if (k < 4) PSTAMP(7 + k);
    }
    {
      int c8f, g8f;
      lane8(c8f, g8f);
      int tB = tBs + g8f * TT + c8f;                             // this lane's corner: [4 r + g][16 j + c]
      asm volatile("" : "+v"(tB));
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(LDSP(tB + (4 * r * TT + 16 * j) * 8), Tacc[j][r]);
      if (NR > 0) {                                              // the groups of 4 columns: this lane holds T[4 (c >> 2) + g][.. + (c & 3)]
        int tR = tBs + ((c8f >> 5) * 4 * 8 + g8f) * TT + (c8f & 0x18);
        asm volatile("" : "+v"(tR));
#pragma unroll
        for (int j = 0; j < NR; ++j) atomicAdd(LDSP(tR + (16 * NF + 4 * j) * 8), TaccR[j]);
      }
    }
    PSTAMP(4);
    if (!role1) {
      PAIR_BARRIER();                        // B_s for waves 0-3
      PSTAMP(5);
      after_barrier(s);
      PSTAMP(6);
    }
#ifdef PAIR_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sq_[0]), "+s"(sq_[1]), "+s"(sq_[2]), "+s"(sq_[3]), "+s"(sq_[4]), "+s"(sq_[5]), "+s"(sq_[6]),
                 "+s"(sq_[7]), "+s"(sq_[8]), "+s"(sq_[9]), "+s"(sq_[10]), "+s"(sq_[11]), "+s"(sq_[12]));
    for (int i = 0; i < 13; ++i) st_[i] += (long long)(sq_[i] - sprev_);         // offsets from the step's start
    sprev_ = role1 ? sq_[4] : sq_[6];
#endif
  }
  // role 1's T adds of the last step came after the last barrier: one more, then the last T tile leaves
  PAIR_BARRIER();
  after_barrier(nsteps);

#ifdef PAIR_STAMPS
  if (a.dbg && lane == 0) {
    st_[15] = nsteps;
    st_[13] = (long long)(__builtin_readcyclecounter() - clk0_);
    st_[14] = (long long)(wall_clock64() - wall0_);
    for (int i = 0; i < 16; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + i] = (double)st_[i];
  }
#endif
  // ---- flush: C partial of this (row range, pair group), pair sums of psi2
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tau = gi * (8 * NT) + tb + k;
    if (BLK) {                                    // the row ranges add into one array in tile order
      if (tau >= a.ntl) continue;
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&a.Csum[((size_t)16 * tau + 4 * r + g) * QP + 16 * j + c], Cacc[k][j][r]);
#pragma unroll
      for (int j = 0; j < NR; ++j)
        atomicAdd(&a.Csum[((size_t)16 * tau + 4 * (c >> 2) + g) * QP + 16 * NF + 4 * j + (c & 3)], CaccR[k][j]);
      const double psb = LDSD(kpo + pso_d + 128 * k);
      if (g == 0 && ((pvalid >> k) & 1)) atomicAdd(&a.p2sum[16 * tau + c], psb);
      continue;
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int q = 16 * j + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) a.Cpart[((size_t)rr * a.Ppad + 16 * tau + 4 * r + g) * QP + q] = Cacc[k][j][r];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j)                  // the groups of 4 columns: this lane holds C[pair 4 (c >> 2) + g][16 NF + 4 j + (c & 3)]
      a.Cpart[((size_t)rr * a.Ppad + 16 * tau + 4 * (c >> 2) + g) * QP + 16 * NF + 4 * j + (c & 3)] = CaccR[k][j];
    const double ps = LDSD(kpo + pso_d + 128 * k);
    if (g == 0 && ((pvalid >> k) & 1)) {
      int pa, pb;
      pair_ab(16 * tau + c, M, a.P, pa, pb);
      const int lo = pa < pb ? pa : pb, hi = pa < pb ? pb : pa;
      a.p2_part[(size_t)rr * a.tri + hi * (hi + 1) / 2 + lo] = ps;
    }
  }
}

template <int QP>
__global__ void __launch_bounds__(512, 2) psi2_pair_kernel(const PairArgs a) {
  const int gi = blockIdx.x % a.G, rr = blockIdx.x / a.G;
  if (threadIdx.x < 256) psi2_pair_body<QP, pair_nt(QP), false, false>(a, gi, rr);
  else psi2_pair_body<QP, pair_nt(QP), true, false>(a, gi, rr);
}

// Blocked form: the (pair group, row range) tasks outnumber the CUs; a workgroup takes every gridDim.x-th one.  Consecutive
// tasks are the pair groups of one row range, so the workgroups running at any time share the range's records in L2.
template <int QP>
__global__ void __launch_bounds__(512, 2) psi2_pairblk_kernel(const PairArgs a) {
  for (int task = blockIdx.x; task < a.ntasks; task += gridDim.x) {
    const int gi = task % a.G, rr = task / a.G;       // (row range fastest instead: 1341 against 1332 ms per config-5 step)
    if (threadIdx.x < 256) psi2_pair_body<QP, pair_nt(QP), false, true>(a, gi, rr);
    else psi2_pair_body<QP, pair_nt(QP), true, true>(a, gi, rr);
    __syncthreads();                              // the next task's set-up overwrites the tables and tiles
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Input records of the main kernel, one per tile of 16 data rows: [-d2 / 2, 16 x TS (zero padded) | h' 16 x HS] with
//   h'[n,m] = sum_q (d2 mu)[n,q] z[m,q] - 1/4 sum_q d2[n,q] z[m,q]^2 - 1/4 sum_q lam_q z[m,q]^2 + g0[n] / 2,
// a [16 x 2 QP] x [2 QP x M] product on the matrix pipe: wave = one tile of rows, left operands (d2 mu | d2) straight from
// global memory into registers, right operands (z | -z^2 / 4) from two LDS images of Z.  Rows beyond N get d2 = 0 and
// h' = -inf (their exponentials are 0).  The kernel also zeroes the tile's rows of the buffer the pair groups add into.
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ void __launch_bounds__(256, 2) psi2_pair_prep_kernel(const int64_t N, const int64_t ntiles, const int M, const int HS, const int RS,
                                                                 const double* __restrict__ Zc, const double* __restrict__ lam,
                                                                 const double* __restrict__ Xmu, const double* __restrict__ d2,
                                                                 const double* __restrict__ g0, double* __restrict__ rec,
                                                                 double* __restrict__ PV, const int PW) {
  constexpr int QT = (QP + 15) / 16, TS = 16 * QT + 1, ZQ = QP + 1, KQ = QP / 4;
  extern __shared__ double sm[];
  const int MT = (M + 15) >> 4;
  double* const Zs = sm;                          // [16 MT][ZQ]   z
  double* const Z2s = Zs + 16 * MT * ZQ;          // [16 MT][ZQ]   -z^2 / 4
  double* const el = Z2s + 16 * MT * ZQ;          // [16 MT]       1/4 sum_q lam z^2
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 16 * MT * ZQ; i += 256) {
    const int m = i / ZQ, q = i - m * ZQ;
    const double z = (m < M && q < QP) ? Zc[(size_t)m * QP + q] : 0.0;
    Zs[i] = z;
    Z2s[i] = -0.25 * z * z;
  }
  __syncthreads();
  for (int m = t; m < 16 * MT; m += 256) {
    double e = 0.0;
    for (int q = 0; q < QP; ++q) e -= lam[q] * Z2s[m * ZQ + q];
    el[m] = e;
  }
  __syncthreads();
  // left operands of the lane: row n0 + c, columns 4 kk + g; the next tile's are requested before the current tile's products
  double dn[KQ], mn[KQ];
  auto fetch = [&](const int64_t tile) {
    const int64_t n = tile * 16 + c;
    const bool ok = n < N;
    const size_t o = (size_t)(ok ? n : N - 1) * QP + g;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) { dn[kk] = ok ? d2[o + 4 * kk] : 0.0; mn[kk] = Xmu[o + 4 * kk]; }
  };
  const int64_t tile0 = (int64_t)blockIdx.x * 4 + w, tstep = (int64_t)gridDim.x * 4;
  if (tile0 < ntiles) fetch(tile0);
  for (int64_t tile = tile0; tile < ntiles; tile += tstep) {
    const int64_t n0 = tile * 16;
    double* const r = rec + (size_t)tile * RS;
    {                                                      // the pair groups ADD their sums over pairs into this tile's rows
      const int64_t lim = (N - n0 < 16 ? N - n0 : 16) * PW;
      for (int i = lane; i < lim; i += 64) PV[n0 * PW + i] = 0.0;
    }
    double Adm[KQ], Ad[KQ];
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      Ad[kk] = dn[kk];
      Adm[kk] = dn[kk] * mn[kk];
      r[c * TS + 4 * kk + g] = -0.5 * dn[kk];                // -d2 / 2: the Gram product of the main kernel then yields the exponent itself
    }
    if (tile + tstep < ntiles) fetch(tile + tstep);
    for (int i = lane; i < 16 * (TS - QP); i += 64) {      // zero padding of the d2 tile
      const int row = i / (TS - QP), col = QP + (i - row * (TS - QP));
      r[row * TS + col] = 0.0;
    }
    double gs[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) gs[rr] = (n0 + 4 * rr + g < N) ? 0.5 * g0[n0 + 4 * rr + g] : -INFINITY;
    for (int mt = 0; mt < MT; ++mt) {
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      const int zo = (16 * mt + c) * ZQ + g;
#pragma unroll
      for (int kk = 0; kk < KQ; ++kk) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Adm[kk], Zs[zo + 4 * kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad[kk], Z2s[zo + 4 * kk], acc, 0, 0, 0);
      }
      const int m = 16 * mt + c;
      const double em = el[m];
      if (m < HS) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) r[16 * TS + (4 * rr + g) * HS + m] = (m < M) ? (acc[rr] - em) + gs[rr] : 0.0;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Row kernel, on the matrix pipe.  A workgroup walks super-tiles of 32 data rows; v = V2 / 2 and (d2 mu | d2) of the
// super-tile are staged in LDS.
//   job B  (wave w owns the inducing-row tiles w and w + 4): [B1 | B2] += v^T (d2 mu | d2), accumulated in registers over all
//           the workgroup's super-tiles; at the end the slab dz[q][m] = 2 B1 - z B2.
//   job A  (wave = (tile of 16 rows, Z or Z^2)): a1 = v Z or vz2 = v Z^2, right operands from an LDS image of the first 4 KL
//           rows of Z -- as many as fit next to the tiles with two workgroups per CU -- and from global memory for the rest
//           (with all of them from global memory a quarter of the kernel's time was the wait for those loads), the row sums
//           of v on the way; the results go to LDS over the (d2 mu | d2) image.
//   epilogue (thread = (window column, every RPP-th row)): the row gradients from a1, vz2, T and the row sums, dinv_l and
//           sum-of-weights partials.
// ------------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ void __launch_bounds__(256, 2) psi2_pair_rows_kernel(const int64_t N, const int M, const int Q, const int PW, const int M2, const int KL,
                                                                 const double* __restrict__ PV, const double* __restrict__ Zc,
                                                                 const double* __restrict__ lam, const double* __restrict__ Xmu,
                                                                 const double* __restrict__ d2, double* __restrict__ gmu,
                                                                 double* __restrict__ gS, const int accumulate,
                                                                 double* __restrict__ dz_part, double* __restrict__ dlam_part,
                                                                 double* __restrict__ wsum_part) {
  constexpr int QT = (QP + 15) / 16, CT = 2 * QT, DS = 16 * CT + 1, RPP = 256 / QP;
  extern __shared__ double sm[];
  const int MT = (M + 15) >> 4, VW = 16 * MT, VS = VW + 1, K4 = (M + 3) >> 2;
  double* const vS = sm;                          // [32][VS]  v = V2 / 2, zero beyond M and beyond N
  double* const dS = vS + 32 * VS;                // [32][DS]  (d2 mu | d2), each half padded to 16 QT; later (a1 | vz2)
  double* const wsS = dS + 32 * DS;               // [32]      row sums of v
  constexpr int ZQ = QP + 1;
  double* const Zl = wsS + 32;                    // [4 KL][ZQ]  the first 4 KL rows of Z
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t nst = (N + 31) >> 5;
  const bool two = w + 4 < MT;                    // this wave owns a second tile of inducing rows
  for (int i = t; i < 4 * KL * ZQ; i += 256) {    // (the first barrier of the loop over the super-tiles orders these writes)
    const int m = i / ZQ, q = i - m * ZQ;
    Zl[i] = (m < M && q < QP) ? Zc[(size_t)m * QP + q] : 0.0;
  }
  d4 Bacc[2][CT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) Bacc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int eq = t % QP, er = t / QP;             // epilogue identity
  double dl_acc = 0.0, ws_acc = 0.0;
  const int rt = w & 1, ch = w >> 1;              // job A identity

  // The next super-tile's rows are requested from global memory before the current one is worked on and go to LDS after it:
  // thread t holds elements t, t + 256, ... of the [32][VW] image of V2.
  constexpr int NV = 14, ND = 2 * CT;             // 32 VW / 256 = 2 MT <= 14;  32 * 16 CT / 256
  double pvn[NV];
  // (the element offsets are recomputed from an opaque copy of the thread index wherever they are used: hoisted out of the
  // loop over the super-tiles they would occupy some sixty registers next to the accumulators)
  auto fetch = [&](const int64_t st) {
    const int64_t n0 = st * 32;
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tt + 256 * k;
      const int row = i / VW, col = i - row * VW;
      const int64_t n = n0 + row;
      pvn[k] = (k < 2 * MT && col < M2 && n < N) ? PV[(size_t)n * PW + col] : 0.0;
    }
  };
  if ((int64_t)blockIdx.x < nst) fetch(blockIdx.x);
  for (int64_t st = blockIdx.x; st < nst; st += gridDim.x) {
    const int64_t n0 = st * 32;
    __syncthreads();                              // the previous epilogue is done with dS and wsS
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (k < 2 * MT) {
        const int i = tt + 256 * k;
        const int row = i / VW, col = i - row * VW;
        vS[row * VS + col] = 0.5 * pvn[k];
      }
#pragma unroll
    for (int k = 0; k < ND; ++k) {                 // (d2 mu | d2): not prefetched (no gain measured), the registers go to the accumulators
      const int i = tt + 256 * k;
      const int row = i / (16 * CT), col = i - row * (16 * CT);
      const int qq = col >= 16 * QT ? col - 16 * QT : col;
      const int64_t n = n0 + row;
      double v = 0.0;
      if (qq < QP && n < N) {
        const double dv = d2[(size_t)n * QP + qq];
        v = col < 16 * QT ? dv * Xmu[(size_t)n * QP + qq] : dv;
      }
      dS[row * DS + col] = v;
    }
    __syncthreads();
    if (st + gridDim.x < nst) fetch(st + gridDim.x);
    // ---- job B: K = the super-tile's 32 rows
    if (w < MT) {
#pragma unroll 2
      for (int kk = 0; kk < 8; ++kk) {
        const int ro = 4 * kk + g;
        const double a0 = vS[ro * VS + 16 * w + c];
        const double a1 = two ? vS[ro * VS + 16 * (w + 4) + c] : 0.0;
        double b[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) b[j] = dS[ro * DS + 16 * j + c];
#pragma unroll
        for (int j = 0; j < CT; ++j) Bacc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[j], Bacc[0][j], 0, 0, 0);
        if (two) {
#pragma unroll
          for (int j = 0; j < CT; ++j) Bacc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[j], Bacc[1][j], 0, 0, 0);
        }
      }
    }
    // ---- job A: K = the inducing rows
    d4 Aacc[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) Aacc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    double wsum = 0.0;
    {
      const double* vr = vS + (16 * rt + c) * VS + g;
      const double* zl = Zl + g * ZQ + c;
#pragma unroll 2
      for (int kk = 0; kk < KL; ++kk) {
        const double a = vr[4 * kk];
        wsum += a;
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const double z = (16 * j + c < QP) ? zl[4 * kk * ZQ + 16 * j] : 0.0;
          Aacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ch ? z * z : z, Aacc[j], 0, 0, 0);
        }
      }
#pragma unroll 2
      for (int kk = KL; kk < K4; ++kk) {
        const double a = vr[4 * kk];
        wsum += a;
        const int m = 4 * kk + g;
        const double* zr = Zc + (size_t)(m < M ? m : M - 1) * QP + c;
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const double z = (16 * j + c < QP) ? zr[16 * j] : 0.0;
          Aacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ch ? z * z : z, Aacc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                              // every wave is done with (d2 mu | d2): job A's results take its place
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dS[(16 * rt + 4 * r + g) * DS + 16 * (ch * QT + j) + c] = Aacc[j][r];
    if (ch == 0) {
      wsum += __shfl_xor(wsum, 16);
      wsum += __shfl_xor(wsum, 32);
      if (g == 0) wsS[16 * rt + c] = wsum;
    }
    __syncthreads();
    // ---- epilogue
    if (er < RPP) {
      const double il = 1.0 / lam[eq < Q ? eq : 0];
      for (int row = er; row < 32; row += RPP) {
        const int64_t n = n0 + row;
        if (n >= N) break;
        const size_t o = (size_t)n * QP + eq;
        if (eq < Q) {
          const double a1 = dS[row * DS + eq], vz2 = dS[row * DS + 16 * QT + eq], ws = wsS[row];
          const double Tt = PV[(size_t)n * PW + M2 + eq], mu = Xmu[o], dq = d2[o];
          const double a23 = 0.5 * (vz2 + Tt);
          const double dmu = 2.0 * dq * (a1 - mu * ws);
          const double ddq = -mu * mu * ws + 2.0 * mu * a1 - a23 + 0.5 * ws / dq;
          const double dSv = -2.0 * dq * dq * ddq;
          if (accumulate) { atomicAdd(&gmu[o], dmu); atomicAdd(&gS[o], dSv); }      // (no return value: nothing waits for the old sums)
          else { gmu[o] = dmu; gS[o] = dSv; }
          dl_acc += ddq * (dq * il) * (dq * il) - 0.5 * ws * il;
          if (eq == 0) ws_acc += ws;
        } else if (!accumulate) {
          gmu[o] = 0.0; gS[o] = 0.0;
        }
      }
    }
  }
  // slab of this workgroup: dz[q][m] = 2 B1[m,q] - z[m,q] B2[m,q]   (layout [QP][128]; rows of inducing points beyond the tiles: 0)
  {
    double* slab = dz_part + (size_t)blockIdx.x * QP * RGP_PSI2_TEAM;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mt = w + 4 * i;
      if (mt < MT) {
#pragma unroll
        for (int j = 0; j < QT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * mt + 4 * r + g, q = 16 * j + c;
            if (q < QP) slab[q * RGP_PSI2_TEAM + m] = (m < M) ? 2.0 * Bacc[i][j][r] - Zc[(size_t)m * QP + q] * Bacc[i][j + QT][r] : 0.0;
          }
      }
    }
    const int rest = RGP_PSI2_TEAM - VW;
    for (int i = t; i < QP * rest; i += 256) {
      const int q = i / rest, m = VW + (i - q * rest);
      slab[q * RGP_PSI2_TEAM + m] = 0.0;
    }
  }
  __syncthreads();
  if (er < RPP) {
    dS[er * QP + eq] = dl_acc;
    if (eq == 0) wsS[er] = ws_acc;
  }
  __syncthreads();
  if (t < QP) {
    double s = 0.0;
    for (int r = 0; r < RPP; ++r) s += dS[r * QP + t];
    dlam_part[(size_t)blockIdx.x * QP + t] = (t < Q) ? s : 0.0;
  }
  if (t == 0) {
    double s = 0.0;
    for (int r = 0; r < RPP; ++r) s += wsS[r];
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
#pragma unroll 8
  for (int mp = 0; mp < M; ++mp) {
    int d = mp - m, a = m;
    if (d < 0) d += M;
    if (2 * d > M || (2 * d == M && mp < m)) { d = M - d; a = mp; }
    const int p = d * M + a;
    const double cv = Csum[(size_t)p * QP + q];
    s += Zc[(size_t)mp * QP + q] * (mp == m ? cv : 0.5 * cv);
  }
  slab[q * RGP_PSI2_TEAM + m] = 2.0 * s;                  // the C sums were taken over -d2 / 2
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------

bool psi2_pair_geom(rgp_handle h, int64_t N, int M, int QP, PairGeom& g) {
  if (M < 2 || M > 112 || QP < 8 || QP > RGP_MAX_QP || (QP & 3) || N < 1) return false;
  g.M = M; g.QP = QP;
  g.QT = (QP + 15) / 16;
  g.TS = 16 * g.QT + 1;
  g.HS = PAIR_HS;
  g.OS = ((16 * g.QT) % 32 == 16) ? 16 * g.QT : 16 * g.QT + 16;      // row stride of the T tiles (= 16 mod 32)
  g.RS = (16 * (g.TS + g.HS) + 127) / 128 * 128;                       // record length: whole 1-KiB pieces
  g.M2 = (M + 1) & ~1;
  g.PW = g.M2 + QP;
  g.P = M * (M + 1) / 2;
  const int group = 128 * pair_nt(QP);                                  // pairs per workgroup
  g.G = (g.P + group - 1) / group;
  g.Ppad = g.G * group;
  g.ntiles = (N + 15) / 16;
  // One workgroup per CU, all resident at once: a workgroup that finds its CU taken by a kernel of the side streams would run
  // as a second round and double the launch's duration, so a few CUs are left to those kernels.
  // A short pass (few steps per workgroup) leaves more: the side-stream kernels of the running step -- the previous layer's
  // finalisation and update, the next layer's chain -- are then on the critical path and must not sit out the pass (measured at
  // M = 100 on 256 CUs, steps/s with 250 / 240 / 230 / 220 / 210 workgroups: 4 sequences 354 / 374 / 367 / 395 / 386, 8 sequences
  // 223 / 233 / 227 / 228 / 221, 16 sequences 128.9 / 126.3 with 250 / 240).
  int spare = 0;
  if (h->overlap && h->num_cu > 64) {
    const int64_t steps = g.ntiles / ((h->num_cu - 6) / g.G > 0 ? (h->num_cu - 6) / g.G : 1);
    spare = steps >= 128 ? 6 : (steps >= 64 ? 16 : 36);
  }
  if (const char* e = getenv("RGP_PSI2_PAIR_SPARE")) spare = atoi(e);
  int64_t rr = (h->num_cu - spare) / g.G;
  if (rr < 1) rr = 1;
  if (rr > g.ntiles) rr = g.ntiles;
  g.RR = (int)rr;
  int64_t nb = 2 * (int64_t)h->num_cu;
  if (nb > g.ntiles) nb = g.ntiles;
  g.NB = (int)nb;
  const size_t zs = (size_t)M * g.TS;
  g.lds = (zs + PSI2_EXP_TAB + 2 * (size_t)g.RS + 2 * 16 * (size_t)g.HS + 3 * 16 * (size_t)g.OS + PAIR_WAVES * 16 * 17 + 3 * (size_t)group) * sizeof(double);
  return g.lds <= 160 * 1024;
}

// The pair form pays once a workgroup's 512 or 640 pairs are mostly real and every workgroup has a few dozen steps of 16 rows to walk.
// Measured against the sweep kernels at M = 100, Q = 40 / 40 / 20 (steps/s of the whole SVI step, pair : sweep):
// 130432 rows 77.2 : 59.1, 65216 rows 145 : 110, 32608 rows 265 : 203, 16304 rows 441 : 364, 8152 rows 582 : 533,
// 4076 rows 643 : 612.  Smaller sizes were not measured: the default takes the pair form from 4000 rows.
bool psi2_pair_selected(rgp_handle h, int64_t N, int M, int QP, PairGeom& g) {
  int mode = h->psi2_pair;
  if (mode < 0) {
    static int env_mode = -2;
    if (env_mode == -2) { const char* e = getenv("RGP_PSI2_PAIR"); env_mode = e ? atoi(e) : PAIR_DEFAULT_MODE; }
    mode = env_mode;
  }
  if (mode <= 0 || h->deterministic) return false;
  if (!psi2_pair_geom(h, N, M, QP, g)) return false;
  if (mode >= 2) return true;
  return M >= 48 && N >= 4000;
}

size_t psi2_pair_ws_doubles(const PairGeom& g, int64_t N) {
  auto r = [](size_t n) { return (n + 31) & ~(size_t)31; };
  return r((size_t)g.ntiles * g.RS) + r((size_t)N * g.PW) + r((size_t)g.RR * g.Ppad * g.QP) + r((size_t)g.Ppad * g.QP) + 64;
}

template <int QP>
static hipError_t pair_launch(hipStream_t s, const PairGeom& g, int64_t N, int Q, const ZTerms& zt, const RowTerms& rt,
                              const double* dLs, const Psi2Out& o, double* ws) {
  auto r = [](size_t n) { return (n + 31) & ~(size_t)31; };
  double* rec = ws;
  double* PV = rec + r((size_t)g.ntiles * g.RS);
  double* Cpart = PV + r((size_t)N * g.PW);
  double* Csum = Cpart + r((size_t)g.RR * g.Ppad * g.QP);
  const int M = g.M;
  const int MT = (M + 15) / 16;
  const size_t lp = (2 * (size_t)16 * MT * (QP + 1) + 16 * MT) * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)psi2_pair_prep_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp);
  if (e != hipSuccess) return e;
  int64_t prep_grid = (g.ntiles + 3) / 4;                 // one round of workgroups: each pays for its LDS images of Z once
  if (prep_grid > (int64_t)g.NB) prep_grid = g.NB;
  hipLaunchKernelGGL((psi2_pair_prep_kernel<QP>), dim3((unsigned)prep_grid), dim3(256), lp, s, N, g.ntiles, M, g.HS, g.RS, zt.Zc, zt.lam,
                     rt.Xmu, rt.d2, rt.g0, rec, PV, g.PW);
  PairArgs a{};
  a.N = N; a.ntiles = g.ntiles; a.M = M; a.P = g.P; a.G = g.G; a.RR = g.RR; a.PW = g.PW; a.M2 = g.M2;
  a.tri = M * (M + 1) / 2; a.Ppad = g.Ppad;
  a.rec = rec; a.Zc = zt.Zc; a.lam = zt.lam; a.dLs = dLs; a.PV = PV; a.Cpart = Cpart; a.p2_part = o.p2_part;
  e = hipFuncSetAttribute((const void*)psi2_pair_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
  if (e != hipSuccess) return e;
#ifdef PAIR_STAMPS
  static double* dbg_s = nullptr;
  if (!dbg_s && getenv("RGP_STAMPS")) (void)hipMalloc(&dbg_s, 256 * 8 * 16 * sizeof(double));
  a.dbg = dbg_s;
#endif
  hipLaunchKernelGGL((psi2_pair_kernel<QP>), dim3(g.G * g.RR), dim3(512), g.lds, s, a);
#ifdef PAIR_STAMPS
  if (dbg_s && g.G * g.RR <= 256) {
    (void)hipStreamSynchronize(s);
    static double hb[256 * 8 * 16];
    (void)hipMemcpy(hb, dbg_s, sizeof(hb), hipMemcpyDeviceToHost);
    const int nwg = g.G * g.RR;
    fprintf(stderr, "pair QP=%d core clock during the step loop of workgroup 0: %.0f MHz\n", QP, 100.0 * hb[13] / (hb[14] > 0 ? hb[14] : 1));
    {                                             // spread of the workgroups' step loops: wall time (100 MHz ticks) and cycles, wave 0 of each
      double wmin = 1e30, wmax = 0, wsum = 0, cmin = 1e30, cmax = 0;
      for (int wg = 0; wg < nwg; ++wg) {
        const double w = hb[(wg * 8) * 16 + 14], c = hb[(wg * 8) * 16 + 13];
        wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax; wsum += w; cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
      }
      fprintf(stderr, "pair QP=%d step loop per workgroup: wall min %.1f avg %.1f max %.1f us; cycles min %.0f max %.0f\n", QP, wmin / 100, wsum / nwg / 100,
              wmax / 100, cmin, cmax);
      fprintf(stderr, "  wall us by workgroup:");
      for (int wg = 0; wg < nwg; ++wg) fprintf(stderr, " %.0f", hb[(wg * 8) * 16 + 14] / 100);
      fprintf(stderr, "\n");
      for (int x = 0; x < 8; ++x) {               // by XCD (workgroup i runs on XCD i mod 8)
        double ws = 0, cs = 0; int n = 0;
        for (int wg = x; wg < nwg; wg += 8) { ws += hb[(wg * 8) * 16 + 14]; cs += hb[(wg * 8) * 16 + 13]; ++n; }
        fprintf(stderr, "  XCD %d: wall %.1f us, %.0f MHz\n", x, ws / n / 100, 100.0 * cs / ws);
      }
    }
    for (int role = 0; role < 2; ++role) {
      double acc[16] = {0};
      for (int wg = 0; wg < nwg; ++wg) for (int w = 4 * role; w < 4 * role + 4; ++w) for (int i = 0; i < 13; ++i)
        acc[i] += hb[(wg * 8 + w) * 16 + i] / (hb[(wg * 8 + w) * 16 + 15] > 0 ? hb[(wg * 8 + w) * 16 + 15] : 1) / (4 * nwg);
      fprintf(stderr, "pair QP=%d role %d steps/wg %.0f, cycles from the step's start: head %.0f gram %.0f exp+v2 %.0f | C quarter %.0f T prologue %.0f C %.0f | T tiles %.0f %.0f %.0f %.0f flush %.0f | barrier %.0f after %.0f\n", QP,
              role, hb[15], acc[0], acc[1], acc[2], acc[11], acc[12], acc[3], acc[7], acc[8], acc[9], acc[10], acc[4], acc[5], acc[6]);
    }
  }
#endif
  // row gradients, dinv_l / weight-sum partials, B slabs
  const size_t lr0 = (32 * (size_t)(16 * MT + 1) + 32 * (size_t)(32 * ((QP + 15) / 16) + 1) + 32) * sizeof(double);
  int KL = (int)((80 * 1024 - lr0) / (4 * (QP + 1) * sizeof(double)));       // k-steps of job A whose rows of Z fit: two workgroups per CU
  if (KL > (M + 3) / 4) KL = (M + 3) / 4;
  if (KL < 0) KL = 0;
  const size_t lr = lr0 + (size_t)4 * KL * (QP + 1) * sizeof(double);
  e = hipFuncSetAttribute((const void*)psi2_pair_rows_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((psi2_pair_rows_kernel<QP>), dim3(g.NB), dim3(256), lr, s, N, M, Q, g.PW, g.M2, KL, PV, zt.Zc, zt.lam, rt.Xmu, rt.d2,
                     o.gmu, o.gS, o.accumulate, o.dz_part, o.dlam_part, o.wsum_part);
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

// ==================================================================================================================
// Blocked form for 112 < M <= 1024 (the blocked layer step, rgp_big_impl.h): the same main kernel over pair groups that
// touch at most PAIR_LB blocks of 16 inducing rows each.  Its outputs are V2 [N][Mld] and T [N][QP] (sums over pairs, added by the
// groups with global atomics), C [tile order][QP] and the pair sums of psi2 [tile order] (sums over rows, added by the row ranges);
// the caller finishes the row gradients and dZ from those with matrix products, as for wide windows.
// ==================================================================================================================
// slot 16 tau + c of the unordered pair (a, b) in the tile order of the main kernel's set-up
static __host__ __device__ inline int pairblk_slot(int a, int b, int nb, int rem) {
  int Ia = a >> 4, Ib = b >> 4, ca = a & 15, cb = b & 15;
  int I, J, c, d;
  if (Ia == Ib) {
    I = J = Ia;
    const int diff = (cb - ca) & 15;
    if (diff < 8) { d = diff; c = ca; }
    else if (diff > 8) { d = 16 - diff; c = cb; }
    else { d = 8; c = ca < cb ? ca : cb; }
  } else {
    if (Ia > Ib) { int x = Ia; Ia = Ib; Ib = x; x = ca; ca = cb; cb = x; }
    I = Ia; J = Ib; c = ca;
    d = (J == nb - 1) ? ((cb - ca) % rem + rem) % rem : (cb - ca) & 15;
  }
  const int S = 9 * I + 16 * (I * (nb - 2) - I * (I - 1) / 2) + rem * I;       // tiles of the block rows before I
  const int tau = S + (J == I ? d : 9 + 16 * (J - I - 1) + d);
  return 16 * tau + c;
}

// Records of the blocked form, one per tile of 16 data rows: rec [tile][ND2 x 128] = -d2 / 2 as [16][TS], zero padded;
// Hq [tile][column block][16 rows][16] = h[n,m] + g0[n] / 2 (h: the (row, m) part of the exponent as the blocked psi1 forward pass
// leaves it, [N][M]; nullptr if that pass has written Hq itself); rows beyond N get -inf (their exponentials are 0), columns
// beyond M get 0.  Also zeroes every array the main kernel adds into.
template <int QP>
__global__ void __launch_bounds__(256) psi2_pairblk_prep_kernel(const int64_t N, const int64_t ntiles, const int M, const int nb,
                                                                 const int Mld, const double* __restrict__ d2,
                                                                 const double* __restrict__ g0, const double* __restrict__ H,
                                                                 double* __restrict__ rec, double* __restrict__ Hq,
                                                                 double* __restrict__ V2, double* __restrict__ T,
                                                                 double* __restrict__ Csum, const int64_t ncs,
                                                                 double* __restrict__ p2sum, const int64_t np2) {
  constexpr int QT = (QP + 15) / 16, TS = 16 * QT + 1, ND2 = (16 * TS + 127) / 128;
  const int t = threadIdx.x;
  for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < ncs; i += (int64_t)gridDim.x * 256) Csum[i] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < np2; i += (int64_t)gridDim.x * 256) p2sum[i] = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * 16;
    double* const r = rec + (size_t)tile * (ND2 * 128);
    for (int i = t; i < ND2 * 128; i += 256) {
      const int row = i / TS, col = i - row * TS;
      const int64_t n = n0 + row;
      r[i] = (row < 16 && col < QP && n < N) ? -0.5 * d2[(size_t)n * QP + col] : 0.0;
    }
    double* const hq = Hq + (size_t)tile * nb * 256;
    if (H) for (int i = t; i < nb * 256; i += 256) {           // (H == nullptr: the psi1 forward pass has written the blocks itself)
      const int blk = i >> 8, row = (i >> 4) & 15, m = 16 * blk + (i & 15);
      const int64_t n = n0 + row;
      hq[i] = (n < N) ? ((m < M) ? H[(size_t)n * M + m] + 0.5 * g0[n] : 0.0) : -INFINITY;
    }
    const int64_t rows = (N - n0 < 16) ? N - n0 : 16;
    for (int64_t i = t; i < rows * Mld; i += 256) V2[n0 * Mld + i] = 0.0;
    for (int64_t i = t; i < rows * QP; i += 256) T[n0 * QP + i] = 0.0;
  }
}

// psi2[a,b] = psi2[b,a] = pair sum;  dZ3[m,q] = sum_m' z[m',q] sum_n d2[n,q] W[n,m,m'] from the C sums (taken over -d2 / 2 with
// c_p = 2 off the diagonal in the weights)
__global__ void __launch_bounds__(256) psi2_pairblk_sym_kernel(const int M, const int nb, const int rem, const double* __restrict__ p2sum,
                                                                double* __restrict__ psi2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * M) return;
  const int a = i / M, b = i - a * M;
  psi2[i] = p2sum[pairblk_slot(a, b, nb, rem)];
}
__global__ void __launch_bounds__(256) psi2_pairblk_dz3_kernel(const int M, const int QP, const int nb, const int rem, const double* __restrict__ Csum,
                                                                const double* __restrict__ Zc, double* __restrict__ dz3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * QP) return;
  const int m = i / QP, q = i - m * QP;         // consecutive threads walk q: coalesced reads of C
  double s = 0.0;
#pragma unroll 4
  for (int mp = 0; mp < M; ++mp) {
    const double cv = Csum[(size_t)pairblk_slot(m, mp, nb, rem) * QP + q];
    s += Zc[(size_t)mp * QP + q] * (mp == m ? cv : 0.5 * cv);
  }
  dz3[i] = -2.0 * s;
}

bool psi2_pairblk_geom(rgp_handle h, int64_t N, int M, int QP, int Mld, PairBlkGeom& g) {
  if (M <= 112 || M > 1024 || QP < 8 || QP > RGP_MAX_QP || (QP & 3) || N < 1) return false;
  g.M = M; g.QP = QP; g.Mld = Mld;
  g.nb = (M + 15) / 16;
  if (Mld < 16 * g.nb) return false;
  g.NT = pair_nt(QP);
  g.TG = 8 * g.NT;
  g.rem = M - 16 * (g.nb - 1);
  g.ntl = 9 * g.nb + 8 * (g.nb - 1) * (g.nb - 2) + g.rem * (g.nb - 1);
  g.G = (g.ntl + g.TG - 1) / g.TG;
  {                                               // the block list of every group must fit: walk the tile order as the kernel does
    int I = 0, J = 0, d = 0;
    for (int gi = 0; gi < g.G; ++gi) {
      int list[2 * PAIR_GROUP_MAX / 16], nl = 0;
      for (int i = 0; i < g.TG && I < g.nb; ++i) {
        bool fa = false, fb = false;
        for (int j = 0; j < nl; ++j) { fa |= list[j] == I; fb |= list[j] == J; }
        if (!fa) list[nl++] = I;
        if (!fb && J != I) list[nl++] = J;
        if (++d == (I == J ? 9 : (J == g.nb - 1 ? g.rem : 16))) { d = 0; if (++J == g.nb) { ++I; J = I; } }
      }
      if (nl > PAIR_LB) return false;
    }
  }
  g.ntiles = (N + 15) / 16;
  // row ranges: a task should walk a few hundred steps (its set-up and its flush of 512 x QP sums are then negligible), and the
  // tasks should fill the last round of workgroups
  const int cu = h->num_cu > 0 ? h->num_cu : 256;
  int64_t rmax = g.ntiles / 256;
  if (rmax > 64) rmax = 64;
  {                                               // few pair groups (small M): shorter tasks, down to 24 steps, until every CU has two
    int64_t rfill = 2 * ((cu + g.G - 1) / g.G), rcap = g.ntiles / 24;
    if (rfill > rcap) rfill = rcap;
    if (rmax < rfill) rmax = rfill;
  }
  if (rmax < 1) rmax = 1;
  int best = (int)rmax;
  double bfill = 0.0;
  for (int64_t r = rmax; r >= (rmax + 1) / 2; --r) {
    const int64_t tasks = (int64_t)g.G * r, rounds = (tasks + cu - 1) / cu;
    const double fill = (double)tasks / (double)(rounds * cu);
    if (fill > bfill + 1e-9) { bfill = fill; best = (int)r; }
  }
  g.RR = best;
  g.ntasks = g.G * g.RR;
  g.grid = g.ntasks < cu ? g.ntasks : cu;
  const int QT = (QP + 15) / 16, TS = 16 * QT + 1, ZS = TS, TT = ((16 * QT) % 32 == 16) ? 16 * QT : 16 * QT + 16;
  g.ND2 = (16 * TS + 127) / 128;
  const size_t dbl = PSI2_EXP_TAB + 2 * (size_t)(g.ND2 * 128 + PAIR_LB * 256) + 2 * (size_t)PAIR_LB * 256 + 3 * 16 * (size_t)TT +
                     PAIR_WAVES * 16 * 17 + 3 * 128 * (size_t)g.NT + (size_t)PAIR_LB * 16 * ZS + (8 * g.NT + PAIR_LB + 1) / 2;
  g.lds = dbl * sizeof(double);
  return g.lds <= 160 * 1024;
}

// Measured against the blocked forward + backward sweep kernels through rgp_psi_backward (tools/pairblk_ab.py; ms per call,
// sweeps : pair form): M = 500, Q = 40: 1024 rows 2.2 : 1.9, 4096 rows 6.4 : 5.2, 16384 rows 23.5 : 15.4, 65536 rows 93.5 : 50.4;
// M = 200: 1024 rows 0.58 : 0.80, 2048 rows 0.95 : 0.76, 65536 rows 20.4 : 9.6; M = 1000: 1.4x at 512 rows, 2.1x at 16384.
// The blocked form is chosen from 2048 rows.
bool psi2_pairblk_selected(rgp_handle h, int64_t N, int M, int QP, int Mld, PairBlkGeom& g) {
  int mode = h->psi2_pair;
  if (mode < 0) {
    static int env_mode = -2;
    if (env_mode == -2) { const char* e = getenv("RGP_PSI2_PAIR"); env_mode = e ? atoi(e) : PAIR_DEFAULT_MODE; }
    mode = env_mode;
  }
  if (mode <= 0 || h->deterministic) return false;
  if (!psi2_pairblk_geom(h, N, M, QP, Mld, g)) return false;
  if (mode >= 2) return true;
  return N >= 2048;
}

static inline size_t pb_r(size_t n) { return (n + 31) & ~(size_t)31; }
double* psi2_pairblk_hq(const PairBlkGeom& g, double* ws) { return ws + pb_r((size_t)g.ntiles * g.ND2 * 128); }
size_t psi2_pairblk_ws_doubles(const PairBlkGeom& g, int64_t N) {
  return pb_r((size_t)g.ntiles * g.ND2 * 128) + pb_r((size_t)g.ntiles * g.nb * 256) + pb_r((size_t)N * g.Mld) + pb_r((size_t)N * g.QP) +
         pb_r((size_t)g.G * g.TG * 16 * g.QP) + pb_r((size_t)g.G * g.TG * 16) + 64;
}

template <int QP>
static hipError_t pairblk_launch(hipStream_t s, const PairBlkGeom& g, int64_t N, const ZTerms& zt, const RowTerms& rt, const double* H,
                                 const double* dLs, double* ws, PairBlkOut& o, int num_cu) {
  double* rec = ws;
  double* Hq = rec + pb_r((size_t)g.ntiles * g.ND2 * 128);
  double* V2 = Hq + pb_r((size_t)g.ntiles * g.nb * 256);
  double* T = V2 + pb_r((size_t)N * g.Mld);
  double* Csum = T + pb_r((size_t)N * g.QP);
  double* p2sum = Csum + pb_r((size_t)g.G * g.TG * 16 * g.QP);
  const int64_t ncs = (int64_t)g.G * g.TG * 16 * QP, np2 = (int64_t)g.G * g.TG * 16;
  int64_t pg = g.ntiles < 4 * (int64_t)num_cu ? g.ntiles : 4 * (int64_t)num_cu;
  if (pg < 1) pg = 1;
  hipLaunchKernelGGL((psi2_pairblk_prep_kernel<QP>), dim3((unsigned)pg), dim3(256), 0, s, N, g.ntiles, g.M, g.nb, g.Mld, rt.d2, rt.g0, H, rec,
                     Hq, V2, T, Csum, ncs, p2sum, np2);
  PairArgs a{};
  a.N = N; a.ntiles = g.ntiles; a.M = g.M; a.G = g.G; a.RR = g.RR;
  a.rec = rec; a.Zc = zt.Zc; a.lam = zt.lam; a.dLs = dLs;
  a.Hq = Hq; a.V2 = V2; a.Tg = T; a.Csum = Csum; a.p2sum = p2sum; a.nb = g.nb; a.Mld = g.Mld; a.ntl = g.ntl; a.ntasks = g.ntasks; a.rem = g.rem;
  hipError_t e = hipFuncSetAttribute((const void*)psi2_pairblk_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((psi2_pairblk_kernel<QP>), dim3(g.grid), dim3(512), g.lds, s, a);
  hipLaunchKernelGGL(psi2_pairblk_sym_kernel, dim3((g.M * g.M + 255) / 256), dim3(256), 0, s, g.M, g.nb, g.rem, p2sum, o.psi2);
  hipLaunchKernelGGL(psi2_pairblk_dz3_kernel, dim3((g.M * QP + 255) / 256), dim3(256), 0, s, g.M, QP, g.nb, g.rem, Csum, zt.Zc, o.dz3);
  o.V2 = V2; o.T = T;
  return hipGetLastError();
}

rgp_status launch_psi2_pairblk(rgp_handle h, hipStream_t s, const PairBlkGeom& g, int64_t N, const ZTerms& zt, const RowTerms& rt,
                               const double* H, const double* dLs, double* ws, PairBlkOut& out) {
  hipError_t e = hipSuccess;
  switch (g.QP) {
#define PBL(qp) case qp: e = pairblk_launch<qp>(s, g, N, zt, rt, H, dLs, ws, out, h->num_cu); break;
    PBL(8) PBL(16) PBL(20) PBL(24) PBL(32) PBL(40) PBL(48) PBL(56) PBL(64)
#undef PBL
    default: return rgp_fail(h, RGP_ERR_BAD_ARG, "psi2 (pair form, blocked): unsupported padded Q");
  }
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string("psi2 (pair form, blocked) launch: ") + hipGetErrorString(e));
  return RGP_OK;
}
