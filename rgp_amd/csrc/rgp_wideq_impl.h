// rgp_wideq_impl.h -- the psi statistics and their gradients for windows wider than 64 columns (64 < padded Q <= 1024, any
// M <= 1024).  Included by rgp_big_impl.h; entered from the blocked layer steps of both bounds (big_svi_layer_step,
// big_vardtc_layer_step) and from the unit entry points (big_psi_forward / big_psi_backward).
//
// With Q in the hundreds the per-thread arrays of the sweep kernels (dzl[Q], y[Q]) cannot stay in registers, and the two
// O(N M^2 Q) contractions of the psi2 term become what they are algebraically: batched matrix products with the data rows as
// the batch.  Written over the P = M (M + 1) / 2 pairs of inducing rows (the weights are symmetric) they are three plain, large
// matrix products on the MFMA GEMM (rgp_gemm.hip), with the N x P weights in HBM:
//   gram   E[n, p] = sum_q dm[n,q] ZZ2[q,p],   ZZ2[q,p] = z[a,q] z[b,q] for the pair p = (a <= b)            (N x P x Q product)
//   exp    W = dLs o exp(E + g0[n] + h[n,a] + h[n,b]);  psi2 = sum_n exp(.);  v[n,m] = sum_m' W           (wq_exp_kernel, c o W in place)
//   back   T[n,q] = sum_p c_p W[n,p] ZZ2[q,p]   (N x Q x P product);   C[p,q] = sum_n c_p W[n,p] d2[n,q]   (P x Q x N product),
//          dZ3[m,q] = sum_m' z[m',q] C[(m,m'),q] / c
// and, once per layer step, from v alone:  [a1 | vz2 | w] = V [Z | Z^2 | 1],  B1 = V^T (d2 o mu),  B2 = V^T d2, so that
//   a23 = (vz2 + T) / 2,  row gradients as in the sweep kernels' epilogue,  dZ2 = 2 B1 - Z o B2 - dZ3.
// (psi2 exponent and its gradients: GPy's rbf_psi_comp closed forms as restated in rgp_psi.hip; bound algebra:
// autoreg/inference/svi_vardtc.py:70-195, vardtc.py:88-208.)  psi1 and h are four N x M x Q products (wq_psi1_fwd); everything
// else of a step is the blocked path's (GEMM-shaped psi1 backward side, chain by M, finalisation, window scatter).  The N x P
// weights are held for X.nc data rows at a time; the collapsed bound, whose dL/dpsi2 needs psi2 first, runs the Gram product
// and a forward-only exponential pass before the chain (wq_psi2_forward) and everything again after it (wq_psi2).
#pragma once


// device scalars of the blocked step's fused GEMM epilogues: [3] beta, [4] beta/2, [5] -beta, [6] 2/(beta D)
__global__ void wq_scal_kernel(const double* __restrict__ noise, int D, double* __restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double beta = 1.0 / noise[0];
    scal[3] = beta; scal[4] = 0.5 * beta; scal[5] = -beta; scal[6] = 2.0 / (beta * (double)D);
  }
}

// DM[n,q] = d2[n,q] mu[n,q]
__global__ void __launch_bounds__(256) wq_dmu_kernel(int64_t n, const double* __restrict__ d2, const double* __restrict__ mu,
                                                      double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d2[i] * mu[i];
}

// psi1 and the per-row part h of the psi2 exponent, for any Q, as four N x M x Q products on the right operand ZZ = [Z | Z^2 | 1]:
//   P[n,m] = exp(g1[n] - 1/2 (e (Z^2)^T - 2 (e mu) Z^T));     H[n,m] = (d2 mu) Z^T - 1/4 d2 (Z^2)^T - 1/4 sum_q lam z[m,q]^2
__global__ void __launch_bounds__(128) wq_ellq_kernel(const int M, const int QP, const double* __restrict__ Zc,
                                                       const double* __restrict__ lam, double* __restrict__ ellq) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int q = 0; q < QP; ++q) { const double z = Zc[(size_t)m * QP + q]; s += 0.25 * lam[q] * z * z; }
  ellq[m] = s;
}

__global__ void __launch_bounds__(256) wq_pfin_kernel(const int64_t NM, const int M, const double* __restrict__ g1,
                                                       const double* __restrict__ ellq, double* __restrict__ P, double* __restrict__ H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NM) return;
  const int64_t n = i / M;
  const int m = (int)(i - n * M);
  P[i] = exp(g1[n] - 0.5 * P[i]);
  H[i] -= ellq[m];
}

static void wq_psi1_fwd(hipStream_t s, const BigPlan& p, int64_t N, int M, BigWs& L, WqWs& X) {
  const int QP = p.QP;
  hipLaunchKernelGGL(wq_ellq_kernel, dim3((M + 127) / 128), dim3(128), 0, s, M, QP, L.zt.Zc, L.zt.lam, X.ellq);
  hipLaunchKernelGGL(wq_dmu_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, L.rt.d2, L.rt.Xmu, X.DM);
  const double* Z1 = L.ZZ;            // z      (M x QP inside rows of width W)
  const double* Z2 = L.ZZ + QP;       // z^2
  GemmArgs g;
  g.m = (int)N; g.n = M; g.k = QP; g.lda = QP; g.ldb = p.W; g.tb = 1; g.ldc = M;
  g.A = L.rt.e1; g.B = Z2; g.C = L.P;
  launch_gemm(s, g);
  g.A = L.rt.em; g.B = Z1; g.alpha = -2.0; g.gamma = 1.0; g.Cin = L.P; g.ldcin = M;
  launch_gemm(s, g);
  g.A = X.DM; g.B = Z1; g.C = L.H; g.alpha = 1.0; g.gamma = 0.0; g.Cin = nullptr;
  launch_gemm(s, g);
  g.A = L.rt.d2; g.B = Z2; g.alpha = -0.25; g.gamma = 1.0; g.Cin = L.H; g.ldcin = M;
  launch_gemm(s, g);
  hipLaunchKernelGGL(wq_pfin_kernel, dim3(big_nb256(N * M)), dim3(256), 0, s, N * M, M, L.rt.g1, X.ellq, L.P, L.H);
}

// pair tables: p = index of (a, b), a <= b, rows of the upper triangle one after the other
__global__ void __launch_bounds__(256) wq_pairs_kernel(const int M, int* __restrict__ pa, int* __restrict__ pb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * M) return;
  const int a = i / M, b = i % M;
  if (b < a) return;
  const int p = a * M - a * (a - 1) / 2 + (b - a);
  pa[p] = a; pb[p] = b;
}

// ZZ2[q, p] = z[a,q] z[b,q]       (QP x P, the right operand of the Gram product and of the product that gives T)
__global__ void __launch_bounds__(256) wq_zz2_kernel(const int P, const int QP, const int* __restrict__ pa, const int* __restrict__ pb,
                                                      const double* __restrict__ Zc, double* __restrict__ ZZ2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)P * QP) return;
  const int q = (int)(i / P), p = (int)(i - (int64_t)q * P);
  ZZ2[i] = Zc[(size_t)pa[p] * QP + q] * Zc[(size_t)pb[p] * QP + q];
}

// The exponential pass over the rows [row0, row0 + nrows) whose weights E[n - row0, p] (nrows x P) are in memory.  Workgroup = rb
// consecutive data rows, whose h rows and v accumulators are in LDS; thread = one pair at a time, walking the block's rows:
//   e = exp(E + g0[n] + h[n,a] + h[n,b]);  psi2 partial [wg][p] = sum_n e;  w = e dLs[a,b];  E <- c_p w  (c_p = 2 off the
//   diagonal: one stored pair stands for two entries of the symmetric weights);  v[n,a] += w, v[n,b] += w (a != b)
__global__ void __launch_bounds__(256) wq_exp_kernel(const int64_t row0, const int64_t nrows, const int rb, const int M, const int P,
                                                      const int* __restrict__ pa, const int* __restrict__ pb, double* __restrict__ E,
                                                      const double* __restrict__ g0, const double* __restrict__ H,
                                                      const double* __restrict__ dLs, double* __restrict__ p2_part,
                                                      double* __restrict__ vrow) {
  extern __shared__ double wq_sm[];
  double* hl = wq_sm;                       // [rb][M]
  double* vl = wq_sm + rb * M;              // [rb][M]
  double* gl = vl + rb * M;                 // [rb]
  const int64_t l0 = (int64_t)blockIdx.x * rb;                  // first row of the block within the chunk
  const int nr = (int)((nrows - l0 < rb) ? nrows - l0 : rb);
  const int64_t r0 = row0 + l0;
  const bool fwd_only = dLs == nullptr;                         // forward sums alone (collapsed bound: before the chain)
  for (int i = threadIdx.x; i < nr * M; i += 256) { hl[i] = H[r0 * M + i]; vl[i] = 0.0; }
  if (threadIdx.x < nr) gl[threadIdx.x] = g0[r0 + threadIdx.x];
  __syncthreads();
  double* __restrict__ pp = p2_part + (size_t)blockIdx.x * P;
  for (int p = threadIdx.x; p < P; p += 256) {
    const int a = pa[p], b = pb[p];
    double* __restrict__ e = E + (size_t)l0 * P + p;
    double acc = 0.0;
    int n = 0;
    if (fwd_only) {
      for (; n + 4 <= nr; n += 4) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = e[(size_t)(n + u) * P] + gl[n + u] + hl[(n + u) * M + a] + hl[(n + u) * M + b];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += exp(x[u]);
      }
      for (; n < nr; ++n) acc += exp(e[(size_t)n * P] + gl[n] + hl[n * M + a] + hl[n * M + b]);
      pp[p] = acc;
      continue;
    }
    const double dl = 0.5 * (dLs[(size_t)a * M + b] + dLs[(size_t)b * M + a]);
    const double cp = (a == b) ? 1.0 : 2.0;
    for (; n + 4 <= nr; n += 4) {
      double x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = e[(size_t)(n + u) * P] + gl[n + u] + hl[(n + u) * M + a] + hl[(n + u) * M + b];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = exp(x[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += x[u];
        const double w = x[u] * dl;
        e[(size_t)(n + u) * P] = cp * w;
        atomicAdd(&vl[(n + u) * M + a], w);
        if (a != b) atomicAdd(&vl[(n + u) * M + b], w);
      }
    }
    for (; n < nr; ++n) {
      const double x = exp(e[(size_t)n * P] + gl[n] + hl[n * M + a] + hl[n * M + b]);
      acc += x;
      const double w = x * dl;
      e[(size_t)n * P] = cp * w;
      atomicAdd(&vl[n * M + a], w);
      if (a != b) atomicAdd(&vl[n * M + b], w);
    }
    pp[p] = acc;
  }
  __syncthreads();
  if (!fwd_only)
    for (int i = threadIdx.x; i < nr * M; i += 256) vrow[r0 * M + i] = vl[i];
}

// psi2[a,b] = psi2[b,a] = pair sums
__global__ void __launch_bounds__(256) wq_sym_kernel(const int M, const int P, const int* __restrict__ pa, const int* __restrict__ pb,
                                                      const double* __restrict__ pair, double* __restrict__ psi2) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int a = pa[p], b = pb[p];
  const double v = pair[p];
  psi2[(size_t)a * M + b] = v;
  psi2[(size_t)b * M + a] = v;
}

// dZ3[m,q] = sum_m' z[m',q] sum_n d2[n,q] W[n,m,m'] from the split partials C[s][p][q] of (c o W)^T d2   (thread = (m, q))
__global__ void __launch_bounds__(256) wq_dz3_kernel(const int M, const int P, const int QP, const int nsplit,
                                                      const double* __restrict__ C, const double* __restrict__ Zc,
                                                      double* __restrict__ dz3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * QP) return;
  const int m = i / QP, q = i % QP;
  double s = 0.0;
  for (int mp = 0; mp < M; ++mp) {
    const int a = mp < m ? mp : m, b = mp < m ? m : mp;
    const size_t p = (size_t)a * M - (size_t)(a * (a - 1) / 2) + (b - a);
    double c = 0.0;
    for (int k = 0; k < nsplit; ++k) c += C[((size_t)k * P + p) * QP + q];
    s += Zc[(size_t)mp * QP + q] * (mp == m ? c : 0.5 * c);
  }
  dz3[i] = s;
}

// Row gradients of the psi2 term from S22[n] = [a1 | vz2 | w | ..] = v[n,:] [Z | Z^2 | 1] and T (same expressions as the sweep
// kernels' epilogue):  a23 = (vz2 + T) / 2;  dmu = 2 d (a1 - mu w);  dd = -mu^2 w + 2 mu a1 - a23 + w / (2 d);  dS = -2 d^2 dd;
// dinv_l += dd (d/lam)^2 - w / (2 lam)        (grid: row blocks x 64-column chunks; gradients are ADDED to gmu / gS)
__global__ void __launch_bounds__(256) wq_rowgrad_kernel(int64_t N, int Q, int QP, int W, const double* __restrict__ S22,
                                                          const double* __restrict__ T, const double* __restrict__ Xmu,
                                                          const double* __restrict__ d2, const double* __restrict__ lam,
                                                          double* __restrict__ gmu, double* __restrict__ gS,
                                                          double* __restrict__ dlam_part, double* __restrict__ wsum_part) {
  __shared__ double sm[4][64];
  __shared__ double sl[4];
  const int ql = threadIdx.x & 63, q = (int)blockIdx.y * 64 + ql, g = threadIdx.x >> 6;
  double dl = 0.0, wsum = 0.0;
  const double il = (q < Q) ? 1.0 / lam[q] : 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 4 + g; n < N; n += (int64_t)gridDim.x * 4) {
    if (q < Q) {
      const double w = S22[n * W + 2 * QP], a1 = S22[n * W + q], a23 = 0.5 * (S22[n * W + QP + q] + T[n * QP + q]);
      const double mu = Xmu[n * QP + q], dq = d2[n * QP + q];
      const double dmu = 2.0 * dq * (a1 - mu * w);
      const double dd = -mu * mu * w + 2.0 * mu * a1 - a23 + 0.5 * w / dq;
      gmu[n * QP + q] += dmu;
      gS[n * QP + q] += -2.0 * dq * dq * dd;
      dl += dd * (dq * il) * (dq * il) - 0.5 * w * il;
      if (q == 0) wsum += w;
    }
  }
  sm[g][ql] = dl;
  if (q == 0) sl[g] = wsum;
  __syncthreads();
  if (g == 0) {
    if (q < QP) dlam_part[(size_t)blockIdx.x * QP + q] = (sm[0][ql] + sm[1][ql]) + (sm[2][ql] + sm[3][ql]);
    if (q == 0) wsum_part[blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
  }
}

// dz2[q][m] (row stride Mld) = 2 B1[m][q] - z[m][q] B2[m][q] - dZ3[m][q]
__global__ void __launch_bounds__(256) wq_dz2_kernel(int M, int QP, int Mld, const double* __restrict__ B1, const double* __restrict__ B2,
                                                      const double* __restrict__ dZ3, const double* __restrict__ Zc,
                                                      double* __restrict__ dz2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * QP) return;
  const int m = i / QP, q = i % QP;
  dz2[(size_t)q * Mld + m] = 2.0 * B1[i] - Zc[i] * B2[i] - dZ3[i];
}

// psi2 forward sum alone (collapsed bound: dL/dpsi2 needs psi2 before the backward products can run)
static rgp_status wq_psi2_forward(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, BigWs& L, WqWs& X) {
  const int QP = p.QP, P = X.P;
  hipLaunchKernelGGL(wq_pairs_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, X.pa, X.pb);
  hipLaunchKernelGGL(wq_zz2_kernel, dim3(big_nb256((int64_t)P * QP)), dim3(256), 0, s, P, QP, X.pa, X.pb, L.zt.Zc, X.ZZ2);
  int chunk = 0;
  for (int64_t r0 = 0; r0 < N; r0 += X.nc, ++chunk) {
    const int64_t nc = (N - r0 < X.nc) ? N - r0 : X.nc;
    GemmArgs g;
    g.m = (int)nc; g.n = P; g.k = QP; g.A = L.dm + r0 * QP; g.lda = QP; g.B = X.ZZ2; g.ldb = P; g.C = X.E; g.ldc = P;
    launch_gemm(s, g);
    const int nexp = (int)((nc + X.rb - 1) / X.rb);
    hipLaunchKernelGGL(wq_exp_kernel, dim3(nexp), dim3(256), (2 * (size_t)X.rb * M + X.rb) * sizeof(double), s, r0, nc, X.rb, M, P,
                       X.pa, X.pb, X.E, L.rt.g0, L.H, (const double*)nullptr, X.p2w_part, X.vrow);
    launch_reduce(s, nexp, P, P, X.p2w_part, X.p2pair, 1.0, chunk > 0);
  }
  hipLaunchKernelGGL(wq_sym_kernel, dim3((P + 255) / 256), dim3(256), 0, s, M, P, X.pa, X.pb, X.p2pair, L.psi2);
  return last_launch(h, "psi2 forward (wide windows)");
}

// psi2 forward sum, row gradients (added to L.gmu / L.gS), dz2, dlam2, wsum2 for a wide window
static rgp_status wq_psi2(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, int Q, BigWs& L, WqWs& X) {
  const int QP = p.QP, P = X.P;
  hipLaunchKernelGGL(wq_pairs_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, X.pa, X.pb);
  hipLaunchKernelGGL(wq_zz2_kernel, dim3(big_nb256((int64_t)P * QP)), dim3(256), 0, s, P, QP, X.pa, X.pb, L.zt.Zc, X.ZZ2);
  // the N x P weights exist for X.nc data rows at a time; psi2 and C accumulate over the chunks, v and T are per row
  int chunk = 0;
  for (int64_t r0 = 0; r0 < N; r0 += X.nc, ++chunk) {
    const int64_t nc = (N - r0 < X.nc) ? N - r0 : X.nc;
    {                                                        // Gram: E = dm ZZ2   (nc x P, k = QP)
      GemmArgs g;
      g.m = (int)nc; g.n = P; g.k = QP; g.A = L.dm + r0 * QP; g.lda = QP; g.B = X.ZZ2; g.ldb = P; g.C = X.E; g.ldc = P;
      launch_gemm(s, g);
    }
    // exponential pass: c o W in place of E, forward partial sums, v
    const int nexp = (int)((nc + X.rb - 1) / X.rb);
    hipLaunchKernelGGL(wq_exp_kernel, dim3(nexp), dim3(256), (2 * (size_t)X.rb * M + X.rb) * sizeof(double), s, r0, nc, X.rb, M, P,
                       X.pa, X.pb, X.E, L.rt.g0, L.H, L.dLs, X.p2w_part, X.vrow);
    launch_reduce(s, nexp, P, P, X.p2w_part, X.p2pair, 1.0, chunk > 0);
    {                                                        // T = (c o W) ZZ2^T   (nc x QP, k = P)
      GemmArgs g;
      g.m = (int)nc; g.n = QP; g.k = P; g.A = X.E; g.lda = P; g.B = X.ZZ2; g.ldb = P; g.tb = 1; g.C = X.T + r0 * QP; g.ldc = QP;
      launch_gemm(s, g);
    }
    {                                                        // C (+)= (c o W)^T d2   (P x QP, k = nc, split over the rows)
      GemmArgs g;
      g.m = P; g.n = QP; g.k = (int)nc; g.A = X.E; g.lda = P; g.ta = 1; g.B = L.rt.d2 + r0 * QP; g.ldb = QP;
      g.C = X.C_part; g.ldc = QP; g.splits = X.nsplit; g.split_stride = (int64_t)P * QP;
      if (X.nsplit == 1) g.C = X.C;
      if (X.nsplit == 1 && chunk > 0) { g.gamma = 1.0; g.Cin = X.C; g.ldcin = QP; }
      launch_gemm(s, g);
      if (X.nsplit > 1) launch_reduce(s, X.nsplit, (int64_t)P * QP, (int64_t)P * QP, X.C_part, X.C, 1.0, chunk > 0);
    }
  }
  hipLaunchKernelGGL(wq_sym_kernel, dim3((P + 255) / 256), dim3(256), 0, s, M, P, X.pa, X.pb, X.p2pair, L.psi2);
  hipLaunchKernelGGL(wq_dz3_kernel, dim3((M * QP + 255) / 256), dim3(256), 0, s, M, P, QP, 1, X.C, L.zt.Zc, X.dz3);
  // from v alone: [a1 | vz2 | w] = V [Z | Z^2 | 1],  B1 = V^T (d2 o mu),  B2 = V^T d2
  {
    GemmArgs g;
    g.m = (int)N; g.n = p.W; g.k = M; g.A = X.vrow; g.lda = M; g.B = L.ZZ; g.ldb = p.W; g.C = X.S22; g.ldc = p.W;
    launch_gemm(s, g);
  }
  for (int k = 0; k < 2; ++k) {                              // X.DM = d2 o mu is still there from wq_psi1_fwd
    GemmArgs g;
    g.m = M; g.n = QP; g.k = (int)N; g.A = X.vrow; g.lda = M; g.ta = 1; g.B = k == 0 ? X.DM : L.rt.d2; g.ldb = QP;
    g.C = X.B_part; g.ldc = QP; g.splits = p.sk; g.split_stride = (int64_t)M * QP;
    if (p.sk == 1) g.C = k == 0 ? X.B1 : X.B2;
    launch_gemm(s, g);
    if (p.sk > 1) launch_reduce(s, p.sk, (int64_t)M * QP, (int64_t)M * QP, X.B_part, k == 0 ? X.B1 : X.B2, 1.0, 0);
  }
  hipLaunchKernelGGL(wq_dz2_kernel, dim3(big_nb256((int64_t)M * QP)), dim3(256), 0, s, M, QP, p.Mld, X.B1, X.B2, X.dz3, L.zt.Zc, L.dz2);
  hipLaunchKernelGGL(wq_rowgrad_kernel, dim3(p.rows_blk, (QP + 63) / 64), dim3(256), 0, s, N, Q, QP, p.W, X.S22, X.T, L.rt.Xmu,
                     L.rt.d2, L.zt.lam, L.gmu, L.gS, L.dlam2_part, L.wsum2_part);
  const ReduceSeg segs[2] = {{L.dlam2_part, L.dlam2, p.rows_blk, QP, QP}, {L.wsum2_part, L.wsum2, p.rows_blk, 1, 1}};
  launch_reduce_batch(s, 2, segs);
  return RGP_OK;
}
