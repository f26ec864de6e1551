// rgp_wideq_impl.h -- SVI layer step for windows wider than 64 columns (64 < padded Q <= 1024, M <= 128).  Included by
// rgp_layer.hip after rgp_big_impl.h; entered from big_svi_layer_step.
//
// With Q in the hundreds the per-thread arrays of the sweep kernels (dzl[Q], y[Q]) cannot stay in registers, and the two
// O(N M^2 Q) contractions of the psi2 term become what they are algebraically: batched matrix products with the data rows as
// the batch, run on the MFMA GEMM (rgp_gemm.hip) in chunks of WQ_QC window columns, with the N x M x M weights in HBM:
//   gram   E[(n,m), m'] (+)= A_c Z_c^T,      A_c[(n,m), q] = dm[n,q] z[m,q]           (A_c written by wq_a_kernel)
//   exp    W = dLs o exp(E + g0[n] + H[n,m] + H[n,m']);  psi2 += sum_n exp(.);  v[n,m] = sum_m' W      (wq_exp_kernel, W in place)
//   back   Y_c[(n,m), q] = W Z_c;   T[n,q] = sum_m z[m,q] Y_c;   dZ3[m,q] = sum_n d2[n,q] Y_c          (wq_t_kernel, wq_dz3_kernel)
// and, once per layer step, from v alone:  [a1 | vz2 | w] = V [Z | Z^2 | 1],  B1 = V^T (d2 o mu),  B2 = V^T d2, so that
//   a23 = (vz2 + T) / 2,  row gradients as in the sweep kernels' epilogue,  dZ2 = 2 B1 - Z o B2 - dZ3.
// (psi2 exponent and its gradients: GPy's rbf_psi_comp closed forms as restated in rgp_psi.hip; bound algebra:
// autoreg/inference/svi_vardtc.py:70-195.)  psi1 and h come from a generic-Q kernel; everything else of the step is the
// blocked path's (GEMM-shaped psi1 side, chain by M, finalisation, window scatter).
#pragma once


// device scalars of the blocked step's fused GEMM epilogues: [3] beta, [4] beta/2, [5] -beta, [6] 2/(beta D)
__global__ void wq_scal_kernel(const double* __restrict__ noise, int D, double* __restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double beta = 1.0 / noise[0];
    scal[3] = beta; scal[4] = 0.5 * beta; scal[5] = -beta; scal[6] = 2.0 / (beta * (double)D);
  }
}

// P[n,m] = exp(g1[n] - 1/2 sum_q z (e z - 2 e mu));   H[n,m] = sum_q d2 z (mu - z/4) - 1/4 sum_q lam z^2      (any Q, M <= 128)
// thread = inducing row m (z read from the transposed copy: coalesced), the row's operands are wave-uniform loads
__global__ void __launch_bounds__(128) wq_psi1_fwd_kernel(const int64_t N, const int M, const int QP, const double* __restrict__ ZcT,
                                                           const double* __restrict__ lam, const double* __restrict__ e1,
                                                           const double* __restrict__ em, const double* __restrict__ d2,
                                                           const double* __restrict__ Xmu, const double* __restrict__ g1,
                                                           double* __restrict__ P, double* __restrict__ H) {
  const int m = threadIdx.x;
  const bool act = m < M;
  double ellq = 0.0;
  for (int q = 0; q < QP; ++q) { const double z = ZcT[(size_t)q * RGP_PSI2_TEAM + m]; ellq += 0.25 * lam[q] * z * z; }
  for (int64_t r = blockIdx.x; r < N; r += gridDim.x) {
    const double* __restrict__ e = e1 + r * QP;
    const double* __restrict__ mm = em + r * QP;
    const double* __restrict__ d = d2 + r * QP;
    const double* __restrict__ mu = Xmu + r * QP;
    double a0 = 0.0, a1 = 0.0, h0 = 0.0, h1 = 0.0;
    for (int q = 0; q < QP; q += 2) {
      const double z0 = ZcT[(size_t)q * RGP_PSI2_TEAM + m], z1 = ZcT[(size_t)(q + 1) * RGP_PSI2_TEAM + m];
      a0 += z0 * (e[q] * z0 - 2.0 * mm[q]);
      a1 += z1 * (e[q + 1] * z1 - 2.0 * mm[q + 1]);
      h0 += z0 * d[q] * (mu[q] - 0.25 * z0);
      h1 += z1 * d[q + 1] * (mu[q + 1] - 0.25 * z1);
    }
    if (act) {
      P[r * M + m] = exp(g1[r] - 0.5 * (a0 + a1));
      H[r * M + m] = (h0 + h1) - ellq;
    }
  }
}

// A[(n,m), j] = dm[n, q0 + j] z[m, q0 + j]   (j < qc; zero beyond the padded width)
__global__ void __launch_bounds__(256) wq_a_kernel(const int64_t NM, const int M, const int QP, const int q0, const int qc,
                                                    const double* __restrict__ dm, const double* __restrict__ Zc,
                                                    double* __restrict__ A) {
  const int64_t total = NM * qc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / qc;
    const int j = (int)(i - row * qc);
    const int64_t n = row / M;
    const int m = (int)(row - n * M);
    const int q = q0 + j;
    A[i] = (q < QP) ? dm[n * QP + q] * Zc[(size_t)m * QP + q] : 0.0;
  }
}

// The exponential pass over E[(n,m), m'] (row stride M).  Workgroup = block of data rows, team of 128 threads = one inducing
// row m at a time (thread j = m'); for every m the team walks the block's rows n:
//   p = exp(E + g0[n] + H[n,m] + H[n,j]);  psi2 partial [wg][m][j] += p;  W = p dLs[m][j] -> E;  v[n,m] += sum_j W
__global__ void __launch_bounds__(256) wq_exp_kernel(const int64_t N, const int M, double* __restrict__ E,
                                                      const double* __restrict__ g0, const double* __restrict__ H,
                                                      const double* __restrict__ dLs, double* __restrict__ p2_part,
                                                      double* __restrict__ vrow) {
  const int team = threadIdx.x >> 7, j = threadIdx.x & 127;
  const bool act = j < M;
  const int64_t per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  double* __restrict__ pp = p2_part + (size_t)blockIdx.x * M * M;
  for (int m = team; m < M; m += 2) {
    const double dl = act ? dLs[(size_t)m * M + j] : 0.0;
    double acc = 0.0;
    for (int64_t n = r0; n < r1; ++n) {
      double w = 0.0;
      if (act) {
        const size_t idx = ((size_t)n * M + m) * M + j;
        const double p = exp(E[idx] + g0[n] + H[n * M + m] + H[n * M + j]);
        acc += p;
        w = p * dl;
        E[idx] = w;
      }
      w = wave_sum_dpp(w);
      if ((threadIdx.x & 63) == 0) atomicAdd(&vrow[n * M + m], w);      // two addends per element: the order cannot matter
    }
    if (act) pp[(size_t)m * M + j] = acc;
  }
}

// T[n, q0 + j] = sum_m z[m, q0 + j] Y[(n,m), j]
__global__ void __launch_bounds__(128) wq_t_kernel(const int64_t N, const int M, const int QP, const int q0, const int qc,
                                                    const double* __restrict__ Y, const double* __restrict__ Zc,
                                                    double* __restrict__ T) {
  const int j = threadIdx.x;
  const int q = q0 + j;
  if (j >= qc || q >= QP) return;
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += Zc[(size_t)m * QP + q] * Y[((size_t)n * M + m) * qc + j];
    T[n * QP + q] = s;
  }
}

// dZ3 partial [split][m][q0 + j] = sum over the split's rows n of d2[n, q0 + j] Y[(n,m), j]        (grid: M x splits)
__global__ void __launch_bounds__(128) wq_dz3_kernel(const int64_t N, const int M, const int QP, const int q0, const int qc,
                                                      const double* __restrict__ Y, const double* __restrict__ d2,
                                                      double* __restrict__ part) {
  const int j = threadIdx.x, m = blockIdx.x;
  const int q = q0 + j;
  if (j >= qc || q >= QP) return;
  const int64_t per = (N + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per;
  const int64_t r1 = (r0 + per < N) ? r0 + per : N;
  double s = 0.0;
  for (int64_t n = r0; n < r1; ++n) s += d2[n * QP + q] * Y[((size_t)n * M + m) * qc + j];
  part[((size_t)blockIdx.y * M + m) * QP + q] = s;
}

// DM[n,q] = d2[n,q] mu[n,q]
__global__ void __launch_bounds__(256) wq_dmu_kernel(int64_t n, const double* __restrict__ d2, const double* __restrict__ mu,
                                                      double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d2[i] * mu[i];
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

// psi2 forward sum, row gradients (added to L.gmu / L.gS), dz2, dlam2, wsum2 for a wide window
static rgp_status wq_psi2(rgp_handle h, hipStream_t s, const BigPlan& p, int64_t N, int M, int Q, BigWs& L, WqWs& X) {
  const int QP = p.QP;
  const int64_t NM = N * M;
  if (NM > 0x7fffffffLL) return rgp_fail(h, RGP_ERR_BAD_ARG, "wide windows: N * M exceeds 2^31 rows of the batched product");
  const int nchunk = (QP + WQ_QC - 1) / WQ_QC;
  const unsigned ablocks = (unsigned)((NM * WQ_QC + 255) / 256 < 65536 * 4 ? (NM * WQ_QC + 255) / 256 : 65536 * 4);
  // gram: E = sum over the chunks of A_c Z_c^T
  for (int c = 0; c < nchunk; ++c) {
    const int q0 = c * WQ_QC, qc = (QP - q0 < WQ_QC) ? QP - q0 : WQ_QC;
    hipLaunchKernelGGL(wq_a_kernel, dim3(ablocks), dim3(256), 0, s, NM, M, QP, q0, qc, L.dm, L.zt.Zc, X.A);
    GemmArgs g;
    g.m = (int)NM; g.n = M; g.k = qc; g.A = X.A; g.lda = qc; g.B = L.zt.Zc + q0; g.ldb = QP; g.tb = 1; g.C = X.E; g.ldc = M;
    if (c > 0) { g.gamma = 1.0; g.Cin = X.E; g.ldcin = M; }
    launch_gemm(s, g);
  }
  // exponential pass: W in place of E, forward partial sums, v
  hipLaunchKernelGGL(big_fill_kernel, dim3(big_nb256(NM)), dim3(256), 0, s, NM, X.vrow, 0.0);
  hipLaunchKernelGGL(wq_exp_kernel, dim3(X.nexp), dim3(256), 0, s, N, M, X.E, L.rt.g0, L.H, L.dLs, X.p2w_part, X.vrow);
  launch_reduce(s, X.nexp, (int64_t)M * M, (int64_t)M * M, X.p2w_part, L.psi2, 1.0, 0);
  // back: per chunk Y = W Z_c, then the two reductions of Y
  for (int c = 0; c < nchunk; ++c) {
    const int q0 = c * WQ_QC, qc = (QP - q0 < WQ_QC) ? QP - q0 : WQ_QC;
    GemmArgs g;
    g.m = (int)NM; g.n = qc; g.k = M; g.A = X.E; g.lda = M; g.B = L.zt.Zc + q0; g.ldb = QP; g.C = X.A; g.ldc = qc;
    launch_gemm(s, g);
    hipLaunchKernelGGL(wq_t_kernel, dim3((unsigned)(N < 8192 ? N : 8192)), dim3(128), 0, s, N, M, QP, q0, qc, X.A, L.zt.Zc, X.T);
    hipLaunchKernelGGL(wq_dz3_kernel, dim3(M, X.nsplit), dim3(128), 0, s, N, M, QP, q0, qc, X.A, L.rt.d2, X.dz3_part);
  }
  launch_reduce(s, X.nsplit, (int64_t)M * QP, (int64_t)M * QP, X.dz3_part, X.dz3, 1.0, 0);
  // from v alone: [a1 | vz2 | w] = V [Z | Z^2 | 1],  B1 = V^T (d2 o mu),  B2 = V^T d2
  {
    GemmArgs g;
    g.m = (int)N; g.n = p.W; g.k = M; g.A = X.vrow; g.lda = M; g.B = L.ZZ; g.ldb = p.W; g.C = X.S22; g.ldc = p.W;
    launch_gemm(s, g);
  }
  hipLaunchKernelGGL(wq_dmu_kernel, dim3(big_nb256(N * QP)), dim3(256), 0, s, N * QP, L.rt.d2, L.rt.Xmu, X.DM);
  for (int k = 0; k < 2; ++k) {
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
