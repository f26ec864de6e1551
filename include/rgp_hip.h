/* rgp_hip.h -- C ABI of librgp_hip.so: the MI355X (gfx950) implementation of the recurrent-GP variational
 * inference hot path (per-layer ARD-RBF K_uu, psi0/psi1/psi2 statistics over lagged latent windows, the M x M
 * Cholesky/Woodbury algebra of the collapsed / SVI bound, all gradients, and the window scatter-add).
 *
 * The reference (zhenwendai/RGP) is pure Python with NO FFI; its seams are duck-typed Python protocols
 * (GPy's kernel / inference-method / posterior objects).  Each entry point below names the reference interface it
 * replaces.  A maintainer of the reference would bind these with ctypes from autoreg/layers.py, as rgp_amd/_lib.py does.
 *
 * Conventions
 *  - plain C linkage, POD structs, no exceptions; every call returns rgp_status (0 = ok);
 *  - every pointer inside rgp_layer_in/out and every "dev" pointer is a DEVICE pointer to row-major fp64
 *    (or int64 where stated) owned by the caller; the library owns only its opaque workspace;
 *  - all work is enqueued on the stream given to rgp_create; no call synchronises except rgp_sync_status;
 *  - a handle is not thread-safe; handles on different devices/ranks are independent;
 *  - a non-positive-definite K_uu / q(U) covariance does NOT get GPy's silent jitter retry (jitchol): the step
 *    finishes with NaNs and rgp_sync_status reports RGP_ERR_NOT_PD with the failing pivot index.
 */
#ifndef RGP_HIP_H
#define RGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgp_ctx* rgp_handle;
typedef int rgp_status;

enum {
  RGP_OK = 0,
  RGP_ERR_BAD_ARG = 1,      /* shape / mode outside what this build supports (message in rgp_last_error) */
  RGP_ERR_HIP = 2,          /* a HIP runtime call failed */
  RGP_ERR_NOT_PD = 3,       /* Cholesky met a non-positive pivot (index in rgp_sync_status's pivot output) */
  RGP_ERR_WORKSPACE = 4     /* workspace allocation failed */
};

enum { RGP_MODE_VARDTC = 0, RGP_MODE_SVI = 1 };

/* Geometry + inputs of one sparse-GP layer evaluated on a minibatch of sequences.
 * Row n of sequence s has inputs   [ own[n .. n+X_win)  |  up[u_off+n .. u_off+n+U_win) ]   (time-major then
 * dimension, right aligned: u_off = up_len[s] - N_s - U_win + 1, N_s = own_len[s] - X_win) and target own[X_win+n]
 * -- the window rule of autoreg/layers.py:1399-1431 (Layer._update_conv/_update_X), kept implicit (never copied by
 * the caller). */
typedef struct {
  int32_t M, D;             /* inducing points; output width (= X_dim for hidden layers, D_y for the observed layer) */
  int32_t X_win, X_dim;     /* own-history window (0 = not autoregressive) and width of the own signal */
  int32_t U_win, U_dim;     /* input window and width (latent of the layer above, or the controls for the top layer) */
  int32_t n_seq;            /* sequences in this call (minibatch) */
  int32_t mode;             /* RGP_MODE_VARDTC (vardtc.py:88) or RGP_MODE_SVI (svi_vardtc.py:70) */
  int32_t y_uncertain;      /* 1: hidden layer, outputs are q(x) rows >= X_win with variances (own_var != NULL) */
  int32_t reserved;
  int64_t N;                /* total data rows of the call = sum_s (own_len[s] - X_win) */
  int64_t own_rows;         /* sum_s own_len[s] */
  int64_t up_rows;          /* sum_s up_len[s] (0 when U_win == 0) */
  double  qU_ratio;         /* n_batch / n_total (layers.py:328); multiplies KL(q(U)||p(U)) */
  double  up_var_const;     /* used when up_var == NULL (controls: 1e-10, model.py:669) */
  /* per-sequence tables, int64[n_seq] on the device: *_off = ELEMENT offset of the sequence's first row from the
   * signal's base pointer (so several signals may live in one flat parameter buffer), *_len = number of rows */
  const int64_t *own_off, *own_len, *up_off, *up_len;
  /* signals */
  const double *own_mean, *own_var;   /* hidden: latent mean/var store; observed: Y and NULL */
  const double *up_mean, *up_var;
  /* global parameters (device; scalars are 1-element arrays so no host sync is needed) */
  const double *Z;          /* M x Q,  Q = X_win*X_dim + U_win*U_dim */
  const double *variance;   /* kern.variance      (sigma_f^2) */
  const double *inv_l;      /* kern.inv_l [Q]     (1/lengthscale^2, GPy RBF(ARD=True, inv_l=True)) */
  const double *noise;      /* likelihood.variance */
  const double *qU_m;       /* SVI: M x D */
  const double *qU_W;       /* SVI: M x M, q(U) covariance = W W^T + a I (layers.py:71) */
  const double *qU_a;       /* SVI: scalar */
} rgp_layer_in;

/* Outputs of one layer step: what SparseGP_MPI._inference_vardtc (layers.py:66-176), Layer._update_qX_gradients
 * (:1452), Layer._prepare_gradients (:1460) and Layer.update_latent_gradients (:1433) leave behind. */
typedef struct {
  double *elbo;             /* [1]: logL - qU_ratio*KL + entropy/prior terms of this layer's own latent */
  double *gZ, *gvariance, *ginv_l, *gnoise;        /* M x Q, [1], [Q], [1] */
  double *gqU_m, *gqU_W, *gqU_a;                   /* SVI only */
  double *g_own_mean, *g_own_var;   /* hidden layers: WRITTEN (=) for the rows of the sequences in the call; same layout
                                       as own_mean/own_var */
  double *g_up_mean, *g_up_var;     /* ACCUMULATED (+=), same layout as up_mean/up_var; NULL for the top layer
                                       (controls are data, layers.py:1462-1465) */
  double *woodbury_inv, *woodbury_vector;          /* M x M, M x D (posterior for prediction); may be NULL */
} rgp_layer_out;

/* ---- lifecycle --------------------------------------------------------------------------------------------- */
rgp_status rgp_create(rgp_handle* out, int device, void* hip_stream);
void       rgp_destroy(rgp_handle h);
const char* rgp_last_error(rgp_handle h);
int        rgp_abi_version(void);
/* Blocks until the handle's stream is idle; returns RGP_ERR_NOT_PD / RGP_ERR_HIP if any step since the last call
 * failed.  pivot (may be NULL) receives the failing Cholesky column. */
rgp_status rgp_sync_status(rgp_handle h, int* pivot);
/* rgp_layer_step runs the M x M algebra of a layer on internal side streams so that it overlaps the psi passes of the
 * next layer; its outputs (gradients of the global parameters, elbo, woodbury_*) are complete on the handle's stream
 * only after rgp_join (a stream-side wait, no host synchronisation).  rgp_sync_status joins implicitly. */
rgp_status rgp_join(rgp_handle h);
/* Optional: start the data-independent part of a layer (K_uu, its factorisation and everything of the SVI bound that
 * needs only the parameters) early on a side stream.  Call it for the layers of a step in the order in which
 * rgp_layer_step will be called; each rgp_layer_step then consumes the oldest prepared layer (same geometry and
 * parameter pointers).  Up to 3 layers may be outstanding.  No-op for RGP_MODE_VARDTC. */
rgp_status rgp_layer_prepare(rgp_handle h, const rgp_layer_in* in);
/* Upper bound of the workspace (bytes) a layer step of this geometry needs; the workspace grows on demand. */
int64_t    rgp_workspace_bytes(int64_t N, int M, int Q, int D);

/* Optional in-library timing of the dominant kernel (the fused psi2 pass): when on, rgp_layer_step brackets that
 * launch with HIP events on the handle's stream; rgp_get_profile synchronises those events and returns the summed
 * duration [ms], the number of launches and the summed algorithmic work sum(N*M*M*Q) since the last reset. */
rgp_status rgp_set_profiling(rgp_handle h, int on);
rgp_status rgp_get_profile(rgp_handle h, double* psi2_ms, int64_t* psi2_launches, double* psi2_nmmq, int reset);

/* Summation order of the psi2 forward statistic.  By default (on = 0) workgroups of the fused psi2 pass keep several data
 * rows in flight and add their terms to a shared fp64 accumulator in hardware order: results are reproducible to ~1 ulp
 * of each entry, not bit for bit.  on = 1 selects the rotating-team kernel whose additions happen in a fixed order
 * (bit-reproducible run to run, ~23% slower at M = 100).  The environment variable RGP_PSI2_NO_PACKED=1 has the same effect. */
rgp_status rgp_set_deterministic(rgp_handle h, int on);

/* Which kernels run the fused psi2 pass of the SVI bound (kern.psi2 + its gradients, svi_vardtc.py:48-50, layers.py:98-132,
 * 1452-1458) for M <= 112, Q <= 64, and the psi2 term of the blocked layer step for 112 < M <= 1024, padded Q <= 48.
 * mode 0: the sweep kernels on the vector pipe (one thread per (data row, inducing row));
 * 1: the pair form on the fp64 matrix pipe (exponent, exponential and both back products once per UNORDERED pair of inducing
 * rows) where it pays (48 <= M <= 112 from 4000 data rows, M > 128 from 2048 data rows), the sweep kernels elsewhere; 2: the
 * pair form wherever its LDS image fits; -1 (default): the environment variable RGP_PSI2_PAIR if set, else 1.
 * rgp_set_deterministic(1) always selects the fixed-order sweep kernels.  Results agree to rounding between the modes. */
rgp_status rgp_set_psi2_pair(rgp_handle h, int mode);
/* 1 if the psi2 term of an SVI layer step with N data rows, M inducing points and Q window columns would run in the pair form
 * under the handle's current mode, 0 if in the sweep kernels (a query for benchmarks and logs; no reference counterpart). */
int        rgp_psi2_pair_would_run(rgp_handle h, int64_t N, int M, int Q);

/* The library overlaps the M x M algebra and the row preparation with the psi passes on three streams of its own.  HIP maps
 * all streams of a process onto a few hardware queues in creation order (GPU_MAX_HW_QUEUES), and work on one hardware queue
 * runs in submission order, so rgp_create picks its streams by test (a kernel on the candidate must finish while a bounded
 * spin kernel occupies the handle's stream and the streams already kept).  Returns how many of the three passed (3 = full
 * overlap; fewer: the process has fewer free hardware queues than the library has streams, results are unaffected). */
int        rgp_side_streams_independent(rgp_handle h);

/* ---- the fused layer step (replaces Layer.parameters_changed / Layer_new.update_layer + update_latent_gradients,
 *      autoreg/layers.py:1483-1488, :617-621, :1433-1450) ------------------------------------------------------
 * Sizes (Q = X_win*X_dim + U_win*U_dim; RGP_ERR_BAD_ARG beyond): 1 <= D <= 2048 (the fused SVI step for
 *   M <= 128 accepts more; the tests go up to D = 64);
 *   M <= 128,  Q <= 64  : both modes, fused kernels, side-stream pipelining (rgp_layer_prepare);
 *   M <= 1024, Q <= 64  : both modes, blocked kernels (steps run whole on the handle's stream; rgp_layer_prepare is a no-op);
 *   M <= 1024, Q <= 1024: both modes, the psi2 term as large fp64 matrix products over the pairs of inducing rows (likewise;
 *                         the N x M(M+1)/2 weights are held in the workspace for about 6 GB of data rows at a time). */
rgp_status rgp_layer_step(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out);

/* ---- pipelining consecutive SVI steps -------------------------------------------------------------------------------
 * The M x M chain of a layer needs only that layer's global parameters.  Once they have been updated, the chain of the
 * NEXT step can start while the running step is still busy with the psi passes of later layers:
 *   rgp_layer_step(L_k) ; [all-reduce of L_k's gradients on the stream rgp_get_stream(h, 1)] ;
 *   rgp_adam_step_post(L_k's global parameters)            -- ordered after L_k's gradients on the library's side stream
 *   rgp_layer_prepare(in_k with reserved = RGP_PREPARE_CHAIN_ONLY)
 * and, at the start of the next step, rgp_layer_prepare(in_k with reserved = RGP_PREPARE_ROWS_ONLY) for every layer in the
 * same order (window tables / row terms of the new minibatch; not larger than the one given to the chain-only call).
 * rgp_discard_prepared drops prepared layers that will not be stepped (parameters changed by other means). */
#define RGP_PREPARE_CHAIN_ONLY 1
#define RGP_PREPARE_ROWS_ONLY 2
/* with RGP_PREPARE_ROWS_ONLY: the sequences of this minibatch are not touched by any work in flight (e.g. the NEXT minibatch of
 * a data streamer, disjoint from the running one), so the row terms need not wait for the main stream's position; they are
 * computed behind the layer's chain-only prepare while the running step still occupies the main stream. */
#define RGP_PREPARE_ROWS_EARLY 4
rgp_status rgp_adam_step_post(rgp_handle h, int64_t n, double* x, double* p, const double* g, double* m, double* v,
                              const uint8_t* positive, double lr, double beta1, double beta2, double eps, int64_t step);
rgp_status rgp_discard_prepared(rgp_handle h);
/* which = 0: the handle's main stream; 1: the side stream that carries each layer's gradients (hipStream_t as void*). */
rgp_status rgp_get_stream(rgp_handle h, int which, void** stream);

/* ---- collapsed bound in two phases, for data-parallel use.  The reference's dead MPI stub all-reduces the sufficient
 *      statistics [psi0, psi2, YRY, psi1Y] (svi_vardtc.py:65-67); for the collapsed bound dL/dpsi2 depends on the GLOBAL
 *      psi2 (vardtc.py:141-142), so the exchange sits between the forward statistics and the M x M algebra.
 * rgp_vardtc_stats : forward pass of the local rows; packs [psi0sum, yry, psi2 (M*M), psi1Y (D*M), PsiR (M*M)] into
 *                    `stats` (device, 2 + 2*M*M + D*M doubles).  The caller sums `stats` over ranks (one all-reduce).
 * rgp_vardtc_finish: M x M algebra on the summed statistics (N_global = total rows over ranks), then the local backward
 *                    pass.  Quantities that every rank computes identically (the bound's data-independent part, dL/dnoise,
 *                    the K_uu backward term) are multiplied by replicated_weight (= 1/world), so that ONE sum all-reduce of
 *                    [elbo, global gradients] reproduces the single-device result.  Must directly follow rgp_vardtc_stats of
 *                    the same layer on this handle. */
rgp_status rgp_vardtc_stats(rgp_handle h, const rgp_layer_in* in, double* stats);
rgp_status rgp_vardtc_finish(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out, double* stats,
                             int64_t N_global, double replicated_weight);

/* ---- unit-testable pieces (dense, caller-materialised X; replace the GPy kernel protocol) --------------------
 * Sizes: M <= 1024, Q <= 1024 (the kernels are picked by size as in rgp_layer_step). */
/* kern.K(Z) + jitter*I (vardtc.py:110-114). */
rgp_status rgp_kuu(rgp_handle h, int M, int Q, const double* Z, const double* variance, const double* inv_l,
                   double jitter, double* Kuu);
/* kern.psi1(Z, q(X)) -> N x M and kern.psi2(Z, q(X)) -> M x M (vardtc.py:60-61); either output may be NULL. */
rgp_status rgp_psi_forward(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S, const double* Z,
                           const double* variance, const double* inv_l, double* psi1, double* psi2);
/* rgp_psi_forward for callers that cannot allocate.  It has NO counterpart in the reference: it exists for callers
 * that cannot allocate, such as tight serving loops and harnesses that forbid hipMalloc (it has not been exercised under HIP
 * graph capture: it waits on events of the library's side streams and sets a kernel attribute).  Same arguments, same kernels and the same results
 * (to rounding: the psi2 partial sums are combined over fewer workgroups) as rgp_psi_forward, but it calls no hipMalloc /
 * hipFree or any other allocating routine: its scratch (Z terms, row terms, partial sums) is a 20 MB static device array
 * that every GPU the library runs on holds until the process ends.  Limits: N <= 4096, M <= 128, Q <= 64; larger problems
 * are rejected with RGP_ERR_BAD_ARG.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two
 * handles, two streams) race; consecutive calls on one stream are safe (nothing is read before this call has written it). */
rgp_status rgp_psi_forward_static(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S, const double* Z,
                                  const double* variance, const double* inv_l, double* psi1, double* psi2);
/* kern.update_gradients_expectations + gradients_Z_expectations + gradients_qX_expectations (layers.py:98-132,
 * 1452-1458) for given dL_dpsi1 (N x M, may be NULL) and dL_dpsi2 (M x M, may be NULL); dL_dpsi0 enters as its sum.
 * Outputs: dvar[1], dinv_l[Q], dZ[M x Q], dmu[N x Q], dS[N x Q]. */
rgp_status rgp_psi_backward(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S, const double* Z,
                            const double* variance, const double* inv_l, double dL_dpsi0_sum, const double* dL_dpsi1,
                            const double* dL_dpsi2, double* dvar, double* dinv_l, double* dZ, double* dmu, double* dS);
/* The M x M algebra of SVI_VarDTC.inference + comp_KL_qU (svi_vardtc.py:70-215) given the statistics.
 * psi1Y is D x M (= Y^T psi1, unscaled), yry = sum(Ymean^2)+sum(Yvar).  Outputs (any may be NULL):
 * logL_minus_KL[1], dL_dKmm (M x M, includes -ratio*dKL_dKuu), dL_dpsi2 (M x M), Kinv_mu (M x D),
 * dL_dthetaL[1], gqU_m (M x D), gqU_W (M x M), gqU_a[1], woodbury_inv (M x M). */
rgp_status rgp_svi_chain(rgp_handle h, int64_t N, int M, int D, double qU_ratio, const double* Kuu_jit,
                         const double* noise, const double* qU_m, const double* qU_W, const double* qU_a,
                         const double* psi0sum, const double* psi1Y, const double* psi2, const double* yry,
                         double* logL_minus_KL, double* dL_dKmm, double* dL_dpsi2, double* Kinv_mu,
                         double* dL_dthetaL, double* gqU_m, double* gqU_W, double* gqU_a, double* woodbury_inv);

/* The fp64 matrix product every GEMM-shaped part of the path runs on (np.dot / tdot / dtrmm-style products of
 * svi_vardtc.py:88-169,197-215 once M > 128, the N x M x D products of layers.py:98-132 for wide outputs, the pair-indexed
 * psi2 products of wide windows):  C[m x n] = alpha op(A) op(B) + gamma C, row-major, op(X) = X or X^T (ta / tb != 0; A is
 * then stored k x m, B n x k).  splits > 1 cuts k into that many pieces whose partial products are summed in fixed order
 * (gamma must then be 0).  C must not overlap A or B. */
rgp_status rgp_matmul(rgp_handle h, int m, int n, int k, double alpha, const double* A, int lda, int ta, const double* B,
                      int ldb, int tb, double gamma, double* C, int ldc, int splits);

/* ---- free-run prediction with moment matching (Layer.freerun, autoreg/layers.py:1537-1571 + GPy _raw_predict) ------
 * For each of n_seq independent sequences: X[0..X_win) = init (or N(0,1) when init_* == NULL, layers.py:1545), then for
 * n = 0..steps-1 the input row [X[n..n+X_win) | U[u_skip+n .. u_skip+n+U_win)] (means AND variances) is pushed through
 * the layer's posterior (woodbury_inv M x M, woodbury_vector M x D from rgp_layer_step) and the predictive mean/variance
 * become X[X_win+n] (variance clipped at 1e-15; <= 0 -> 1e-10, layers.py:1570).  X_dim == D.
 * Layouts: init_* [n_seq][X_win*D]; U_* [n_seq][u_rows][U_dim] (U_var NULL -> u_var_const); X_* [n_seq][X_win+steps][D].
 * Sizes: M <= 1024, Q = X_win*D + U_win*U_dim <= 1024 (RGP_ERR_BAD_ARG beyond). */
rgp_status rgp_freerun(rgp_handle h, int n_seq, int M, int D, int X_win, int U_win, int U_dim, int steps,
                       const double* Z, const double* variance, const double* inv_l, const double* woodbury_inv,
                       const double* woodbury_vector, const double* init_mean, const double* init_var,
                       const double* U_mean, const double* U_var, double u_var_const, int64_t u_rows, int64_t u_skip,
                       double* X_mean, double* X_var);

/* ---- MLP back-constraint recursion as a per-sequence persistent scan --------------------------------------------
 * Replaces Layer._encoder_freerun (autoreg/layers.py:623-666) and the reverse loop of Layer._encoder_update_gradient
 * (layers.py:668-756) for the network of autoreg/mlp.py:13-175 (tanh hidden layers, linear output):
 *     mean[X_win+n] = MLP([mean[n..n+X_win) | up[u_off+n .. u_off+n+U_win)]),  n = 0..N-1,  u_off = up_len - N - U_win + 1.
 * One workgroup per sequence; all n_seq sequences have the same lengths.  dims[0] = X_win*X_dim + U_win*U_dim,
 * dims[n_layers] = X_dim; W[k] is dims[k+1] x dims[k] row-major (out x in), b[k] dims[k+1].
 * Layouts: init_mean [n_seq][X_win][X_dim]; up [n_seq][up_len][U_dim] (NULL iff U_win == 0); mean [n_seq][X_win+N][X_dim];
 * act[k] [n_seq][N][dims[k+1]] for the hidden layers k < n_layers-1 (written by the forward, read by the backward).
 * Backward: g_mean [n_seq][X_win+N][X_dim] = dL/d mean from the GP layers (rgp_layer_step) -> g_init [n_seq][X_win][X_dim]
 * (gradient of the first rows), g_up [n_seq][up_len][U_dim] (gradient of the signal above; may be NULL) and
 * delta[k] [n_seq][N][dims[k+1]] = dL/d(pre-activation of layer k at step n); the weight gradients are
 * dW[k] = sum_{seq,n} delta[k] in_k^T, db[k] = sum delta[k] (GEMMs over stored activations, done by the caller). */
#define RGP_MLP_MAX_LAYERS 4
typedef struct rgp_mlp_scan {
  int32_t n_seq, X_win, X_dim, U_win, U_dim, N, up_len, n_layers;
  int32_t dims[RGP_MLP_MAX_LAYERS + 1];
  int32_t reserved[3];
  const double* W[RGP_MLP_MAX_LAYERS];
  const double* b[RGP_MLP_MAX_LAYERS];
  const double* init_mean;
  const double* up;
  double* mean;                              /* forward output */
  double* act[RGP_MLP_MAX_LAYERS - 1];       /* forward output / backward input */
  const double* g_mean;                      /* backward input */
  double* g_init;                            /* backward outputs */
  double* g_up;
  double* delta[RGP_MLP_MAX_LAYERS];
} rgp_mlp_scan;
rgp_status rgp_mlp_scan_forward(rgp_handle h, const rgp_mlp_scan* a);
rgp_status rgp_mlp_scan_backward(rgp_handle h, const rgp_mlp_scan* a);

/* ---- recurrence of the sequence-encoder back-constraint as a per-sequence persistent scan -------------------------
 * Replaces the time loop inside PyTorch's nn.GRU / nn.RNN(tanh) / nn.LSTM as used by autoreg/rnn_encoder.py:21-83 (one
 * layer, zero initial state, PyTorch gate orders).  The input projections gi[t] = W_ih x_t + b_ih are the caller's GEMM.
 *   GRU : gh = W_hh h + b_hh; r = s(gi_r + gh_r); z = s(gi_z + gh_z); n = tanh(gi_n + r gh_n); h' = (1 - z) n + z h
 *   RNN : h' = tanh(gi + W_hh h + b_hh)
 *   LSTM: i, f, o = s(gi_. + gh_.); g = tanh(gi_g + gh_g); c' = f c + i g; h' = o tanh(c')
 * With G = 3 / 1 / 4 gate rows per unit: gi [T][B][G H]; W_hh [G H][H]; b_hh [G H]; hseq [T][B][H] (PyTorch's (seq_len, batch,
 * feature)); gates [T][B][NS H] (forward output, backward input): GRU r | z | n | gh_n (NS = 4), LSTM i | f | g | o | c (NS = 5),
 * RNN none (may be NULL).
 * Backward: g_out [T][B][H] = dL/dh_t -> dgi [T][B][G H] = dL/d gi_t and dgh [T][B][G H] = dL/d gh_t;
 * dW_hh = sum_t dgh_t h_{t-1}^T, db_hh = sum_t dgh_t (GEMMs on the caller's side).  G H <= 256 and W_hh must fit in LDS
 * (GRU: H <= 80, LSTM: H <= 64, RNN: H <= 140). */
#define RGP_CELL_GRU 0
#define RGP_CELL_RNN_TANH 1
#define RGP_CELL_LSTM 2
typedef struct rgp_rnn_scan {
  int32_t T, B, H, cell;
  const double* gi;
  const double* W_hh;
  const double* b_hh;
  double* hseq;
  double* gates;
  const double* g_out;
  double* dgi;
  double* dgh;
} rgp_rnn_scan;
rgp_status rgp_rnn_scan_forward(rgp_handle h, const rgp_rnn_scan* a);
rgp_status rgp_rnn_scan_backward(rgp_handle h, const rgp_rnn_scan* a);

/* ---- optimiser step on device (the outer SVI loop's Adam; positive parameters live in softplus space) -------- */
/* x, m, v: n doubles of optimiser-space parameter and Adam moments; g: gradient w.r.t. the CONSTRAINED value p;
 * positive[i] != 0 -> p = softplus(x) and the chain factor 1-exp(-p) is applied (paramz Logexp).  The new constrained
 * values are written to p.  Maximises (gradient ASCENT on the ELBO).  step is the 1-based Adam step count. */
rgp_status rgp_adam_step(rgp_handle h, int64_t n, double* x, double* p, const double* g, double* m, double* v,
                         const uint8_t* positive, double lr, double beta1, double beta2, double eps, int64_t step);

#ifdef __cplusplus
}
#endif
#endif /* RGP_HIP_H */
