// rgp_layer.hip -- C ABI of librgp_hip.so and the per-layer orchestration (see include/rgp_hip.h).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rgp_chain.h"
#include "rgp_gemm.h"
#include "rgp_psi.h"
#ifdef RGP_GAPPROF
void rgp_gap_dump();
#endif

// ==================================================================================================================
// static device scratch of rgp_psi_forward_static (the entry point that allocates nothing, include/rgp_hip.h)
// ==================================================================================================================
// Each GPU the library runs on holds its own copy of this array until the process ends.  In host code the array's NAME is
// only a stand-in for the symbol, not a device address: the address is looked up once per device with hipGetSymbolAddress
// (rgp_create, that device current) and every later use -- kernel arguments, copies -- goes through that pointer.
#define RGP_STATIC_MAX_N 4096
#define RGP_STATIC_NBLK 128
#define RGP_STATIC_SCRATCH_BYTES (20u << 20)
static __device__ __align__(256) unsigned char rgp_static_scratch[RGP_STATIC_SCRATCH_BYTES];
#define RGP_MAX_DEVICES 64
static char* g_static_scratch_ptr[RGP_MAX_DEVICES] = {};

static hipError_t static_scratch_lookup(int device, char** out) {
  if (device < 0 || device >= RGP_MAX_DEVICES) return hipErrorInvalidDevice;
  if (!g_static_scratch_ptr[device]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rgp_static_scratch));       // the device is current (rgp_create)
    if (e != hipSuccess) return e;
    g_static_scratch_ptr[device] = static_cast<char*>(p);
  }
  *out = g_static_scratch_ptr[device];
  return hipSuccess;
}

// ==================================================================================================================
// handle
// ==================================================================================================================
rgp_status rgp_fail(rgp_handle h, rgp_status code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

void rgp_join_streams(rgp_handle h) {
  for (int k = 0; k < rgp_ctx::NSLOT; ++k)
    if (h->done_recorded[k]) (void)hipStreamWaitEvent(h->stream, h->ev_done[k], 0);
  if (h->post_tail_recorded) (void)hipStreamWaitEvent(h->stream, h->ev_post_tail, 0);
}

extern "C" rgp_status rgp_join(rgp_handle h) {
  if (!h) return RGP_ERR_BAD_ARG;
  rgp_join_streams(h);
  return RGP_OK;
}

rgp_status rgp_ensure_ws_slot(rgp_handle h, int slot, size_t bytes) {
  if (bytes <= h->ws_slot_bytes[slot]) return RGP_OK;
  (void)hipStreamSynchronize(h->stream);
  if (h->side_pre) (void)hipStreamSynchronize(h->side_pre);
  if (h->side_post) (void)hipStreamSynchronize(h->side_post);
  if (h->side_rows) (void)hipStreamSynchronize(h->side_rows);
  if (h->ws_slot[slot]) (void)hipFree(h->ws_slot[slot]);
  h->ws_slot[slot] = nullptr; h->ws_slot_bytes[slot] = 0;
  bytes = bytes + bytes / 8 + (1u << 20);
  if (hipMalloc((void**)&h->ws_slot[slot], bytes) != hipSuccess) return rgp_fail(h, RGP_ERR_WORKSPACE, "workspace hipMalloc failed");
  h->ws_slot_bytes[slot] = bytes;
  return RGP_OK;
}

rgp_status rgp_ensure_ws(rgp_handle h, size_t bytes) {
  rgp_join_streams(h);                      // the unit entry points share nothing with in-flight layer steps
  if (bytes <= h->ws_bytes) return RGP_OK;
  if (h->ws) {
    hipStreamSynchronize(h->stream);
    hipFree(h->ws);
    h->ws = nullptr; h->ws_bytes = 0;
  }
  bytes = bytes + bytes / 8 + (1u << 20);
  if (hipMalloc((void**)&h->ws, bytes) != hipSuccess) return rgp_fail(h, RGP_ERR_WORKSPACE, "workspace hipMalloc failed");
  h->ws_bytes = bytes;
  return RGP_OK;
}

extern "C" int rgp_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------------------------------
// Side streams that really run next to the main stream.  HIP multiplexes all streams of a process onto a few hardware
// queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and packets of one hardware queue execute in submission order.
// A process that has created other streams first (an RCCL communicator makes six) can therefore hand this library a side
// stream that shares the main stream's queue: the M x M chain of the next step then sits in front of the next layer's psi
// passes and the step serialises (measured: 2.9 -> 6.2 ms per step at 4 sequences per rank).  So the streams are picked by
// test: a candidate is kept if a kernel launched on it finishes while a bounded spin kernel still occupies the main stream
// and every stream kept so far.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) stream_probe_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();                  // constant-rate counter (100 MHz): the loop ends after `ticks`
  while (wall_clock64() - t0 < ticks) {}
}
__global__ void __launch_bounds__(64) stream_probe_noop_kernel() {}

static bool streams_run_side_by_side_once(hipStream_t busy, hipStream_t cand, hipEvent_t eb, hipEvent_t ec, long long ticks) {
  hipLaunchKernelGGL(stream_probe_spin_kernel, dim3(1), dim3(64), 0, busy, ticks);
  (void)hipEventRecord(eb, busy);
  hipLaunchKernelGGL(stream_probe_noop_kernel, dim3(1), dim3(64), 0, cand);
  (void)hipEventRecord(ec, cand);
  (void)hipEventSynchronize(ec);
  const bool side_by_side = hipEventQuery(eb) == hipErrorNotReady;
  (void)hipGetLastError();                              // hipErrorNotReady is sticky for hipGetLastError
  (void)hipEventSynchronize(eb);
  return side_by_side;
}
// ~300 us of spin; a candidate that fails is tried once more against ~3 ms before it is discarded: on a GPU shared with
// other processes the no-op kernel of a good candidate can be late, while one that shares the busy stream's hardware queue
// fails at any length.
static bool streams_run_side_by_side(hipStream_t busy, hipStream_t cand, hipEvent_t eb, hipEvent_t ec) {
  return streams_run_side_by_side_once(busy, cand, eb, ec, 30000LL) || streams_run_side_by_side_once(busy, cand, eb, ec, 300000LL);
}

// Fills side[0..2]; returns how many of them passed the test against the main stream and each other (3 = all).
static int pick_side_streams(hipStream_t main_stream, hipStream_t side[3]) {
  const bool probe = getenv("RGP_NO_STREAM_PROBE") == nullptr;
  hipEvent_t eb, ec;
  (void)hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&ec, hipEventDisableTiming);
  hipStream_t rejected[16];
  int n_ok = 0, n_rej = 0;
  for (int tries = 0; n_ok < 3 && tries < 16; ++tries) {
    hipStream_t c = nullptr;
    if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
    bool ok = true;
    if (probe) {
      hipLaunchKernelGGL(stream_probe_noop_kernel, dim3(1), dim3(64), 0, c);     // the stream's queue exists before the test
      (void)hipStreamSynchronize(c);
      ok = streams_run_side_by_side(main_stream, c, eb, ec);
      for (int k = 0; ok && k < n_ok; ++k) ok = streams_run_side_by_side(side[k], c, eb, ec);
    }
    if (ok) side[n_ok++] = c; else rejected[n_rej++] = c;
  }
  const int passed = n_ok;
  while (n_ok < 3 && n_rej > 0) side[n_ok++] = rejected[--n_rej];               // fewer hardware queues than streams: share
  while (n_rej > 0) (void)hipStreamDestroy(rejected[--n_rej]);
  (void)hipEventDestroy(eb);
  (void)hipEventDestroy(ec);
  if (n_ok == 3) return passed;
  while (n_ok > 0) { (void)hipStreamDestroy(side[--n_ok]); side[n_ok] = nullptr; }     // could not create three streams at all
  return -1;
}

extern "C" void rgp_destroy(rgp_handle h);
extern "C" rgp_status rgp_create(rgp_handle* out, int device, void* hip_stream) {
  if (!out) return RGP_ERR_BAD_ARG;
  rgp_ctx* h = new rgp_ctx();
  h->device = device;
  h->stream = (hipStream_t)hip_stream;
  if (hipSetDevice(device) != hipSuccess) { delete h; return RGP_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cu = prop.multiProcessorCount;
  if (static_scratch_lookup(device, &h->static_scratch) != hipSuccess) { rgp_destroy(h); return RGP_ERR_HIP; }
  if (hipMalloc((void**)&h->status, 2 * sizeof(int)) != hipSuccess) { h->status = nullptr; rgp_destroy(h); return RGP_ERR_HIP; }
  hipMemsetAsync(h->status, 0, 2 * sizeof(int), h->stream);
  h->overlap = getenv("RGP_NO_OVERLAP") == nullptr;
  {
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    h->side_streams_independent = pick_side_streams(h->stream, side);
    if (h->side_streams_independent < 0) { rgp_destroy(h); return RGP_ERR_HIP; }      // frees the status words too
    h->side_pre = side[0]; h->side_post = side[1]; h->side_rows = side[2];
  }
  hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming);
  hipEventCreateWithFlags(&h->ev_post_pos, hipEventDisableTiming);
  hipEventCreateWithFlags(&h->ev_post_tail, hipEventDisableTiming);
  hipEventCreateWithFlags(&h->ev_prep, hipEventDisableTiming);
  hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming);
  for (int k = 0; k < rgp_ctx::NSLOT; ++k) {
    hipEventCreateWithFlags(&h->ev_pre[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_pre_a[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_rows[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_yry[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_prepz[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_pre_b[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_slot_main[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_done[k], hipEventDisableTiming);
  }
  *out = h;
  return RGP_OK;
}

extern "C" void rgp_destroy(rgp_handle h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->side_pre) { hipStreamSynchronize(h->side_pre); hipStreamDestroy(h->side_pre); }
  if (h->side_post) { hipStreamSynchronize(h->side_post); hipStreamDestroy(h->side_post); }
  if (h->side_rows) { hipStreamSynchronize(h->side_rows); hipStreamDestroy(h->side_rows); }
  if (h->ws) hipFree(h->ws);
  for (int k = 0; k < rgp_ctx::NSLOT; ++k) if (h->ws_slot[k]) hipFree(h->ws_slot[k]);
  if (h->status) hipFree(h->status);
  hipEvent_t single[] = {h->ev_main, h->ev_post_pos, h->ev_post_tail, h->ev_prep, h->ev_stats};
  for (hipEvent_t e : single) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < rgp_ctx::NSLOT; ++k) {
    hipEvent_t slot[] = {h->ev_pre[k], h->ev_pre_a[k], h->ev_rows[k], h->ev_yry[k], h->ev_prepz[k], h->ev_pre_b[k], h->ev_slot_main[k],
                         h->ev_done[k]};
    for (hipEvent_t e : slot) if (e) (void)hipEventDestroy(e);
  }
  for (auto& pe : h->prof_events) { (void)hipEventDestroy(pe.first); (void)hipEventDestroy(pe.second); }
  delete h;
}

extern "C" const char* rgp_last_error(rgp_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" rgp_status rgp_sync_status(rgp_handle h, int* pivot) {
  if (!h) return RGP_ERR_BAD_ARG;
  int st[2] = {0, 0};
  rgp_join_streams(h);
  RGP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
#ifdef RGP_GAPPROF
  rgp_gap_dump();
#endif
  RGP_HIP_CHECK(h, hipMemcpy(st, h->status, sizeof(st), hipMemcpyDeviceToHost));
  if (st[0] != 0) {
    hipMemset(h->status, 0, 2 * sizeof(int));
    if (pivot) *pivot = st[1];
    char buf[128];
    snprintf(buf, sizeof(buf), "Cholesky: matrix not positive definite at column %d (no jitter retry, see rgp_hip.h)", st[1]);
    return rgp_fail(h, (rgp_status)st[0], buf);
  }
  return RGP_OK;
}

extern "C" int rgp_side_streams_independent(rgp_handle h) { return h ? h->side_streams_independent : -1; }

extern "C" rgp_status rgp_set_deterministic(rgp_handle h, int on) {
  if (!h) return RGP_ERR_BAD_ARG;
  h->deterministic = on != 0;
  return RGP_OK;
}

extern "C" rgp_status rgp_set_psi2_pair(rgp_handle h, int mode) {
  if (!h || mode < -1 || mode > 2) return RGP_ERR_BAD_ARG;
  h->psi2_pair = mode;
  return RGP_OK;
}

extern "C" int rgp_psi2_pair_would_run(rgp_handle h, int64_t N, int M, int Q) {
  if (!h || N < 1 || M < 1 || Q < 1 || Q > RGP_MAX_QP) return 0;
  if (M > RGP_MAX_M) {                            // blocked layer step: the blocked form
    PairBlkGeom gb;
    return psi2_pairblk_selected(h, N, M, rgp_qp(Q), (M + 63) / 64 * 64, gb) ? 1 : 0;
  }
  PairGeom g;
  return psi2_pair_selected(h, N, M, rgp_qp(Q), g) ? 1 : 0;
}

extern "C" rgp_status rgp_set_profiling(rgp_handle h, int on) {
  if (!h) return RGP_ERR_BAD_ARG;
  h->profiling = on != 0;
  return RGP_OK;
}

extern "C" rgp_status rgp_get_profile(rgp_handle h, double* psi2_ms, int64_t* psi2_launches, double* psi2_nmmq, int reset) {
  if (!h) return RGP_ERR_BAD_ARG;
  double tot = 0.0;
  for (size_t i = 0; i < h->prof_used; ++i) {
    float ms = 0.f;
    RGP_HIP_CHECK(h, hipEventSynchronize(h->prof_events[i].second));
    RGP_HIP_CHECK(h, hipEventElapsedTime(&ms, h->prof_events[i].first, h->prof_events[i].second));
    tot += ms;
  }
  if (psi2_ms) *psi2_ms = tot;
  if (psi2_launches) *psi2_launches = (int64_t)h->prof_used;
  if (psi2_nmmq) *psi2_nmmq = h->prof_nmmq;
  if (reset) { h->prof_used = 0; h->prof_nmmq = 0.0; }
  return RGP_OK;
}

#ifdef RGP_GAPPROF      // profiling builds only (make CXXFLAGS+=-DRGP_GAPPROF): where the main stream waits inside a layer step
#include <chrono>
struct GapRec { hipEvent_t e[4]; double host_us; };
static std::vector<GapRec> g_gap;
static void gap_mark(rgp_handle h, int k, bool first) {
  if (first) {
    GapRec r{};
    for (auto& e : r.e) (void)hipEventCreate(&e);
    r.host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    g_gap.push_back(r);
  }
  (void)hipEventRecord(g_gap.back().e[k], h->stream);
}
void rgp_gap_dump() {
  const size_t n = g_gap.size();
  if (n < 14) return;
  const size_t a = n - 13;
  for (size_t i = a; i < n; ++i) {
    float t[4];
    for (int k = 0; k < 4; ++k) (void)hipEventElapsedTime(&t[k], g_gap[a].e[0], g_gap[i].e[k]);
    fprintf(stderr, "[gap] call %zu host %+9.1f us | main reaches step %9.1f, rows+pre_a ready %9.1f, pre_b ready %9.1f, main work done %9.1f us\n",
            i - a, g_gap[i].host_us - g_gap[a].host_us, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3);
  }
}
#define GAP_MARK(k, first) gap_mark(h, k, first)
#else
#define GAP_MARK(k, first) do { } while (0)
#endif

static void prof_begin(rgp_handle h, double nmmq) {
  if (!h->profiling) return;
  if (h->prof_used == h->prof_events.size()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    h->prof_events.push_back({a, b});
  }
  h->prof_nmmq += nmmq;
  (void)hipEventRecord(h->prof_events[h->prof_used].first, h->stream);
}
static void prof_end(rgp_handle h) {
  if (!h->profiling) return;
  (void)hipEventRecord(h->prof_events[h->prof_used].second, h->stream);
  h->prof_used++;
}

// ==================================================================================================================
// small kernels of the layer step
// ==================================================================================================================
// dLdpsi1[n,m] = sum_d Ym[n,d] Cpsi1[m,d];  dYm[n,d] = -beta Ym[n,d] + sum_m psi1[n,m] Cpsi1[m,d]   (svi_vardtc.py:164,192)
__global__ void __launch_bounds__(256) svi_rows_kernel(int64_t N, int M, int D, const double* __restrict__ Ym,
                                                        const double* __restrict__ Cpsi1, const double* __restrict__ psi1,
                                                        const double* __restrict__ noise, double* __restrict__ dLdpsi1,
                                                        double* __restrict__ dYm) {
  extern __shared__ double lds[];
  const int DS = D + 1;
  double* cs = lds;                       // [M x DS]
  for (int i = threadIdx.x; i < M * D; i += blockDim.x) cs[(i / D) * DS + (i % D)] = Cpsi1[i];
  __syncthreads();
  const double beta = 1.0 / noise[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wpb = blockDim.x >> 6;
  const int m0 = lane < M ? lane : M - 1, m1 = lane + 64 < M ? lane + 64 : M - 1;     // M <= 128: two rows per lane
  for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < N; r += (int64_t)gridDim.x * wpb) {
    // the row's y and psi1 values live in lanes and are broadcast with v_readlane (no serialised scalar loads)
    double a0 = 0.0, a1 = 0.0;
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int dn = D - d0 < 64 ? D - d0 : 64;
      const double yv = lane < dn ? Ym[r * D + d0 + lane] : 0.0;
      for (int d = 0; d < dn; ++d) {
        const double yd = readlane_f64(yv, d);
        a0 += yd * cs[m0 * DS + d0 + d];
        a1 += yd * cs[m1 * DS + d0 + d];
      }
    }
    if (lane < M) dLdpsi1[r * M + lane] = a0;
    if (lane + 64 < M) dLdpsi1[r * M + lane + 64] = a1;
    if (dYm) {
      const double p0 = lane < M ? psi1[r * M + lane] : 0.0;
      const double p1 = lane + 64 < M ? psi1[r * M + lane + 64] : 0.0;
      for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        const int dc = d < D ? d : D - 1;
        double acc = 0.0;
        const int Ma = M < 64 ? M : 64;
        for (int m = 0; m < Ma; ++m) acc += readlane_f64(p0, m) * cs[m * DS + dc];
        for (int m = 64; m < M; ++m) acc += readlane_f64(p1, m - 64) * cs[m * DS + dc];
        if (d < D) dYm[r * D + d] = -beta * Ym[r * D + d] + acc;
      }
    }
  }
}

// out partial [blk][k][p] = sum over the block's rows of A[n,k] B[n,p]        (psi1Y = Ym^T psi1, vardtc.py:83)
#define CM_TILE 32
#define CM_MAXK 32
__global__ void __launch_bounds__(256) colmm_kernel(int64_t N, int K, int P, const double* __restrict__ A,
                                                     const double* __restrict__ B, const double* __restrict__ w,
                                                     double* __restrict__ part) {
  extern __shared__ double lds[];
  double* a_s = lds;                          // [CM_TILE x K]
  double* b_s = a_s + (size_t)CM_TILE * K;    // [CM_TILE x P]
  const int G = blockDim.x / P;
  const int p = threadIdx.x % P, g = threadIdx.x / P;
  const int k0 = blockIdx.y * G * CM_MAXK;    // this workgroup's chunk of output rows k
  const bool act = g < G;
  double acc[CM_MAXK];
#pragma unroll
  for (int k = 0; k < CM_MAXK; ++k) acc[k] = 0.0;
  const int64_t per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = (r0 + per < N) ? r0 + per : N;
  for (int64_t rb = r0; rb < r1; rb += CM_TILE) {
    const int nr = (int)((r1 - rb) < CM_TILE ? (r1 - rb) : CM_TILE);
    __syncthreads();
    for (int i = threadIdx.x; i < nr * K; i += blockDim.x) a_s[i] = A[rb * K + i];
    for (int i = threadIdx.x; i < nr * P; i += blockDim.x) b_s[i] = B[rb * P + i] * (w ? w[rb + i / P] : 1.0);
    __syncthreads();
    if (act) {
      for (int r = 0; r < nr; ++r) {
        const double b = b_s[r * P + p];
#pragma unroll
        for (int k = 0; k < CM_MAXK; ++k) {
          const int kk = k0 + g + k * G;
          if (kk < K) acc[k] += a_s[r * K + kk] * b;
        }
      }
    }
  }
  if (act) {
#pragma unroll
    for (int k = 0; k < CM_MAXK; ++k) {
      const int kk = k0 + g + k * G;
      if (kk < K) part[((size_t)blockIdx.x * K + kk) * P + p] = acc[k];
    }
  }
}

// out [K x P] = sum_n w_n A[n,:]^T B[n,:]   (fixed-order reduction of per-workgroup partials)
static void launch_colmm(hipStream_t s, int64_t N, int K, int P, const double* A, const double* B, const double* w,
                         double* part, double* out, int nblk) {
  const int G = 256 / P;
  const int kchunks = (K + G * CM_MAXK - 1) / (G * CM_MAXK);
  const size_t lds = ((size_t)CM_TILE * K + (size_t)CM_TILE * P) * sizeof(double);
  hipLaunchKernelGGL(colmm_kernel, dim3(nblk, kchunks), dim3(256), lds, s, N, K, P, A, B, w, part);
  launch_reduce(s, nblk, (int64_t)K * P, (int64_t)K * P, part, out, 1.0, 0);
}

__global__ void __launch_bounds__(256) rowsum_kernel(int64_t N, int D, const double* __restrict__ X, double* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  double s = 0.0;
  for (int d = 0; d < D; ++d) s += X[r * D + d];
  out[r] = s;
}

// Row part of the collapsed bound (vardtc.py:177-205), one wave per row:
//   T = psi1 G2 (G2 = L^-T L^-1);  dL_dpsi1 = beta Ym v + beta^2 r^2 T   (r^2 = sum_d Yvar; second term only for
//   uncertain outputs);  dL_dYmean = -beta Ym + beta psi1 v^T;  dL_dYvar = -beta/2 + beta^2/2 T.psi1   (shape (N,))
__global__ void __launch_bounds__(256) vardtc_rows_kernel(int64_t N, int M, int D, int y_unc, const double* __restrict__ Ym,
                                                           const double* __restrict__ r2, const double* __restrict__ psi1,
                                                           const double* __restrict__ vT, const double* __restrict__ G2,
                                                           const double* __restrict__ beta_p, double* __restrict__ dLdpsi1,
                                                           double* __restrict__ dYm, double* __restrict__ dYvar) {
  extern __shared__ double lds[];
  const int DS = D + 1, MS = M + 1;
  double* vs = lds;                                   // [M x DS]
  double* gs = vs + (size_t)M * DS;                   // [M x MS]   (uncertain outputs only)
  for (int i = threadIdx.x; i < M * D; i += blockDim.x) vs[(i / D) * DS + (i % D)] = vT[i];
  if (y_unc) for (int i = threadIdx.x; i < M * M; i += blockDim.x) gs[(i / M) * MS + (i % M)] = G2[i];
  __syncthreads();
  const double beta = beta_p[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < N; r += (int64_t)gridDim.x * wpb) {
    const double* __restrict__ y = Ym + r * D;
    const double* __restrict__ p = psi1 + r * M;
    const double rr = y_unc ? r2[r] : 0.0;
    double tp = 0.0;
    for (int m = lane; m < M; m += 64) {
      double acc = 0.0;
      for (int d = 0; d < D; ++d) acc += y[d] * vs[m * DS + d];
      acc *= beta;
      if (y_unc) {
        double tt = 0.0;
        for (int k = 0; k < M; ++k) tt += p[k] * gs[k * MS + m];
        acc += beta * beta * rr * tt;
        tp += tt * p[m];
      }
      dLdpsi1[r * M + m] = acc;
    }
    if (y_unc) {
      tp = wave_sum(tp);
      if (lane == 0) dYvar[r] = -0.5 * beta + 0.5 * beta * beta * tp;
      for (int d = lane; d < D; d += 64) {
        double acc = 0.0;
        for (int m = 0; m < M; ++m) acc += p[m] * vs[m * DS + d];
        dYm[r * D + d] = -beta * y[d] + beta * acc;
      }
    }
  }
}

// dYm -= Ym / noise    (the -beta Ym term of dL/dYmean, svi_vardtc.py:192)
__global__ void __launch_bounds__(256) dym_noise_kernel(int64_t n, const double* __restrict__ Ym, const double* __restrict__ noise,
                                                         double* __restrict__ dYm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dYm[i] -= Ym[i] / noise[0];
}

__global__ void symmetrise_kernel(int M, const double* __restrict__ A, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M * M) out[i] = 0.5 * (A[i] + A[(i % M) * M + i / M]);
}

__global__ void psi0sum_kernel(int64_t N, const double* variance, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)N * variance[0];
}

// Kernel-hyperparameter gradients: psi1 + psi2 partial sums, the Zdist factor of psi2 and the K_uu backward
// (kern.update_gradients_full / gradients_X on dL_dKmm, layers.py:105,134).
struct FinalizeArgs {
  int M, Q, QP;
  const double *Zc, *lam, *variance;
  const double *Kj;        // K_uu + jitter I (may be null when dKmm is null)
  const double *dKmm;      // may be null
  const double *dLs, *psi2;  // symmetrised dL_dpsi2 and summed psi2 (may be null)
  const double *dz1;       // [M x QP] psi1 part (may be null)
  const double *dz2;       // [QP x 128] psi2 part, transposed layout (may be null)
  int dz2_ld;              // row stride of dz2 (0: RGP_PSI2_TEAM)
  const double *dlam1, *dlam2, *lsum1, *wsum2;    // may be null
  double dk_scale;         // weight of the K_uu backward term (1, or 1/world for the data-parallel collapsed bound)
  double psi0_coef;        // sum_n dL_dpsi0[n]  (d psi0/d variance = 1)
  const double* psi0_dev;  // optional device scalar c: adds c[0]*psi0_dev_scale (used for -D beta N / 2)
  double psi0_dev_scale;
  double *gZ, *ginv_l, *gvariance;   // [M x Q], [Q], [1]
};

// grid = M workgroups (one inducing row each; blockIdx.y: 64 window columns each) of 64 x 4 threads: lane q, 4 slices of m'.
// Writes gZ[m,:], the per-row dinv_l contributions dlrow[m,:] and the per-row sum of dKmm*K.
__global__ void __launch_bounds__(256) finalize_rows_kernel(FinalizeArgs a, double* __restrict__ dlrow,
                                                             double* __restrict__ gsum_row) {
  __shared__ double sm[2][4][64];
  __shared__ double scratch[16];
  const int M = a.M, Q = a.Q, QP = a.QP;
  const int m = blockIdx.x, ql = threadIdx.x & 63, q = (int)blockIdx.y * 64 + ql, sl = threadIdx.x >> 6;
  double s1 = 0.0, s2 = 0.0, gs = 0.0;
  const double zm = (q < QP) ? a.Zc[m * QP + q] : 0.0;
  for (int mp = sl; mp < M; mp += 4) {
    double gk = 0.0, gkT = 0.0, ws = 0.0;
    if (a.dKmm) {
      const double k1 = a.Kj[m * M + mp] - (m == mp ? RGP_JITTER : 0.0);
      gk = a.dk_scale * a.dKmm[m * M + mp] * k1;
      gkT = a.dk_scale * a.dKmm[mp * M + m] * k1;
      if (ql == 0) gs += gk;
    }
    if (a.dLs) ws = a.dLs[m * M + mp] * a.psi2[m * M + mp];
    if (q < QP) {
      const double df = zm - a.Zc[mp * QP + q];
      s1 += (gk + gkT + ws) * df;
      s2 += (gk + 0.5 * ws) * df * df;
    }
  }
  sm[0][sl][ql] = s1;
  sm[1][sl][ql] = s2;
  const double gtot = block_sum(gs, scratch);
  __syncthreads();
  if (sl == 0 && q < QP) {
    s1 = (sm[0][0][ql] + sm[0][1][ql]) + (sm[0][2][ql] + sm[0][3][ql]);
    s2 = (sm[1][0][ql] + sm[1][1][ql]) + (sm[1][2][ql] + sm[1][3][ql]);
    if (q < Q) {
      double dz = -a.lam[q] * s1;
      if (a.dz1) dz += a.dz1[m * QP + q];
      if (a.dz2) dz += a.dz2[(size_t)q * (a.dz2_ld ? a.dz2_ld : RGP_PSI2_TEAM) + m];
      a.gZ[m * Q + q] = dz;
    }
    dlrow[m * QP + q] = (q < Q) ? -0.5 * s2 : 0.0;
  }
  if (threadIdx.x == 0 && blockIdx.y == 0) gsum_row[m] = gtot;
}

__global__ void __launch_bounds__(256) finalize_kernel(FinalizeArgs a, const double* __restrict__ dlrow,
                                                        const double* __restrict__ gsum_row) {
  const int M = a.M, Q = a.Q, QP = a.QP, t = threadIdx.x;
  const double var = a.variance[0];
  for (int q = t; q < Q; q += 256) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += dlrow[m * QP + q];
    if (a.dlam1) s += a.dlam1[q];
    if (a.dlam2) s += a.dlam2[q];
    a.ginv_l[q] = s;
  }
  if (t == 0) {
    double gsum = 0.0;
    for (int m = 0; m < M; ++m) gsum += gsum_row[m];
    double g = a.psi0_coef + gsum / var;
    if (a.psi0_dev) g += a.psi0_dev[0] * a.psi0_dev_scale;
    if (a.lsum1) g += a.lsum1[0] / var;
    if (a.wsum2) g += 2.0 * a.wsum2[0] / var;
    a.gvariance[0] = g;
  }
}

static void launch_finalize(hipStream_t s, const FinalizeArgs& f, double* dlrow, double* gsum_row) {
  hipLaunchKernelGGL(finalize_rows_kernel, dim3(f.M, (f.QP + 63) / 64), dim3(256), 0, s, f, dlrow, gsum_row);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, f, dlrow, gsum_row);
}

// -D beta N / 2 (= sum_n dL_dpsi0, svi_vardtc.py:162) as a device scalar
__global__ void psi0_coef_kernel(int64_t N, int D, const double* noise, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = -0.5 * (double)D * (double)N / noise[0];
}
__global__ void psi0_coef_beta_kernel(int64_t N, int D, const double* beta, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = -0.5 * (double)D * (double)N * beta[0];
}

// All per-step tables of a layer in one launch: row offsets, prefix sums of the own / upper signal lengths, and the two
// psi0 scalars (psi0 = N variance; its coefficient in the bound -D N / (2 noise)).
__global__ void svi_tables_kernel(int n_seq, int X_win, const int64_t* __restrict__ own_len, const int64_t* __restrict__ up_len,
                                  int64_t* row_off, int64_t* own_pre, int64_t* up_pre, int64_t N, int D,
                                  const double* variance, const double* noise, double* psi0sum, double* psi0coef) {
  const int t = threadIdx.x;
  if (t == 0) {
    int64_t acc = 0;
    for (int s = 0; s < n_seq; ++s) { row_off[s] = acc; acc += own_len[s] - X_win; }
    row_off[n_seq] = acc;
  } else if (t == 1) {
    int64_t acc = 0;
    for (int s = 0; s < n_seq; ++s) { own_pre[s] = acc; acc += own_len[s]; }
    own_pre[n_seq] = acc;
  } else if (t == 2 && up_len) {
    int64_t acc = 0;
    for (int s = 0; s < n_seq; ++s) { up_pre[s] = acc; acc += up_len[s]; }
    up_pre[n_seq] = acc;
  } else if (t == 3) {
    psi0sum[0] = (double)N * variance[0];
    psi0coef[0] = -0.5 * (double)D * (double)N / noise[0];
  }
}

__global__ void prefix_kernel(int n_seq, const int64_t* __restrict__ len, int64_t* pre) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t acc = 0;
    for (int s = 0; s < n_seq; ++s) { pre[s] = acc; acc += len[s]; }
    pre[n_seq] = acc;
  }
}

static __device__ __forceinline__ int find_seq(const int64_t* __restrict__ pre, int n_seq, int64_t r) {
  int lo = 0, hi = n_seq;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// Gradients w.r.t. the layer's own latent (hidden layers): output-side terms + entropy/prior (layers.py:1460-1481,
// variational.py:4-25) + the own-history part of the window scatter (layers.py:1443-1444) written as a gather so the
// summation order is fixed.  One thread per latent element.
struct LatentArgs {
  WindowSpec w; int QP; int64_t own_rows;
  const int64_t* own_pre;
  const double *gmu, *gS, *dYm, *noise;
  const double* dYvar_row;        // collapsed bound: (N,) array (vardtc.py:205); null -> -beta/2 (svi_vardtc.py:193)
  double *g_own_mean, *g_own_var, *elbo_part;
};

__global__ void __launch_bounds__(256) own_latent_kernel(LatentArgs a) {
  __shared__ double scratch[16];
  const int Xd = a.w.X_dim, Xw = a.w.X_win;
  const int64_t total = a.own_rows * Xd;
  const double beta = 1.0 / a.noise[0];
  double el = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / Xd;
    const int d = (int)(i % Xd);
    const int s = find_seq(a.own_pre, a.w.n_seq, row);
    const int64_t tt = row - a.own_pre[s];
    const int64_t Ns = a.w.own_len[s] - Xw;
    const int64_t gi = a.w.own_off[s] + tt * Xd + d;
    const double mu = a.w.own_mean[gi], S = a.w.own_var[gi];
    double gm, gv;
    if (tt >= Xw) {
      const int64_t r = a.w.row_off[s] + (tt - Xw);
      gm = a.dYm[r * Xd + d];
      gv = (a.dYvar_row ? a.dYvar_row[r] : -0.5 * beta) + 0.5 / S;   // svi_vardtc.py:193 / vardtc.py:205 + variational.py:12
      el += 0.5 * (1.0 + 1.8378770664093453 + log(S));              // -NormalEntropy.comp_value
    } else {
      gm = -mu;                                                     // variational.py:23
      gv = -0.5 * (1.0 - 1.0 / S);                                  // variational.py:24
      el -= 0.5 * (mu * mu + S - log(S)) - 0.5;                     // -NormalPrior.comp_value
    }
    for (int j = 0; j < Xw; ++j) {
      const int64_t n = tt - j;
      if (n >= 0 && n < Ns) {
        const int64_t r = a.w.row_off[s] + n;
        gm += a.gmu[r * a.QP + j * Xd + d];
        gv += a.gS[r * a.QP + j * Xd + d];
      }
    }
    a.g_own_mean[gi] = gm;
    a.g_own_var[gi] = gv;
  }
  const double tot = block_sum(el, scratch);
  if (threadIdx.x == 0) a.elbo_part[blockIdx.x] = tot;
}

// Input-side part of the window scatter (layers.py:1446-1448): accumulate into the gradient of the layer above.
struct UpArgs {
  WindowSpec w; int QP; int64_t up_rows;
  const int64_t* up_pre;
  const double *gmu, *gS;
  double *g_up_mean, *g_up_var;
};

__global__ void __launch_bounds__(256) up_latent_kernel(UpArgs a) {
  const int Ud = a.w.U_dim, Uw = a.w.U_win, Qx = a.w.X_win * a.w.X_dim;
  const int64_t total = a.up_rows * Ud;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / Ud;
    const int d = (int)(i % Ud);
    const int s = find_seq(a.up_pre, a.w.n_seq, row);
    const int64_t tt = row - a.up_pre[s];
    const int64_t Ns = a.w.own_len[s] - a.w.X_win;
    const int64_t uoff = a.w.up_len[s] - Ns - Uw + 1;
    double gm = 0.0, gv = 0.0;
    for (int j = 0; j < Uw; ++j) {
      const int64_t n = tt - uoff - j;
      if (n >= 0 && n < Ns) {
        const int64_t r = a.w.row_off[s] + n;
        gm += a.gmu[r * a.QP + Qx + j * Ud + d];
        gv += a.gS[r * a.QP + Qx + j * Ud + d];
      }
    }
    const int64_t gi = a.w.up_off[s] + tt * Ud + d;
    a.g_up_mean[gi] += gm;
    a.g_up_var[gi] += gv;
  }
}

// w scales the terms that are replicated on every rank of a data-parallel collapsed bound (1 otherwise)
__global__ void finish_elbo_kernel(const double* bound, const double* part, int npart, const double* dthetaL,
                                   double* elbo, double* gnoise, double w) {
  // one wave: lane l adds partials l, l+64, ... in order, then a fixed-order wave sum (deterministic; a single thread walking
  // the 1024 partials took 48 us of dependent loads -- on the side stream's chain of every layer step)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < npart; i += 64) s += part[i];
  s = wave_sum_dpp(s);
  if (threadIdx.x == 0) {
    elbo[0] = w * bound[0] + s;
    if (gnoise) gnoise[0] = w * dthetaL[0];
  }
}

__global__ void unpad_kernel(int64_t N, int Q, int QP, const double* __restrict__ in, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * Q) out[i] = in[(i / Q) * QP + (i % Q)];
}

__global__ void scale_psi_kernel(int64_t n, const double* variance, int power, double* x) {
  // the row constants carry log(variance) already; nothing to do -- kept for ABI symmetry
}

// ==================================================================================================================
// workspace carving shared by the entry points
// ==================================================================================================================
struct LayerWs {
  ZTerms zt; RowTerms rt;
  double *psi1, *Lp, *gmu, *gS, *Ym, *Yv, *dYm;
  double *p2_part, *p2_tri, *psi2, *dLs;
  double *dz1_part, *dz1, *dz2_part, *dz2, *dlam1_part, *dlam1, *dlam2_part, *dlam2, *lsum1_part, *lsum1, *wsum2_part, *wsum2;
  double *yry_part, *yry, *psi1Y_part, *psi1Y, *psi0sum, *psi0coef, *elbo_part, *fin_dlrow, *fin_gsum;
  double *r2, *dYvar_row, *PsiR_part, *PsiR;      // collapsed bound only
  double *pair_ws;                                // pair-form psi2 pass (nullptr when the handle's mode does not select it)
  int64_t *row_off, *own_pre, *up_pre;
  int nblk1, nblk2;
};

#define NBLK1 512          // workgroups of the psi1-side reductions
#define LAT_BLOCKS 1024

static size_t carve_layer(rgp_handle h, char* base, int64_t N, int M, int Q, int D, int n_seq, LayerWs& L, SviChain* ch,
                          VarDtcChain* vc = nullptr) {
  WsAlloc ws(base);
  const int QP = rgp_qp(Q);
  const size_t nq = (size_t)N * QP, nm = (size_t)N * M, nd = (size_t)N * (D > 0 ? D : 1);
  const int tri = M * (M + 1) / 2;
  L.nblk1 = NBLK1;
  L.nblk2 = (int)psi2_num_blocks(h, N);
  L.zt.M = M; L.zt.Q = Q; L.zt.QP = QP;
  L.zt.c = ws.take<double>(QP); L.zt.lam = ws.take<double>(QP); L.zt.Zc = ws.take<double>((size_t)M * QP); L.zt.ZcT = ws.take<double>((size_t)QP * RGP_PSI2_TEAM);
  L.zt.zd = ws.take<double>((size_t)M * M); L.zt.Kuu = ws.take<double>((size_t)M * M);
  if (ch) svi_chain_carve(*ch, ws);        // before anything sized by N: a chain-only prepare fixes these addresses
  // dLs is written by the chain's second kernel (chain-only prepare) and read by the psi2 pass and the finalisation of a
  // layer step that may re-carve the slot with a smaller N: it must sit in front of every N-sized buffer as well
  L.psi2 = ws.take<double>((size_t)M * M); L.dLs = ws.take<double>((size_t)M * M);
  L.rt.N = N; L.rt.QP = QP;
  L.rt.Xmu = ws.take<double>(nq); L.rt.d2 = ws.take<double>(nq); L.rt.e1 = ws.take<double>(nq); L.rt.em = ws.take<double>(nq);
  L.rt.g0 = ws.take<double>(N); L.rt.g1 = ws.take<double>(N);
  L.psi1 = ws.take<double>(nm); L.Lp = ws.take<double>(nm);
  L.gmu = ws.take<double>(nq); L.gS = ws.take<double>(nq);
  L.Ym = ws.take<double>(nd); L.Yv = ws.take<double>(nd); L.dYm = ws.take<double>(nd);
  L.p2_part = ws.take<double>((size_t)L.nblk2 * tri); L.p2_tri = ws.take<double>(tri);
  L.dz1_part = ws.take<double>((size_t)L.nblk1 * M * QP); L.dz1 = ws.take<double>((size_t)M * QP);
  L.dz2_part = ws.take<double>((size_t)2 * L.nblk2 * QP * RGP_PSI2_TEAM); L.dz2 = ws.take<double>((size_t)QP * RGP_PSI2_TEAM);
  L.dlam1_part = ws.take<double>((size_t)PSI1_ROWS_MAXBLK * QP); L.dlam1 = ws.take<double>(QP);     // PSI1_ROWS_MAXBLK >= NBLK1
  L.dlam2_part = ws.take<double>((size_t)4 * L.nblk2 * QP); L.dlam2 = ws.take<double>(QP);      // up to 8 teams x num_cu
  L.lsum1_part = ws.take<double>(PSI1_ROWS_MAXBLK); L.lsum1 = ws.take<double>(1);
  L.wsum2_part = ws.take<double>((size_t)4 * L.nblk2); L.wsum2 = ws.take<double>(1);
  L.yry_part = ws.take<double>(2048); L.yry = ws.take<double>(1);
  L.psi1Y_part = ws.take<double>((size_t)L.nblk1 * (D > 0 ? D : 1) * M); L.psi1Y = ws.take<double>((size_t)(D > 0 ? D : 1) * M);
  L.psi0sum = ws.take<double>(1); L.psi0coef = ws.take<double>(1); L.elbo_part = ws.take<double>(LAT_BLOCKS);
  L.fin_dlrow = ws.take<double>((size_t)M * QP); L.fin_gsum = ws.take<double>(M);
  L.row_off = ws.take<int64_t>(n_seq + 1); L.own_pre = ws.take<int64_t>(n_seq + 1); L.up_pre = ws.take<int64_t>(n_seq + 1);
  if (vc) {
    L.r2 = ws.take<double>(N); L.dYvar_row = ws.take<double>(N);
    L.PsiR_part = ws.take<double>((size_t)L.nblk1 * M * M); L.PsiR = ws.take<double>((size_t)M * M);
    vardtc_chain_carve(*vc, ws);
  }
  {                                               // last: nothing in front of it moves when the mode changes
    PairGeom pg;
    L.pair_ws = (!vc && psi2_pair_selected(h, N, M, QP, pg)) ? ws.take<double>(psi2_pair_ws_doubles(pg, N)) : nullptr;
  }
  return ws.off;
}

static int64_t big_workspace_bytes(int64_t N, int M, int Q, int D);
extern "C" int64_t rgp_workspace_bytes(int64_t N, int M, int Q, int D) {
  if (M > RGP_MAX_M || rgp_qp(Q) > RGP_MAX_QP) return big_workspace_bytes(N, M, Q, D);
  rgp_ctx tmp;
  LayerWs L;
  SviChain ch{}; ch.M = M; ch.D = D > 0 ? D : 1;
  return (int64_t)carve_layer(&tmp, nullptr, N, M, Q, D, 4096, L, &ch);
}

static inline bool rgp_is_big(int M) { return M > RGP_MAX_M; }
// windows wider than 64 columns: the GEMM-shaped path of rgp_wideq_impl.h
static inline bool rgp_is_wide(const rgp_layer_in* in) { return rgp_qp(in->X_win * in->X_dim + in->U_win * in->U_dim) > RGP_MAX_QP; }

static rgp_status check_geometry(rgp_handle h, int64_t N, int M, int Q) {
  if (N <= 0 || M <= 0 || Q <= 0) return rgp_fail(h, RGP_ERR_BAD_ARG, "N, M, Q must be positive");
  if (M > RGP_MAX_M) return rgp_fail(h, RGP_ERR_BAD_ARG, "M > 128 is not supported by this entry point");
  if (rgp_qp(Q) > RGP_MAX_QP) return rgp_fail(h, RGP_ERR_BAD_ARG, "Q > 64 is not supported by this build");
  return RGP_OK;
}

static rgp_status last_launch(rgp_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return RGP_OK;
}

#include "rgp_big_impl.h"

// ==================================================================================================================
// unit entry points
// ==================================================================================================================
extern "C" rgp_status rgp_kuu(rgp_handle h, int M, int Q, const double* Z, const double* variance, const double* inv_l,
                              double jitter, double* Kuu) {
  if (!h) return RGP_ERR_BAD_ARG;
  rgp_status st = (rgp_is_big(M) || rgp_qp(Q) > RGP_MAX_QP) ? big_check(h, 1, M, Q, true) : check_geometry(h, 1, M, Q);
  if (st) return st;
  LayerWs L;
  const size_t need = carve_layer(h, nullptr, 1, M, Q, 1, 1, L, nullptr);
  if ((st = rgp_ensure_ws(h, need))) return st;
  carve_layer(h, h->ws, 1, M, Q, 1, 1, L, nullptr);
  launch_prep_z(h->stream, M, Q, L.zt.QP, Z, variance, inv_l, jitter, L.zt);
  RGP_HIP_CHECK(h, hipMemcpyAsync(Kuu, L.zt.Kuu, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return last_launch(h, "rgp_kuu");
}

extern "C" rgp_status rgp_matmul(rgp_handle h, int m, int n, int k, double alpha, const double* A, int lda, int ta,
                                 const double* B, int ldb, int tb, double gamma, double* C, int ldc, int splits) {
  if (!h) return RGP_ERR_BAD_ARG;
  if (m < 1 || n < 1 || k < 0 || !A || !B || !C) return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_matmul: bad shape or null operand");
  if (lda < (ta ? m : k) || ldb < (tb ? k : n) || ldc < n) return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_matmul: leading dimension too small");
  if (splits > 1 && gamma != 0.0) return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_matmul: a split product cannot accumulate into C");
  if (splits > 64) return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_matmul: at most 64 splits");
  GemmArgs g;
  g.m = m; g.n = n; g.k = k; g.A = A; g.lda = lda; g.ta = ta != 0; g.B = B; g.ldb = ldb; g.tb = tb != 0; g.alpha = alpha;
  if (splits > 1) {
    const size_t mn = (size_t)m * n;
    rgp_status st = rgp_ensure_ws(h, (size_t)splits * mn * sizeof(double) + 256);
    if (st) return st;
    double* part = reinterpret_cast<double*>(h->ws);
    g.C = part; g.ldc = n; g.splits = splits; g.split_stride = (int64_t)mn;
    launch_gemm(h->stream, g);
    if (ldc == n) launch_reduce(h->stream, splits, (int64_t)mn, (int64_t)mn, part, C, 1.0, 0);
    else {
      launch_reduce(h->stream, splits, (int64_t)mn, (int64_t)mn, part, part, 1.0, 0);       // element i reads only part[g * mn + i]
      RGP_HIP_CHECK(h, hipMemcpy2DAsync(C, (size_t)ldc * sizeof(double), part, (size_t)n * sizeof(double), (size_t)n * sizeof(double),
                                        (size_t)m, hipMemcpyDeviceToDevice, h->stream));
    }
  } else {
    g.C = C; g.ldc = ldc; g.gamma = gamma;
    if (gamma != 0.0) { g.Cin = C; g.ldcin = ldc; }
    launch_gemm(h->stream, g);
  }
  return last_launch(h, "rgp_matmul");
}

extern "C" rgp_status rgp_psi_forward(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S,
                                      const double* Z, const double* variance, const double* inv_l, double* psi1,
                                      double* psi2) {
  if (!h) return RGP_ERR_BAD_ARG;
  if (rgp_is_big(M) || rgp_qp(Q) > RGP_MAX_QP) return big_psi_forward(h, N, M, Q, mu, S, Z, variance, inv_l, psi1, psi2);
  rgp_status st = check_geometry(h, N, M, Q);
  if (st) return st;
  LayerWs L;
  const size_t need = carve_layer(h, nullptr, N, M, Q, 1, 1, L, nullptr);
  if ((st = rgp_ensure_ws(h, need))) return st;
  carve_layer(h, h->ws, N, M, Q, 1, 1, L, nullptr);
  hipStream_t s = h->stream;
  launch_prep_z(s, M, Q, L.zt.QP, Z, variance, inv_l, RGP_JITTER, L.zt);
  launch_rowprep_dense(s, N, Q, mu, S, variance, L.zt, L.rt);
  if (psi1) launch_psi1_fwd(s, N, M, L.zt, L.rt, psi1);
  if (psi2) {
    Psi2Out o{};
    o.p2_part = L.p2_part;
    if ((st = launch_psi2(h, true, false, N, M, Q, L.zt, L.rt, nullptr, o, L.nblk2))) return st;
    const int tri = M * (M + 1) / 2;
    launch_reduce(s, L.nblk2, tri, tri, L.p2_part, L.p2_tri, 1.0, 0);
    launch_unpack_tri(s, M, L.p2_tri, psi2);
  }
  return last_launch(h, "rgp_psi_forward");
}

// rgp_psi_forward without any allocation: the Z terms, the row terms and the per-workgroup partial sums of psi2 live in the
// static device scratch.  Same kernels as rgp_psi_forward; the psi2 partials are summed over at most RGP_STATIC_NBLK
// workgroups, so the two entry points agree to rounding (fixed order each), not bit for bit.
extern "C" rgp_status rgp_psi_forward_static(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S,
                                             const double* Z, const double* variance, const double* inv_l, double* psi1,
                                             double* psi2) {
  if (!h) return RGP_ERR_BAD_ARG;
  rgp_status st = check_geometry(h, N, M, Q);
  if (st) return st;
  if (N > RGP_STATIC_MAX_N)
    return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_psi_forward_static: N > 4096 exceeds the static scratch (use rgp_psi_forward)");
  const int QP = rgp_qp(Q);
  const int tri = M * (M + 1) / 2;
  int64_t nb = 2 * (int64_t)h->num_cu;
  if (nb > RGP_STATIC_NBLK) nb = RGP_STATIC_NBLK;
  if (nb > (N + 1) / 2) nb = (N + 1) / 2;
  const int nblk = (int)(nb < 1 ? 1 : nb);
  ZTerms zt{};
  RowTerms rt{};
  WsAlloc ws(h->static_scratch);
  const size_t nq = (size_t)N * QP;
  zt.M = M; zt.Q = Q; zt.QP = QP;
  zt.c = ws.take<double>(QP); zt.lam = ws.take<double>(QP); zt.Zc = ws.take<double>((size_t)M * QP);
  zt.ZcT = ws.take<double>((size_t)QP * RGP_PSI2_TEAM); zt.zd = ws.take<double>((size_t)M * M); zt.Kuu = ws.take<double>((size_t)M * M);
  rt.N = N; rt.QP = QP;
  rt.Xmu = ws.take<double>(nq); rt.d2 = ws.take<double>(nq); rt.e1 = ws.take<double>(nq); rt.em = ws.take<double>(nq);
  rt.g0 = ws.take<double>(N); rt.g1 = ws.take<double>(N);
  double* p2_part = ws.take<double>((size_t)nblk * tri);
  double* p2_tri = ws.take<double>(tri);
  if (ws.off > RGP_STATIC_SCRATCH_BYTES)
    return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_psi_forward_static: the problem does not fit the static scratch");
  rgp_join_streams(h);
  hipStream_t s = h->stream;
  launch_prep_z(s, M, Q, QP, Z, variance, inv_l, RGP_JITTER, zt);
  launch_rowprep_dense(s, N, Q, mu, S, variance, zt, rt);
  if (psi1) launch_psi1_fwd(s, N, M, zt, rt, psi1);
  if (psi2) {
    Psi2Out o{};
    o.p2_part = p2_part;
    if ((st = launch_psi2(h, true, false, N, M, Q, zt, rt, nullptr, o, nblk))) return st;
    launch_reduce(s, nblk, tri, tri, p2_part, p2_tri, 1.0, 0);
    launch_unpack_tri(s, M, p2_tri, psi2);
  }
  return last_launch(h, "rgp_psi_forward_static");
}

extern "C" rgp_status rgp_psi_backward(rgp_handle h, int64_t N, int M, int Q, const double* mu, const double* S,
                                       const double* Z, const double* variance, const double* inv_l,
                                       double dL_dpsi0_sum, const double* dL_dpsi1, const double* dL_dpsi2, double* dvar,
                                       double* dinv_l, double* dZ, double* dmu, double* dS) {
  if (!h) return RGP_ERR_BAD_ARG;
  if (rgp_is_big(M) || rgp_qp(Q) > RGP_MAX_QP)
    return big_psi_backward(h, N, M, Q, mu, S, Z, variance, inv_l, dL_dpsi0_sum, dL_dpsi1, dL_dpsi2, dvar, dinv_l, dZ, dmu, dS);
  rgp_status st = check_geometry(h, N, M, Q);
  if (st) return st;
  LayerWs L;
  const size_t need = carve_layer(h, nullptr, N, M, Q, 1, 1, L, nullptr);
  if ((st = rgp_ensure_ws(h, need))) return st;
  carve_layer(h, h->ws, N, M, Q, 1, 1, L, nullptr);
  hipStream_t s = h->stream;
  const int QP = L.zt.QP;
  launch_prep_z(s, M, Q, QP, Z, variance, inv_l, RGP_JITTER, L.zt);
  launch_rowprep_dense(s, N, Q, mu, S, variance, L.zt, L.rt);
  if (dL_dpsi1) {
    launch_psi1_fwd(s, N, M, L.zt, L.rt, L.psi1);
    launch_psi1_bwd(s, N, M, Q, L.zt, L.rt, dL_dpsi1, L.psi1, L.Lp, L.gmu, L.gS, L.dlam1_part, L.lsum1_part, L.nblk1);
    launch_reduce(s, L.nblk1, QP, QP, L.dlam1_part, L.dlam1, 1.0, 0);
    launch_reduce(s, L.nblk1, 1, 1, L.lsum1_part, L.lsum1, 1.0, 0);
    launch_psi1_dz(s, N, M, L.zt, L.rt, L.Lp, L.dz1_part, L.nblk1);
    launch_reduce(s, L.nblk1, (int64_t)M * QP, (int64_t)M * QP, L.dz1_part, L.dz1, 1.0, 0);
  } else {
    RGP_HIP_CHECK(h, hipMemsetAsync(L.gmu, 0, (size_t)N * QP * sizeof(double), s));
    RGP_HIP_CHECK(h, hipMemsetAsync(L.gS, 0, (size_t)N * QP * sizeof(double), s));
  }
  if (dL_dpsi2) {
    hipLaunchKernelGGL(symmetrise_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, dL_dpsi2, L.dLs);
    Psi2Out o{};
    o.p2_part = L.p2_part; o.dz_part = L.dz2_part; o.dlam_part = L.dlam2_part; o.wsum_part = L.wsum2_part;
    o.gmu = L.gmu; o.gS = L.gS; o.accumulate = 1; o.pair_ws = L.pair_ws;
    const Psi2Plan pl = psi2_plan(h, N, M, QP, true, true);
    if ((st = launch_psi2(h, true, true, N, M, Q, L.zt, L.rt, L.dLs, o, pl.grid))) return st;
    const int tri = M * (M + 1) / 2;
    launch_reduce(s, pl.n_p2, tri, tri, L.p2_part, L.p2_tri, 1.0, 0);
    launch_unpack_tri(s, M, L.p2_tri, L.psi2);
    launch_reduce(s, pl.n_dz, (int64_t)QP * RGP_PSI2_TEAM, (int64_t)QP * RGP_PSI2_TEAM, L.dz2_part, L.dz2, 1.0, 0);
    launch_reduce(s, pl.n_dl, QP, QP, L.dlam2_part, L.dlam2, 1.0, 0);
    launch_reduce(s, pl.n_dl, 1, 1, L.wsum2_part, L.wsum2, 1.0, 0);
  }
  FinalizeArgs f{};
  f.M = M; f.Q = Q; f.QP = QP; f.Zc = L.zt.Zc; f.lam = L.zt.lam; f.variance = variance;
  f.Kj = nullptr; f.dKmm = nullptr;
  f.dLs = dL_dpsi2 ? L.dLs : nullptr; f.psi2 = dL_dpsi2 ? L.psi2 : nullptr;
  f.dz1 = dL_dpsi1 ? L.dz1 : nullptr; f.dz2 = dL_dpsi2 ? L.dz2 : nullptr;
  f.dlam1 = dL_dpsi1 ? L.dlam1 : nullptr; f.dlam2 = dL_dpsi2 ? L.dlam2 : nullptr;
  f.lsum1 = dL_dpsi1 ? L.lsum1 : nullptr; f.wsum2 = dL_dpsi2 ? L.wsum2 : nullptr;
  f.dk_scale = 1.0; f.psi0_coef = dL_dpsi0_sum; f.psi0_dev = nullptr; f.psi0_dev_scale = 0.0;
  f.gZ = dZ; f.ginv_l = dinv_l; f.gvariance = dvar;
  launch_finalize(s, f, L.fin_dlrow, L.fin_gsum);
  const int64_t nq = N * Q;
  hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, N, Q, QP, L.gmu, dmu);
  hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, N, Q, QP, L.gS, dS);
  return last_launch(h, "rgp_psi_backward");
}

extern "C" rgp_status rgp_svi_chain(rgp_handle h, int64_t N, int M, int D, double qU_ratio, const double* Kuu_jit,
                                    const double* noise, const double* qU_m, const double* qU_W, const double* qU_a,
                                    const double* psi0sum, const double* psi1Y, const double* psi2, const double* yry,
                                    double* logL_minus_KL, double* dL_dKmm, double* dL_dpsi2, double* Kinv_mu,
                                    double* dL_dthetaL, double* gqU_m, double* gqU_W, double* gqU_a, double* woodbury_inv) {
  if (!h) return RGP_ERR_BAD_ARG;
  rgp_status st;
  SviChain c{};
  c.N = N; c.M = M; c.D = D; c.qU_ratio = qU_ratio;
  const size_t need = svi_chain_ws_doubles(M, D) * sizeof(double) + 64 * 256;
  if ((st = rgp_ensure_ws(h, need))) return st;
  WsAlloc ws(h->ws);
  svi_chain_carve(c, ws);
  c.Kj = Kuu_jit; c.noise = noise; c.qU_m = qU_m; c.qU_W = qU_W; c.qU_a = qU_a;
  c.psi0sum = psi0sum; c.psi1Y = psi1Y; c.psi2 = psi2; c.yry = yry;
  c.gqU_m = gqU_m; c.gqU_W = gqU_W; c.gqU_a = gqU_a; c.winv = woodbury_inv; c.status = h->status;
  if ((st = launch_svi_chain_pre(h, c, h->stream))) return st;
  if ((st = launch_svi_chain_post(h, c, h->stream))) return st;
  hipStream_t s = h->stream;
  const size_t mm = (size_t)M * M * sizeof(double), md = (size_t)M * D * sizeof(double);
  if (logL_minus_KL) RGP_HIP_CHECK(h, hipMemcpyAsync(logL_minus_KL, c.bound, sizeof(double), hipMemcpyDeviceToDevice, s));
  if (dL_dKmm) RGP_HIP_CHECK(h, hipMemcpyAsync(dL_dKmm, c.dKmm, mm, hipMemcpyDeviceToDevice, s));
  if (dL_dpsi2) RGP_HIP_CHECK(h, hipMemcpyAsync(dL_dpsi2, c.dLdpsi2, mm, hipMemcpyDeviceToDevice, s));
  if (Kinv_mu) RGP_HIP_CHECK(h, hipMemcpyAsync(Kinv_mu, c.Kinv_mu, md, hipMemcpyDeviceToDevice, s));
  if (dL_dthetaL) RGP_HIP_CHECK(h, hipMemcpyAsync(dL_dthetaL, c.dthetaL, sizeof(double), hipMemcpyDeviceToDevice, s));
  return last_launch(h, "rgp_svi_chain");
}

// ==================================================================================================================
// collapsed-bound layer step (VarDTC.inference, vardtc.py:88-208): dL/dpsi2 depends on the statistics, so the psi2
// kernel runs twice (forward sum, then backward with recomputation) around the M x M chain.
// ==================================================================================================================
// phase 0: whole step with local statistics; 1: statistics only (packed into stats_io); 2: finish with the statistics in
// stats_io (e.g. after an all-reduce over ranks), N_global rows in total, replicated terms weighted by rep_w.
// Packed layout: [psi0sum, yry, psi2 (M*M), psi1Y (D*M), PsiR (M*M)].
static rgp_status vardtc_layer_step(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out, int phase = 0,
                                    double* stats_io = nullptr, int64_t N_global = 0, double rep_w = 1.0) {
  const int M = in->M, D = in->D;
  const int Q = in->X_win * in->X_dim + in->U_win * in->U_dim;
  const int64_t N = in->N;
  if (rgp_is_big(M) || rgp_is_wide(in)) return big_vardtc_layer_step(h, in, out, phase, stats_io, N_global, rep_w);
  rgp_status st = check_geometry(h, N, M, Q);
  if (st) return st;
  if (in->y_uncertain && (in->X_dim != D || !in->own_var)) return rgp_fail(h, RGP_ERR_BAD_ARG, "hidden layer needs X_dim == D and own_var");
  if (!in->y_uncertain && in->X_win != 0) return rgp_fail(h, RGP_ERR_BAD_ARG, "observed layer must have X_win == 0 (as in the reference)");
  LayerWs L;
  VarDtcChain c{};
  c.N = N; c.M = M; c.D = D; c.y_uncertain = in->y_uncertain;
  const size_t need = carve_layer(h, nullptr, N, M, Q, D, in->n_seq, L, nullptr, &c);
  // The layers of a collapsed-bound evaluation are independent of each other up to the final overlap-add into the latent
  // gradients, and each is a chain of ~35 small kernels: consecutive layer steps rotate over the handle's three streams
  // (own workspace slot each) so that they run side by side; only the latent-gradient kernels at the end wait for the
  // previous layer step.  rgp_join / rgp_sync_status collect them (ev_done of the slots).
  const bool rotate = h->overlap && phase == 0;
  int slot = -1;
  hipStream_t run = h->stream;
  if (rotate) {
    slot = h->slot_next;
    h->slot_next = (slot + 1) % rgp_ctx::NSLOT;
    if ((st = rgp_ensure_ws_slot(h, slot, need))) return st;
    carve_layer(h, h->ws_slot[slot], N, M, Q, D, in->n_seq, L, nullptr, &c);
    hipStream_t ring[3] = {h->stream, h->side_pre, h->side_post};
    run = ring[h->vardtc_rr++ % 3];
    if (run != h->stream) {                          // inputs were written by earlier main-stream work
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_main, h->stream));
      RGP_HIP_CHECK(h, hipStreamWaitEvent(run, h->ev_main, 0));
    }
    if (h->done_recorded[slot]) RGP_HIP_CHECK(h, hipStreamWaitEvent(run, h->ev_done[slot], 0));
  } else {
    if ((st = rgp_ensure_ws(h, need))) return st;
    carve_layer(h, h->ws, N, M, Q, D, in->n_seq, L, nullptr, &c);
  }
  struct StreamSwap {                                // the launch helpers below take the stream from the handle
    rgp_handle h; hipStream_t saved;
    StreamSwap(rgp_handle h_, hipStream_t s_) : h(h_), saved(h_->stream) { h->stream = s_; }
    ~StreamSwap() { h->stream = saved; }
  } swap_guard(h, run);
  hipStream_t s = run;
  const int QP = L.zt.QP;
  const int tri = M * (M + 1) / 2;

  WindowSpec w{};
  w.n_seq = in->n_seq; w.X_win = in->X_win; w.X_dim = in->X_dim; w.U_win = in->U_win; w.U_dim = in->U_dim; w.D = D;
  w.y_uncertain = in->y_uncertain; w.own_off = in->own_off; w.own_len = in->own_len; w.up_off = in->up_off;
  w.up_len = in->up_len; w.row_off = L.row_off; w.own_mean = in->own_mean; w.own_var = in->own_var;
  w.up_mean = in->up_mean; w.up_var = in->up_var; w.up_var_const = in->up_var_const;
  const size_t mm = (size_t)M * M, dm_ = (size_t)D * M;
  if (phase != 2) {
  // row offsets, prefix sums and psi0 in one launch (psi0coef is overwritten after the chain, with the clipped noise)
  hipLaunchKernelGGL(svi_tables_kernel, dim3(1), dim3(64), 0, s, in->n_seq, in->X_win, in->own_len,
                     in->U_win > 0 ? in->up_len : nullptr, L.row_off, L.own_pre, L.up_pre, N, D, in->variance, in->noise,
                     L.psi0sum, L.psi0coef);
  launch_prep_z(s, M, Q, QP, in->Z, in->variance, in->inv_l, RGP_JITTER, L.zt);
  launch_rowprep_window(s, N, Q, w, in->variance, L.zt, L.rt, L.Ym, in->y_uncertain ? L.Yv : nullptr, L.yry_part);
  launch_reduce(s, 2048, 1, 1, L.yry_part, L.yry, 1.0, 0);

  // statistics (vardtc.py:54-86)
  launch_psi1_fwd(s, N, M, L.zt, L.rt, L.psi1);
  if (D > ((256 / M) > 0 ? (256 / M) : 1) * CM_MAXK * 64) return rgp_fail(h, RGP_ERR_BAD_ARG, "D too large");
  launch_colmm(s, N, D, M, L.Ym, L.psi1, nullptr, L.psi1Y_part, L.psi1Y, L.nblk1);
  if (in->y_uncertain) {
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, D, L.Yv, L.r2);
    launch_colmm(s, N, M, M, L.psi1, L.psi1, L.r2, L.PsiR_part, L.PsiR, L.nblk1);
  }
  {
    Psi2Out o{};
    o.p2_part = L.p2_part;
    if ((st = launch_psi2(h, true, false, N, M, Q, L.zt, L.rt, nullptr, o, L.nblk2))) return st;
    launch_reduce(s, L.nblk2, tri, tri, L.p2_part, L.p2_tri, 1.0, 0);
    launch_unpack_tri(s, M, L.p2_tri, L.psi2);
  }
  }
  if (phase == 1) {                                  // hand the local statistics to the caller (vardtc.py:65-67 stub)
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io, L.psi0sum, sizeof(double), hipMemcpyDeviceToDevice, s));
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 1, L.yry, sizeof(double), hipMemcpyDeviceToDevice, s));
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 2, L.psi2, mm * sizeof(double), hipMemcpyDeviceToDevice, s));
    RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 2 + mm, L.psi1Y, dm_ * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (in->y_uncertain) RGP_HIP_CHECK(h, hipMemcpyAsync(stats_io + 2 + mm + dm_, L.PsiR, mm * sizeof(double), hipMemcpyDeviceToDevice, s));
    else RGP_HIP_CHECK(h, hipMemsetAsync(stats_io + 2 + mm + dm_, 0, mm * sizeof(double), s));
    return last_launch(h, "rgp_vardtc_stats");
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

  // row gradients
  {
    const size_t lds = ((size_t)M * (D + 1) + (in->y_uncertain ? (size_t)M * (M + 1) : 0)) * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)vardtc_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, "vardtc rows LDS attr");
    int64_t b = (N + 3) / 4; if (b > 1024) b = 1024;
    hipLaunchKernelGGL(vardtc_rows_kernel, dim3((unsigned)b), dim3(256), lds, s, N, M, D, in->y_uncertain, L.Ym, L.r2, L.psi1,
                       c.vT, c.G2, c.beta, L.Lp, L.dYm, L.dYvar_row);
  }
  launch_psi1_bwd(s, N, M, Q, L.zt, L.rt, L.Lp, L.psi1, L.Lp, L.gmu, L.gS, L.dlam1_part, L.lsum1_part, L.nblk1);
  launch_psi1_dz(s, N, M, L.zt, L.rt, L.Lp, L.dz1_part, L.nblk1);
  hipLaunchKernelGGL(symmetrise_kernel, dim3((M * M + 255) / 256), dim3(256), 0, s, M, c.dLdpsi2, L.dLs);
  {
    Psi2Out o{};
    o.dz_part = L.dz2_part; o.dlam_part = L.dlam2_part; o.wsum_part = L.wsum2_part;
    o.gmu = L.gmu; o.gS = L.gS; o.accumulate = 1;
    const Psi2Plan pl = psi2_plan(h, N, M, QP, false, true);
    if ((st = launch_psi2(h, false, true, N, M, Q, L.zt, L.rt, L.dLs, o, pl.grid))) return st;
    // the partial sums of the three backward passes in one launch (nothing reads them before the finalisation; the small
    // problems of the collapsed bound are bound by the number of launches, ~5 us each)
    const ReduceSeg segs[6] = {{L.dlam1_part, L.dlam1, L.nblk1, QP, QP},
                               {L.lsum1_part, L.lsum1, L.nblk1, 1, 1},
                               {L.dz1_part, L.dz1, L.nblk1, (int64_t)M * QP, (int64_t)M * QP},
                               {L.dz2_part, L.dz2, pl.n_dz, (int64_t)QP * RGP_PSI2_TEAM, (int64_t)QP * RGP_PSI2_TEAM},
                               {L.dlam2_part, L.dlam2, pl.n_dl, QP, QP},
                               {L.wsum2_part, L.wsum2, pl.n_dl, 1, 1}};
    launch_reduce_batch(s, 6, segs);
  }
  {
    FinalizeArgs f{};
    f.M = M; f.Q = Q; f.QP = QP; f.Zc = L.zt.Zc; f.lam = L.zt.lam; f.variance = in->variance;
    f.Kj = L.zt.Kuu; f.dKmm = c.dKmm; f.dLs = L.dLs; f.psi2 = L.psi2; f.dz1 = L.dz1; f.dz2 = L.dz2;
    f.dlam1 = L.dlam1; f.dlam2 = L.dlam2; f.lsum1 = L.lsum1; f.wsum2 = L.wsum2;
    f.dk_scale = rep_w; f.psi0_coef = 0.0; f.psi0_dev = L.psi0coef; f.psi0_dev_scale = 1.0;
    f.gZ = out->gZ; f.ginv_l = out->ginv_l; f.gvariance = out->gvariance;
    launch_finalize(s, f, L.fin_dlrow, L.fin_gsum);
  }
  int npart = 0;
  // the overlap-add into the latent gradients follows the layer above (its own_latent assigns what this layer adds to)
  if (rotate && h->vardtc_prev_slot >= 0 && h->done_recorded[h->vardtc_prev_slot])
    RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_done[h->vardtc_prev_slot], 0));
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
  hipLaunchKernelGGL(finish_elbo_kernel, dim3(1), dim3(64), 0, s, c.bound, L.elbo_part, npart, c.dthetaL, out->elbo,
                     out->gnoise, rep_w);
  if (out->woodbury_vector)
    RGP_HIP_CHECK(h, hipMemcpyAsync(out->woodbury_vector, c.vT, (size_t)M * D * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (rotate) {
    RGP_HIP_CHECK(h, hipEventRecord(h->ev_done[slot], s));
    h->done_recorded[slot] = true;
    h->vardtc_prev_slot = slot;
  }
  return last_launch(h, "rgp_layer_step(vardtc)");
}

// ==================================================================================================================
// the fused layer step
// ==================================================================================================================
// Takes the next workspace slot, computes the Z terms on the main stream and starts the data-independent part of the
// SVI chain on the side stream.  ev_pre[slot] fires when Cpsi1 / dL_dpsi2 / the factors are ready.
static WindowSpec svi_window_spec(const rgp_layer_in* in, const LayerWs& L) {
  WindowSpec w{};
  w.n_seq = in->n_seq; w.X_win = in->X_win; w.X_dim = in->X_dim; w.U_win = in->U_win; w.U_dim = in->U_dim; w.D = in->D;
  w.y_uncertain = in->y_uncertain; w.own_off = in->own_off; w.own_len = in->own_len; w.up_off = in->up_off;
  w.up_len = in->up_len; w.row_off = L.row_off; w.own_mean = in->own_mean; w.own_var = in->own_var;
  w.up_mean = in->up_mean; w.up_var = in->up_var; w.up_var_const = in->up_var_const;
  return w;
}

static rgp_status svi_check_args(rgp_handle h, const rgp_layer_in* in) {
  const int Q = in->X_win * in->X_dim + in->U_win * in->U_dim;
  rgp_status st = check_geometry(h, in->N, in->M, Q);
  if (st) return st;
  if (in->y_uncertain && (in->X_dim != in->D || !in->own_var)) return rgp_fail(h, RGP_ERR_BAD_ARG, "hidden layer needs X_dim == D and own_var");
  if (!in->y_uncertain && in->X_win != 0) return rgp_fail(h, RGP_ERR_BAD_ARG, "observed layer must have X_win == 0 (as in the reference)");
  if (in->D > 64 * CM_MAXK) return rgp_fail(h, RGP_ERR_BAD_ARG, "D too large");
  return RGP_OK;
}

// Everything of a layer step that depends only on the parameters and the signals, not on the M x M chain: K_uu and the
// chain's first stage (side stream), the window tables and the row terms (main stream).  rgp_layer_prepare issues this for
// every layer before the first step, so the main stream has row work to do while the first chain runs.
// mode: 0 = chain + rows (new slot), 1 = chain only (new slot; ordered after the side_post stream, NOT after the main
// stream: used right after a layer's parameter update while the main stream is still busy with later layers),
// 2 = rows only (attaches to `*slot_out`, a slot prepared with mode 1).
static rgp_status svi_prepare_slot(rgp_handle h, const rgp_layer_in* in, int mode, int* slot_out) {
  const int M = in->M, D = in->D;
  const int Q = in->X_win * in->X_dim + in->U_win * in->U_dim;
  const int64_t N = in->N;
  rgp_status st;
  LayerWs L;
  SviChain c{};
  c.N = N; c.M = M; c.D = D; c.qU_ratio = in->qU_ratio;
  const size_t need = carve_layer(h, nullptr, N, M, Q, D, in->n_seq, L, &c);
  int slot;
  if (mode == 2) {
    slot = *slot_out;
    if (need > h->ws_slot_bytes[slot])
      return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_layer_prepare(rows): the minibatch is larger than the one of the chain-only prepare");
  } else {
    slot = h->slot_next;
    h->slot_next = (slot + 1) % rgp_ctx::NSLOT;
    if ((st = rgp_ensure_ws_slot(h, slot, need))) return st;
  }
  hipStream_t s_pre = h->overlap ? h->side_pre : h->stream;
  // Chain-only prepares of consecutive slots (= consecutive layers) alternate between the chain stream and the row stream:
  // a layer's chain is 0.6 ms of dependent one-workgroup kernels (M = 100), and with all layers' chains behind each other on
  // one stream that stream alone set a floor of 2.2 ms per step for short steps (1 - 4 sequences), while the row stream's own
  // work is tens of microseconds there.  (The row work queued behind such a chain waits for it; it is the row work of a later
  // step.)  Only for short steps: measured at M = 100 (steps/s, one stream : alternating) 16 304 rows 413 : 442, 8 152 rows
  // 464 : 533, 4 076 rows 462 : 609, but 32 608 rows 267 : 263 and 130 432 rows 77.4 : 74.3 -- there the row work is long and
  // must not wait behind a chain.  RGP_CHAIN_ALT=0 keeps every chain on the chain stream, 2 alternates at every size.
  static int chain_alt = -1;
  if (chain_alt < 0) { const char* e = getenv("RGP_CHAIN_ALT"); chain_alt = e ? atoi(e) : 1; }
  if (h->overlap && mode == 1 && (slot & 1) && (chain_alt >= 2 || (chain_alt == 1 && N <= 20000))) s_pre = h->side_rows;
  // everything below is off the main stream: row work on side_rows, a chain-only prepare on its chain stream
  hipStream_t s = h->overlap ? (mode == 1 ? s_pre : h->side_rows) : h->stream;
  if (h->overlap) {
    if (mode == 1) {
      // the parameters were updated on the side_post stream (rgp_adam_step_post); the slot's previous users are covered
      // by their own events, so the main stream's queue (later layers of the running step) is not waited for
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_post_pos, h->side_post));
      RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_post_pos, 0));
      if (h->slot_main_recorded[slot]) RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_slot_main[slot], 0));
    } else if (mode == 2 && (in->reserved & RGP_PREPARE_ROWS_EARLY)) {
      // the caller vouches that nothing in flight writes these sequences' signals: only the slot's previous main-stream user
      // has to be finished (its side-stream users are covered by the chain-only prepare, whose prep_z event is waited for below)
      if (h->slot_main_recorded[slot]) RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_slot_main[slot], 0));
    } else {
      // parameters and signals are written by main-stream work (the previous step's update), and the slot's previous
      // user on the main stream must be finished too
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_main, h->stream));
      RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_main, 0));
    }
  }
  carve_layer(h, h->ws_slot[slot], N, M, Q, D, in->n_seq, L, &c);
  if (mode != 2) {
    // the side-stream work that last used this slot must be finished before it is overwritten
    if (h->done_recorded[slot]) RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_done[slot], 0));
    launch_prep_z(s, M, Q, L.zt.QP, in->Z, in->variance, in->inv_l, RGP_JITTER, L.zt);
    if (h->overlap && mode == 1) {                  // the rows-only prepare (side_rows stream) reads c and lam
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_prepz[slot], s));
      h->prepz_recorded[slot] = true;
    }
    c.Kj = L.zt.Kuu; c.noise = in->noise; c.qU_m = in->qU_m; c.qU_W = in->qU_W; c.qU_a = in->qU_a; c.status = h->status;
    c.dLs_sym = L.dLs;                               // the psi2 pass reads the symmetrised dL/dpsi2
    if (h->overlap && s != s_pre) {
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_prep, s));
      RGP_HIP_CHECK(h, hipStreamWaitEvent(s_pre, h->ev_prep, 0));
    }
    if ((st = launch_svi_chain_pre(h, c, s_pre, h->overlap ? h->ev_pre_a[slot] : nullptr, h->overlap ? h->ev_pre_b[slot] : nullptr))) return st;
    if (h->overlap) RGP_HIP_CHECK(h, hipEventRecord(h->ev_pre[slot], s_pre));
  }
  if (mode != 1) {
    if (h->overlap && mode == 2 && h->prepz_recorded[slot]) RGP_HIP_CHECK(h, hipStreamWaitEvent(s, h->ev_prepz[slot], 0));
    // tables + rows
    hipLaunchKernelGGL(svi_tables_kernel, dim3(1), dim3(64), 0, s, in->n_seq, in->X_win, in->own_len,
                       in->U_win > 0 ? in->up_len : nullptr, L.row_off, L.own_pre, L.up_pre, N, D, in->variance, in->noise,
                       L.psi0sum, L.psi0coef);
    const WindowSpec w = svi_window_spec(in, L);
    launch_rowprep_window(s, N, Q, w, in->variance, L.zt, L.rt, L.Ym, in->y_uncertain ? L.Yv : nullptr, L.yry_part);
    if (h->overlap) RGP_HIP_CHECK(h, hipEventRecord(h->ev_rows[slot], s));
    launch_reduce(s, 2048, 1, 1, L.yry_part, L.yry, 1.0, 0);          // only the chain's last stage reads it
    if (h->overlap) RGP_HIP_CHECK(h, hipEventRecord(h->ev_yry[slot], s));
  }
  *slot_out = slot;
  return RGP_OK;
}

extern "C" rgp_status rgp_vardtc_stats(rgp_handle h, const rgp_layer_in* in, double* stats) {
  if (!h || !in || !stats || in->mode != RGP_MODE_VARDTC) return RGP_ERR_BAD_ARG;
  rgp_layer_out dummy{};
  return vardtc_layer_step(h, in, &dummy, 1, stats);
}

extern "C" rgp_status rgp_vardtc_finish(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out, double* stats,
                                        int64_t N_global, double replicated_weight) {
  if (!h || !in || !out || !stats || in->mode != RGP_MODE_VARDTC) return RGP_ERR_BAD_ARG;
  return vardtc_layer_step(h, in, out, 2, stats, N_global, replicated_weight);
}

extern "C" rgp_status rgp_layer_prepare(rgp_handle h, const rgp_layer_in* in) {
  if (!h || !in) return RGP_ERR_BAD_ARG;
  if (in->mode != RGP_MODE_SVI) return RGP_OK;
  if (rgp_is_big(in->M) || rgp_is_wide(in)) return RGP_OK;      // the blocked path runs whole layer steps on the main stream
  rgp_status st = svi_check_args(h, in);
  if (st) return st;
  const int flags = in->reserved & (RGP_PREPARE_CHAIN_ONLY | RGP_PREPARE_ROWS_ONLY);
  if (flags == RGP_PREPARE_ROWS_ONLY) {
    // attach the rows to the oldest prepared layer that has only its chain
    for (int k = 0; k < h->prepared_count; ++k) {
      const int e = (h->prepared_head + k) % rgp_ctx::NSLOT;
      if (!h->prepared_rows[e]) {
        int slot = h->prepared_slot[e];
        if ((st = svi_prepare_slot(h, in, 2, &slot))) return st;
        h->prepared_rows[e] = true;
        return last_launch(h, "rgp_layer_prepare(rows)");
      }
    }
    return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_layer_prepare(rows): no chain-only layer outstanding");
  }
  if (h->prepared_count >= rgp_ctx::NSLOT - 1) return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_layer_prepare: too many layers outstanding");
  int slot;
  if ((st = svi_prepare_slot(h, in, flags == RGP_PREPARE_CHAIN_ONLY ? 1 : 0, &slot))) return st;
  const int e = (h->prepared_head + h->prepared_count) % rgp_ctx::NSLOT;
  h->prepared_slot[e] = slot;
  h->prepared_rows[e] = flags != RGP_PREPARE_CHAIN_ONLY;
  h->prepared_count++;
  return last_launch(h, "rgp_layer_prepare");
}

extern "C" rgp_status rgp_discard_prepared(rgp_handle h) {
  if (!h) return RGP_ERR_BAD_ARG;
  // The abandoned prepares may still be running on the side streams (prep_z / chain_pre on side_pre, row work on side_rows)
  // and no later layer step will record an event behind them.  The next user of each slot waits for ev_done[slot], so that
  // event is recorded here behind both streams.
  if (h->overlap) {
    for (int k = 0; k < h->prepared_count; ++k) {
      const int slot = h->prepared_slot[(h->prepared_head + k) % rgp_ctx::NSLOT];
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_prep, h->side_pre));
      RGP_HIP_CHECK(h, hipStreamWaitEvent(h->side_rows, h->ev_prep, 0));
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_done[slot], h->side_rows));
      h->done_recorded[slot] = true;
    }
  }
  h->prepared_count = 0;
  h->prepared_head = 0;
  return RGP_OK;
}

extern "C" rgp_status rgp_get_stream(rgp_handle h, int which, void** stream) {
  if (!h || !stream) return RGP_ERR_BAD_ARG;
  *stream = (void*)(which == 1 && h->overlap ? h->side_post : h->stream);
  return RGP_OK;
}

// Make `s` wait for `ev` unless the recorded work has already finished: a wait on a completed event is still a packet the
// stream's queue has to process (~4 us), and with pipelined steps the side-stream work a layer step depends on is usually long
// done when the step is issued.
static hipError_t wait_if_pending(hipStream_t s, hipEvent_t ev) {
  if (hipEventQuery(ev) == hipSuccess) return hipSuccess;
  (void)hipGetLastError();                       // hipErrorNotReady is sticky for hipGetLastError
  return hipStreamWaitEvent(s, ev, 0);
}

extern "C" rgp_status rgp_layer_step(rgp_handle h, const rgp_layer_in* in, const rgp_layer_out* out) {
  if (!h || !in || !out) return RGP_ERR_BAD_ARG;
  if (in->mode == RGP_MODE_VARDTC) return vardtc_layer_step(h, in, out);
  if (in->mode != RGP_MODE_SVI) return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_layer_step: unknown mode");
  if (rgp_is_big(in->M) || rgp_is_wide(in)) return big_svi_layer_step(h, in, out);
  const int M = in->M, D = in->D;
  const int Q = in->X_win * in->X_dim + in->U_win * in->U_dim;
  const int64_t N = in->N;
  rgp_status st = svi_check_args(h, in);
  if (st) return st;
  LayerWs L;
  SviChain c{};
  c.N = N; c.M = M; c.D = D; c.qU_ratio = in->qU_ratio;
  hipStream_t s = h->stream;
  hipStream_t s_post = h->overlap ? h->side_post : s;
  int slot;
  if (h->prepared_count > 0) {                       // K_uu + chain_pre already in flight (rgp_layer_prepare)
    if (!h->prepared_rows[h->prepared_head])
      return rgp_fail(h, RGP_ERR_BAD_ARG, "rgp_layer_step: the prepared layer has no rows yet (call rgp_layer_prepare with RGP_PREPARE_ROWS_ONLY)");
    slot = h->prepared_slot[h->prepared_head];
    h->prepared_head = (h->prepared_head + 1) % rgp_ctx::NSLOT;
    h->prepared_count--;
    carve_layer(h, h->ws_slot[slot], N, M, Q, D, in->n_seq, L, &c);
  } else {
    if ((st = svi_prepare_slot(h, in, 0, &slot))) return st;
    carve_layer(h, h->ws_slot[slot], N, M, Q, D, in->n_seq, L, &c);
  }
  c.N = N; c.M = M; c.D = D; c.qU_ratio = in->qU_ratio;
  const int QP = L.zt.QP;

  // the window tables and row terms were produced by svi_prepare_slot (side_rows stream)
  WindowSpec w = svi_window_spec(in, L);
  GAP_MARK(0, true);
  if (h->overlap) RGP_HIP_CHECK(h, wait_if_pending(s, h->ev_rows[slot]));

  c.Kj = L.zt.Kuu; c.noise = in->noise; c.qU_m = in->qU_m; c.qU_W = in->qU_W; c.qU_a = in->qU_a;
  c.psi0sum = L.psi0sum; c.psi1Y = L.psi1Y; c.psi2 = L.psi2; c.yry = L.yry;
  c.gqU_m = out->gqU_m; c.gqU_W = out->gqU_W; c.gqU_a = out->gqU_a; c.winv = out->woodbury_inv; c.status = h->status;

  // psi1 side
  bool psi1_fast = false;
  int psi1y_splits = L.nblk1;                                  // partial slabs of psi1Y to reduce
  if (psi1_svi_supported(D)) {
    if (h->overlap) RGP_HIP_CHECK(h, wait_if_pending(s, h->ev_pre_a[slot]));  // Cpsi1 comes from the chain's first kernel
    GAP_MARK(1, false);
    launch_psi1_svi(s, N, M, Q, D, L.zt, L.rt, L.Ym, c.Cpsi1, in->noise, L.psi1, L.Lp, L.psi1Y_part, L.gmu, L.gS,
                    in->y_uncertain ? L.dYm : nullptr, L.dlam1_part, L.lsum1_part, L.dz1_part, L.nblk1);
    // the partial sums of the three passes are reduced on the side stream together with the psi2 partials (below): nothing
    // on the main stream reads them
    psi1_fast = true;
  } else {
    // wide outputs (D > 16): the contractions over d run on the MFMA GEMM, the three passes keep their D-free parts
    if (h->overlap) RGP_HIP_CHECK(h, wait_if_pending(s, h->ev_pre_a[slot]));  // Cpsi1 comes from the chain's first kernel
    GAP_MARK(1, false);
    {                                                          // T = Ym Cpsi1^T   (dL/dpsi1 up to beta, svi_vardtc.py:164-167)
      GemmArgs g;
      g.m = (int)N; g.n = M; g.k = D; g.A = L.Ym; g.lda = D; g.B = c.Cpsi1; g.ldb = D; g.tb = 1; g.C = L.Lp; g.ldc = M;
      launch_gemm(s, g);
    }
    launch_psi1_svi_wide(s, N, M, Q, L.zt, L.rt, in->noise, L.psi1, L.Lp, L.gmu, L.gS, L.dlam1_part, L.lsum1_part, L.dz1_part,
                         L.nblk1);
    {                                                          // psi1Y = Ym^T psi1 (D x M): split over the rows, partials reduced below
      int64_t sk = (N + 1023) / 1024;
      if (sk > L.nblk1) sk = L.nblk1;
      psi1y_splits = (int)(sk < 1 ? 1 : sk);
      GemmArgs g;
      g.m = D; g.n = M; g.k = (int)N; g.A = L.Ym; g.lda = D; g.ta = 1; g.B = L.psi1; g.ldb = M; g.C = L.psi1Y_part; g.ldc = M;
      g.splits = psi1y_splits; g.split_stride = (int64_t)D * M;
      if (psi1y_splits == 1) g.splits = 2;                     // at least two splits: the output keeps the partial-slab layout
      psi1y_splits = g.splits;
      launch_gemm(s, g);
    }
    if (in->y_uncertain) {                                     // dL/dYmean = psi1 Cpsi1 - beta Ym   (svi_vardtc.py:192)
      GemmArgs g;
      g.m = (int)N; g.n = D; g.k = M; g.A = L.psi1; g.lda = M; g.B = c.Cpsi1; g.ldb = D; g.C = L.dYm; g.ldc = D;
      launch_gemm(s, g);
      hipLaunchKernelGGL(dym_noise_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, s, N * D, L.Ym, in->noise, L.dYm);
    }
    psi1_fast = true;
  }

  Psi2Plan pl2;
  // psi2: forward sum and backward in one pass (dL_dpsi2 does not depend on the statistics in the SVI bound)
  if (h->overlap) RGP_HIP_CHECK(h, wait_if_pending(s, h->ev_pre_b[slot]));    // dL_dpsi2 (symmetrised into L.dLs) comes
  GAP_MARK(2, false);                                                               // from the chain's second kernel
  {
    Psi2Out o{};
    o.p2_part = L.p2_part; o.dz_part = L.dz2_part; o.dlam_part = L.dlam2_part; o.wsum_part = L.wsum2_part;
    o.gmu = L.gmu; o.gS = L.gS; o.accumulate = 1; o.pair_ws = L.pair_ws;
    prof_begin(h, (double)N * M * M * Q);
    const Psi2Plan pl = psi2_plan(h, N, M, QP, true, true);
    st = launch_psi2(h, true, true, N, M, Q, L.zt, L.rt, L.dLs, o, pl.grid);
    prof_end(h);
    if (st) return st;
    pl2 = pl;
  }

  // latent gradients (main stream; they need only the row gradients and dL/dY)
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
  GAP_MARK(3, false);
  // data-dependent algebra + kernel-hyperparameter gradients on the side stream: overlaps the next layer's psi passes
  if (h->overlap) {
    RGP_HIP_CHECK(h, hipEventRecord(h->ev_slot_main[slot], s));        // one event for both uses: every record is a packet the
    h->slot_main_recorded[slot] = true;                                // main stream's queue has to process (~4 us each)
    RGP_HIP_CHECK(h, hipStreamWaitEvent(s_post, h->ev_slot_main[slot], 0));
    RGP_HIP_CHECK(h, hipStreamWaitEvent(s_post, h->ev_pre[slot], 0));
    RGP_HIP_CHECK(h, hipStreamWaitEvent(s_post, h->ev_yry[slot], 0));
  }
  {
    // fixed-order sums of the per-workgroup partials of the psi passes: consumed only by the algebra and the gradient
    // finalisation below.  Short passes (their grid leaves CUs free): on the side stream, not in front of the next layer's psi
    // passes (+2.4 % per step at 4 sequences per rank).  Longer passes leave few or no CUs free, the side-stream reductions would
    // take CUs from the next pass (-0.2 % at 32 sequences), so they stay on the main stream there.
    const bool red_side = h->overlap && pl2.grid <= h->num_cu - 16;      // (248 workgroups, 16 sequences per rank: -1.4 % on the side)
    hipStream_t s_red = red_side ? s_post : s;
    const int tri = M * (M + 1) / 2;
    const int rb = psi1_svi_rows_blocks(N);
    ReduceSeg segs[8];
    int ns = 0;
    if (psi1_fast) {
      segs[ns++] = {L.psi1Y_part, L.psi1Y, psi1y_splits, (int64_t)D * M, (int64_t)D * M};
      segs[ns++] = {L.dlam1_part, L.dlam1, rb, QP, QP};
      segs[ns++] = {L.lsum1_part, L.lsum1, rb, 1, 1};
      segs[ns++] = {L.dz1_part, L.dz1, L.nblk1, (int64_t)M * QP, (int64_t)M * QP};
      launch_reduce_batch(s_red, ns, segs);
      ns = 0;
    }
    segs[ns++] = {L.p2_part, L.p2_tri, pl2.n_p2, tri, tri};
    segs[ns++] = {L.dz2_part, L.dz2, pl2.n_dz, (int64_t)QP * RGP_PSI2_TEAM, (int64_t)QP * RGP_PSI2_TEAM};
    segs[ns++] = {L.dlam2_part, L.dlam2, pl2.n_dl, QP, QP};
    segs[ns++] = {L.wsum2_part, L.wsum2, pl2.n_dl, 1, 1};
    launch_reduce_batch(s_red, ns, segs);
    launch_unpack_tri(s_red, M, L.p2_tri, L.psi2);
    if (h->overlap && !red_side) {                  // the side stream needs the reduced statistics: re-issue its wait
      RGP_HIP_CHECK(h, hipEventRecord(h->ev_slot_main[slot], s));
      RGP_HIP_CHECK(h, hipStreamWaitEvent(s_post, h->ev_slot_main[slot], 0));
    }
  }
  if ((st = launch_svi_chain_post(h, c, s_post))) return st;
  {
    FinalizeArgs f{};
    f.M = M; f.Q = Q; f.QP = QP; f.Zc = L.zt.Zc; f.lam = L.zt.lam; f.variance = in->variance;
    f.Kj = L.zt.Kuu; f.dKmm = c.dKmm; f.dLs = L.dLs; f.psi2 = L.psi2; f.dz1 = L.dz1; f.dz2 = L.dz2;
    f.dlam1 = L.dlam1; f.dlam2 = L.dlam2; f.lsum1 = L.lsum1; f.wsum2 = L.wsum2;
    f.dk_scale = 1.0; f.psi0_coef = 0.0; f.psi0_dev = L.psi0coef; f.psi0_dev_scale = 1.0;
    f.gZ = out->gZ; f.ginv_l = out->ginv_l; f.gvariance = out->gvariance;
    launch_finalize(s_post, f, L.fin_dlrow, L.fin_gsum);
  }
  hipLaunchKernelGGL(finish_elbo_kernel, dim3(1), dim3(64), 0, s_post, c.bound, L.elbo_part, npart, c.dthetaL, out->elbo,
                     out->gnoise, 1.0);
  if (out->woodbury_vector)
    RGP_HIP_CHECK(h, hipMemcpyAsync(out->woodbury_vector, c.Kinv_mu, (size_t)M * D * sizeof(double), hipMemcpyDeviceToDevice, s_post));
  if (h->overlap) {
    RGP_HIP_CHECK(h, hipEventRecord(h->ev_done[slot], s_post));
    h->done_recorded[slot] = true;
  }
  return last_launch(h, "rgp_layer_step");
}

// ==================================================================================================================
// Adam on optimiser-space parameters (softplus for positive ones, paramz Logexp)
// ==================================================================================================================
__global__ void __launch_bounds__(256) adam_kernel(int64_t n, double* __restrict__ x, double* __restrict__ p,
                                                    const double* __restrict__ g, double* __restrict__ m,
                                                    double* __restrict__ v, const uint8_t* __restrict__ positive, double lr,
                                                    double b1, double b2, double eps, double c1, double c2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool pos = positive && positive[i];
  double gi = g[i];
  if (pos) gi *= -expm1(-p[i]);                   // d softplus(x)/dx = 1 - exp(-p)
  const double mi = b1 * m[i] + (1.0 - b1) * gi;
  const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const double xi = x[i] + lr * (mi / c1) / (sqrt(vi / c2) + eps);     // ascent on the ELBO
  x[i] = xi;
  p[i] = pos ? (xi > 36.0 ? xi : log1p(exp(xi))) : xi;
}

extern "C" rgp_status rgp_adam_step_post(rgp_handle h, int64_t n, double* x, double* p, const double* g, double* m, double* v,
                                         const uint8_t* positive, double lr, double beta1, double beta2, double eps,
                                         int64_t step) {
  if (!h || n <= 0) return RGP_ERR_BAD_ARG;
  hipStream_t s = h->overlap ? h->side_post : h->stream;
  const double c1 = 1.0 - pow(beta1, (double)step), c2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, p, g, m, v, positive, lr, beta1,
                     beta2, eps, c1, c2);
  if (h->overlap) {
    RGP_HIP_CHECK(h, hipEventRecord(h->ev_post_tail, s));
    h->post_tail_recorded = true;
  }
  return last_launch(h, "rgp_adam_step_post");
}

extern "C" rgp_status rgp_adam_step(rgp_handle h, int64_t n, double* x, double* p, const double* g, double* m, double* v,
                                    const uint8_t* positive, double lr, double beta1, double beta2, double eps,
                                    int64_t step) {
  if (!h || n <= 0) return RGP_ERR_BAD_ARG;
  const double c1 = 1.0 - pow(beta1, (double)step), c2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, x, p, g, m, v, positive,
                     lr, beta1, beta2, eps, c1, c2);
  return last_launch(h, "rgp_adam_step");
}
