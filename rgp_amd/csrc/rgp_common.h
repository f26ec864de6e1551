// rgp_common.h -- shared device helpers and the handle of librgp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rgp_hip.h"

#define RGP_JITTER 1e-6   // const_jitter, autoreg/inference/vardtc.py:28, svi_vardtc.py:26

static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Broadcast lane `src` (wave-uniform index) of a double to a scalar.
static __device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// DPP lane exchange of a double (two 32-bit moves); CTRL is a gfx9 dpp_ctrl value.
template <int CTRL>
static __device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of 8 lanes, result in every lane of the group (no LDS round trips)
static __device__ __forceinline__ double sum8_dpp(double x) {
  x += dpp_f64<0xB1>(x);            // quad_perm [1,0,3,2]
  x += dpp_f64<0x4E>(x);            // quad_perm [2,3,0,1]
  x += dpp_f64<0x141>(x);           // row_half_mirror
  return x;
}
// sum over the wave, result in every lane; fixed order
static __device__ __forceinline__ double wave_sum_dpp(double x) {
  x = sum8_dpp(x);
  x += dpp_f64<0x140>(x);           // row_mirror: rows of 16 complete
  return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}

// Sum over the whole workgroup; result valid in every thread. `scratch` holds >= 16 doubles of LDS.
static __device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += scratch[i];   // fixed order -> deterministic
  return s;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// padded window width: multiples of 8, plus 20 (the Q = 20 top layer of the benchmark geometry would waste 17% of the
// multiply-adds at 24)
static inline int rgp_qp(int Q) { return (Q > 16 && Q <= 20) ? 20 : round_up(Q, 8); }

struct rgp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // workspace (grown on demand, never shrunk).  Layer steps alternate between two slots so that the M x M chain of one
  // layer (side streams) can overlap the psi passes of the next layer (main stream).
  char* ws = nullptr;
  size_t ws_bytes = 0;
  static const int NSLOT = 8;        // ring of layer workspaces: the layers of one step plus the next step's early chains
  char* ws_slot[NSLOT] = {};
  size_t ws_slot_bytes[NSLOT] = {};
  int slot_next = 0;                 // next slot to hand out
  int side_streams_independent = 0;            // how many side streams were seen running next to the main stream (3 = all)
  int prepared_head = 0, prepared_count = 0;   // FIFO of slots prepared by rgp_layer_prepare
  int prepared_slot[NSLOT] = {};
  bool prepared_rows[NSLOT] = {};     // FIFO entry also has its window tables / row terms (false: chain-only prepare)
  bool overlap = true;
  bool deterministic = false;       // rgp_set_deterministic: fixed-order psi2 kernels only
  int psi2_pair = -1;               // rgp_set_psi2_pair: 0 sweep kernels, 1 pair form where it pays, 2 wherever it fits; -1: RGP_PSI2_PAIR or 1
  hipStream_t side_pre = nullptr, side_post = nullptr, side_rows = nullptr;
  hipEvent_t ev_prep = nullptr, ev_stats = nullptr;
  hipEvent_t ev_pre[NSLOT] = {};
  hipEvent_t ev_pre_a[NSLOT] = {};   // K_uu factor + Cpsi1 ready (psi1 passes)
  hipEvent_t ev_pre_b[NSLOT] = {};   // dL_dpsi2 ready (psi2 pass)
  hipEvent_t ev_rows[NSLOT] = {};    // window tables + row terms ready (side_rows stream)
  hipEvent_t ev_prepz[NSLOT] = {};   // Z-derived terms of the slot written (chain-only prepare, side_pre stream)
  bool prepz_recorded[NSLOT] = {};
  hipEvent_t ev_yry[NSLOT] = {};     // sum of squared outputs reduced (needed by the chain's last stage only)
  hipEvent_t ev_main = nullptr;                                          // main-stream position at prepare time
  hipEvent_t ev_done[NSLOT] = {};
  bool done_recorded[NSLOT] = {};
  hipEvent_t ev_slot_main[NSLOT] = {};   // main-stream work of the layer step that last used the slot
  bool slot_main_recorded[NSLOT] = {};
  hipEvent_t ev_post_pos = nullptr;      // position of the side_post stream (parameter updates issued there)
  hipEvent_t ev_post_tail = nullptr;     // last rgp_adam_step_post
  bool post_tail_recorded = false;
  int* status = nullptr;        // device: [0] = error code, [1] = pivot
  char* static_scratch = nullptr;    // this device's address of the library's static scratch array (rgp_psi_forward_static)
  int num_cu = 256;
  unsigned vardtc_rr = 0;            // collapsed-bound layer steps rotate over the three streams
  int vardtc_prev_slot = -1;         // workspace slot (its ev_done) of the previous collapsed-bound layer step
  // optional profiling of the psi2 launch (rgp_set_profiling)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_nmmq = 0.0;
};

struct WsAlloc {           // bump allocator over the handle's workspace
  char* base; size_t off = 0;
  explicit WsAlloc(char* b) : base(b) {}
  template <typename T> T* take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += bytes;
    return p;
  }
};

rgp_status rgp_fail(rgp_handle h, rgp_status code, const std::string& msg);
rgp_status rgp_ensure_ws(rgp_handle h, size_t bytes);
rgp_status rgp_ensure_ws_slot(rgp_handle h, int slot, size_t bytes);
void rgp_join_streams(rgp_handle h);

#define RGP_HIP_CHECK(h, expr)                                                        \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) return rgp_fail(h, RGP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
