// rgp_chain.h -- M x M algebra of the SVI bound (rgp_chain.hip). Internal to librgp_hip.so.
#pragma once
#include "rgp_common.h"

struct SviChain {
  int64_t N; int M, D; double qU_ratio;
  // inputs (device)
  const double *Kj, *noise, *qU_m, *qU_W, *qU_a;
  const double *psi0sum, *psi1Y, *psi2, *yry;          // stage 2 only
  // workspace (device, carved by svi_chain_carve)
  double *Linv, *S, *LsInv, *B, *T1, *T2, *T3, *mA, *Kinv, *Gv;
  double *dLdpsi2, *dKmm;                               // results consumed by the psi backward / K backward
  double *dLs_sym = nullptr;                            // optional (layer workspace): 1/2 (dL_dpsi2 + dL_dpsi2^T) for the psi2 pass
  double *Sinv, *Wi;                                    // S^-1 and woodbury_inv, produced by stage 1c for stage 2
  double *gmu, *Kinv_mu, *Cpsi1, *Ybt, *gqm;
  double *scal, *bound, *dthetaL;
  double *bigD = nullptr, *bigP = nullptr;              // M > 128 only: inverses of the diagonal blocks, panel buffer
  // optional outputs (device, caller owned; may be null)
  double *gqU_m, *gqU_W, *gqU_a, *winv;
  int* status;
};

size_t svi_chain_ws_doubles(int M, int D);
void svi_chain_carve(SviChain& c, WsAlloc& ws);
// three kernels: (a) K_uu factor + Cpsi1, (b) dL_dpsi2, (c) KL gradient terms; the events, if given, are recorded after
// (a) and (b)
rgp_status launch_svi_chain_pre(rgp_handle h, const SviChain& c, hipStream_t s, hipEvent_t ev_after_a = nullptr,
                                hipEvent_t ev_after_b = nullptr);
rgp_status launch_svi_chain_post(rgp_handle h, const SviChain& c, hipStream_t s);

// ---- collapsed (Titsias) bound, autoreg/inference/vardtc.py:88-208 ------------------------------------------------
struct VarDtcChain {
  int64_t N; int M, D; int y_uncertain;
  // inputs (device)
  const double *Kj, *noise;
  const double *psi0sum, *psi1Y, *psi2, *yry;       // psi1Y: D x M (unscaled)
  const double *PsiR;                               // M x M: sum_n r_n^2 psi1_n^T psi1_n (uncertain outputs only)
  // workspace
  double *Linv, *A, *LLinv, *Li, *G2, *T1, *T2, *T3, *C, *R, *dKmm, *dLdpsi2, *bT, *vT, *Vb;
  double *scal, *bound, *dthetaL, *beta;
  double *bigD = nullptr, *bigP = nullptr;           // M > 128 only: inverses of the diagonal blocks, panel buffer
  double *winv;                                      // optional output
  int* status;
};
size_t vardtc_chain_ws_doubles(int M, int D);
void vardtc_chain_carve(VarDtcChain& c, WsAlloc& ws);
rgp_status launch_vardtc_chain(rgp_handle h, const VarDtcChain& c);
