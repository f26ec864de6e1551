"""Thin tensor-level wrapper over the C ABI (rgp_amd/_lib.py).  torch is used only to own device memory and to
name the stream; every call below hands raw pointers to librgp_hip.so."""
import ctypes as C

import torch

from . import _lib
from ._lib import LayerIn, LayerOut, RgpError


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'librgp_hip needs contiguous device tensors'
    return C.c_void_p(t.data_ptr())


def _chk64(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.float64, 'fp64 only'


class Handle(object):
    """One handle per device/rank; all work is enqueued on the torch stream that was current at construction."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('rgp_amd needs a GPU: the hot path has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        st = self.lib.rgp_create(C.byref(h), self.device.index, C.c_void_p(self.stream.cuda_stream))
        if st != 0:
            raise RgpError(st, 'rgp_create failed')
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.lib.rgp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise RgpError(st, self.lib.rgp_last_error(self.h).decode())

    def sync(self):
        """Synchronise and raise if a Cholesky failed since the last call (no jitter retry, see rgp_hip.h)."""
        piv = C.c_int(0)
        self._check(self.lib.rgp_sync_status(self.h, C.byref(piv)))

    def join(self):
        """Make the handle's stream wait for the side-stream work of the preceding layer steps (no host sync)."""
        self._check(self.lib.rgp_join(self.h))

    def new(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def scalar(self, v):
        return torch.tensor([float(v)], dtype=torch.float64, device=self.device)

    def set_profiling(self, on=True):
        self._check(self.lib.rgp_set_profiling(self.h, 1 if on else 0))

    def set_deterministic(self, on=True):
        """Fixed summation order in the psi2 pass (bit-reproducible, slower); include/rgp_hip.h rgp_set_deterministic."""
        self._check(self.lib.rgp_set_deterministic(self.h, 1 if on else 0))

    def set_psi2_pair(self, mode):
        """Kernels of the fused psi2 pass: 0 sweep (vector pipe), 1 pair form where it pays, 2 pair form wherever it fits,
        -1 the RGP_PSI2_PAIR environment variable or 1; include/rgp_hip.h rgp_set_psi2_pair."""
        self._check(self.lib.rgp_set_psi2_pair(self.h, int(mode)))

    def psi2_pair_would_run(self, N, M, Q):
        """True if the fused psi2 pass of an SVI layer step of this size runs in the pair form (include/rgp_hip.h)."""
        return bool(self.lib.rgp_psi2_pair_would_run(self.h, int(N), int(M), int(Q)))

    def side_streams_independent(self):
        """How many of the library's three side streams run next to the main stream (include/rgp_hip.h)."""
        return int(self.lib.rgp_side_streams_independent(self.h))

    def get_profile(self, reset=True):
        ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
        self._check(self.lib.rgp_get_profile(self.h, C.byref(ms), C.byref(n), C.byref(w), 1 if reset else 0))
        return ms.value, n.value, w.value

    # ---- unit pieces ---------------------------------------------------------------------------------------------
    def kuu(self, Z, variance, inv_l, jitter=1e-6):
        _chk64(Z, variance, inv_l)
        M, Q = Z.shape
        out = self.new(M, M)
        self._check(self.lib.rgp_kuu(self.h, M, Q, _p(Z), _p(variance), _p(inv_l), jitter, _p(out)))
        return out

    def matmul(self, A, B, ta=False, tb=False, alpha=1.0, gamma=0.0, out=None, splits=1):
        """out = alpha op(A) op(B) + gamma out on the library's fp64 MFMA product (row-major 2-D tensors, rows may be strided)."""
        _chk64(A, B, out)
        for t in (A, B) + ((out,) if out is not None else ()):
            assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1
        m, k = (A.shape[1], A.shape[0]) if ta else A.shape
        n = B.shape[0] if tb else B.shape[1]
        assert (B.shape[1] if tb else B.shape[0]) == k
        if out is None:
            assert gamma == 0.0
            out = self.new(m, n)
        assert out.shape == (m, n)
        dp = lambda t: C.c_void_p(t.data_ptr())
        self._check(self.lib.rgp_matmul(self.h, m, n, k, alpha, dp(A), A.stride(0), int(ta), dp(B), B.stride(0), int(tb), gamma,
                                        dp(out), out.stride(0), splits))
        return out

    def psi_forward(self, mu, S, Z, variance, inv_l, want_psi1=True, want_psi2=True):
        _chk64(mu, S, Z, variance, inv_l)
        N, Q = mu.shape
        M = Z.shape[0]
        p1 = self.new(N, M) if want_psi1 else None
        p2 = self.new(M, M) if want_psi2 else None
        self._check(self.lib.rgp_psi_forward(self.h, N, M, Q, _p(mu), _p(S), _p(Z), _p(variance), _p(inv_l), _p(p1), _p(p2)))
        return p1, p2

    def psi_forward_static(self, mu, S, Z, variance, inv_l, psi1, psi2):
        """rgp_psi_forward_static: the caller passes the output tensors too; the call allocates nothing."""
        _chk64(mu, S, Z, variance, inv_l, psi1, psi2)
        N, Q = mu.shape
        M = Z.shape[0]
        self._check(self.lib.rgp_psi_forward_static(self.h, N, M, Q, _p(mu), _p(S), _p(Z), _p(variance), _p(inv_l), _p(psi1), _p(psi2)))

    def psi_backward(self, mu, S, Z, variance, inv_l, dL_dpsi0_sum, dL_dpsi1, dL_dpsi2):
        _chk64(mu, S, Z, variance, inv_l, dL_dpsi1, dL_dpsi2)
        N, Q = mu.shape
        M = Z.shape[0]
        dvar, dil, dZ = self.new(1), self.new(Q), self.new(M, Q)
        dmu, dS = self.new(N, Q), self.new(N, Q)
        self._check(self.lib.rgp_psi_backward(self.h, N, M, Q, _p(mu), _p(S), _p(Z), _p(variance), _p(inv_l),
                                              float(dL_dpsi0_sum), _p(dL_dpsi1), _p(dL_dpsi2), _p(dvar), _p(dil), _p(dZ),
                                              _p(dmu), _p(dS)))
        return dict(dvar=dvar, dinv_l=dil, dZ=dZ, dmu=dmu, dS=dS)

    def svi_chain(self, N, Kuu_jit, noise, qU_m, qU_W, qU_a, psi0sum, psi1Y, psi2, yry, qU_ratio=1.0):
        M, D = qU_m.shape
        o = dict(bound=self.new(1), dL_dKmm=self.new(M, M), dL_dpsi2=self.new(M, M), Kinv_mu=self.new(M, D),
                 dL_dthetaL=self.new(1), gqU_m=self.new(M, D), gqU_W=self.new(M, M), gqU_a=self.new(1),
                 woodbury_inv=self.new(M, M))
        self._check(self.lib.rgp_svi_chain(self.h, N, M, D, float(qU_ratio), _p(Kuu_jit), _p(noise), _p(qU_m), _p(qU_W),
                                           _p(qU_a), _p(psi0sum), _p(psi1Y), _p(psi2), _p(yry), _p(o['bound']),
                                           _p(o['dL_dKmm']), _p(o['dL_dpsi2']), _p(o['Kinv_mu']), _p(o['dL_dthetaL']),
                                           _p(o['gqU_m']), _p(o['gqU_W']), _p(o['gqU_a']), _p(o['woodbury_inv'])))
        return o

    # ---- fused layer step ----------------------------------------------------------------------------------------
    def vardtc_stats(self, lin, stats):
        self._check(self.lib.rgp_vardtc_stats(self.h, C.byref(lin), _p(stats)))

    def vardtc_finish(self, lin, lout, stats, N_global, weight):
        self._check(self.lib.rgp_vardtc_finish(self.h, C.byref(lin), C.byref(lout), _p(stats), int(N_global), float(weight)))

    def layer_prepare(self, lin, flags=0):
        """flags: 0 = chain + rows, PREPARE_CHAIN_ONLY, PREPARE_ROWS_ONLY (include/rgp_hip.h, pipelining section)."""
        old = lin.reserved
        lin.reserved = int(flags)
        try:
            self._check(self.lib.rgp_layer_prepare(self.h, C.byref(lin)))
        finally:
            lin.reserved = old

    def discard_prepared(self):
        self._check(self.lib.rgp_discard_prepared(self.h))

    def get_stream(self, which=0):
        out = C.c_void_p(0)
        self._check(self.lib.rgp_get_stream(self.h, int(which), C.byref(out)))
        return out.value

    def layer_step(self, lin, lout):
        self._check(self.lib.rgp_layer_step(self.h, C.byref(lin), C.byref(lout)))

    def freerun(self, n_seq, M, D, X_win, U_win, U_dim, steps, Z, variance, inv_l, winv, wvec, init_mean, init_var,
                U_mean, U_var, u_var_const, u_rows, u_skip):
        Xm = self.new(n_seq, X_win + steps, D)
        Xv = self.new(n_seq, X_win + steps, D)
        self._check(self.lib.rgp_freerun(self.h, n_seq, M, D, X_win, U_win, U_dim, steps, _p(Z), _p(variance), _p(inv_l),
                                         _p(winv), _p(wvec), _p(init_mean), _p(init_var), _p(U_mean), _p(U_var),
                                         float(u_var_const), int(u_rows), int(u_skip), _p(Xm), _p(Xv)))
        return Xm, Xv

    def adam_step(self, x, p, g, m, v, positive, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, post=False):
        """post=True: on the library's side stream, ordered after the gradients of the last layer step (rgp_adam_step_post)."""
        n = x.numel()
        fn = self.lib.rgp_adam_step_post if post else self.lib.rgp_adam_step
        self._check(fn(self.h, n, _p(x), _p(p), _p(g), _p(m), _p(v),
                                           C.c_void_p(positive.data_ptr()) if positive is not None else None,
                                           float(lr), float(beta1), float(beta2), float(eps), int(step)))


def _mlp_args(spec, dims, n_seq, Ws, bs, init_mean=None, up=None, mean=None, acts=(), g_mean=None, g_init=None, g_up=None, deltas=()):
    a = _lib.MlpScan()
    a.n_seq, a.X_win, a.X_dim, a.U_win, a.U_dim = n_seq, spec['X_win'], spec['X_dim'], spec['U_win'], spec['U_dim']
    a.N, a.up_len, a.n_layers = spec['N'], (0 if up is None else up.shape[1]), len(Ws)
    for k, v in enumerate(dims):
        a.dims[k] = v
    for k, (W, b) in enumerate(zip(Ws, bs)):
        _chk64(W, b)
        a.W[k], a.b[k] = W.data_ptr(), b.data_ptr()
    for k, t in enumerate(acts):
        a.act[k] = t.data_ptr()
    for k, t in enumerate(deltas):
        a.delta[k] = t.data_ptr()
    for name, t in (('init_mean', init_mean), ('up', up), ('mean', mean), ('g_mean', g_mean), ('g_init', g_init), ('g_up', g_up)):
        if t is not None:
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64
            setattr(a, name, t.data_ptr())
    return a


def mlp_scan_forward(handle, spec, dims, init_mean, up, Ws, bs):
    """rgp_mlp_scan_forward: returns (mean (B, X_win+N, X_dim), [hidden activations (B, N, dims[k+1])])."""
    if len(Ws) > _lib.MLP_MAX_LAYERS:
        raise RgpError(_lib.RGP_ERR_BAD_ARG, 'mlp scan: at most %d linear layers' % _lib.MLP_MAX_LAYERS)
    B = init_mean.shape[0]
    mean = handle.new(B, spec['X_win'] + spec['N'], spec['X_dim'])
    acts = [handle.new(B, spec['N'], dims[k + 1]) for k in range(len(Ws) - 1)]
    a = _mlp_args(spec, dims, B, Ws, bs, init_mean=init_mean, up=up, mean=mean, acts=acts)
    handle._check(handle.lib.rgp_mlp_scan_forward(handle.h, C.byref(a)))
    return mean, acts


def mlp_scan_backward(handle, spec, dims, up, Ws, bs, acts, g_mean, want_g_up):
    """rgp_mlp_scan_backward: returns (g_init (B, X_win, X_dim), g_up or None, [delta_k (B, N, dims[k+1])])."""
    B = g_mean.shape[0]
    g_init = handle.new(B, spec['X_win'], spec['X_dim'])
    g_up = handle.new(*up.shape) if (want_g_up and up is not None) else None
    deltas = [handle.new(B, spec['N'], dims[k + 1]) for k in range(len(Ws))]
    a = _mlp_args(spec, dims, B, Ws, bs, up=up, acts=acts, g_mean=g_mean, g_init=g_init, g_up=g_up, deltas=deltas)
    handle._check(handle.lib.rgp_mlp_scan_backward(handle.h, C.byref(a)))
    return g_init, g_up, deltas


_CELL_G = {_lib.CELL_GRU: (3, 4), _lib.CELL_RNN_TANH: (1, 0), _lib.CELL_LSTM: (4, 5)}      # gate rows, stored values per unit


def rnn_scan_forward(handle, cell, gi, W_hh, b_hh):
    """rgp_rnn_scan_forward: gi (T, B, G H) -> (hseq (T, B, H), gates (T, B, NS H) or None)."""
    _chk64(gi, W_hh, b_hh)
    G, NS = _CELL_G[cell]
    T, B, HG = gi.shape
    H = HG // G
    hseq = handle.new(T, B, H)
    gates = handle.new(T, B, NS * H) if NS else None
    a = _lib.RnnScan()
    a.T, a.B, a.H, a.cell = T, B, H, cell
    a.gi, a.W_hh, a.b_hh, a.hseq = gi.data_ptr(), W_hh.data_ptr(), b_hh.data_ptr(), hseq.data_ptr()
    a.gates = gates.data_ptr() if gates is not None else None
    handle._check(handle.lib.rgp_rnn_scan_forward(handle.h, C.byref(a)))
    return hseq, gates


def rnn_scan_backward(handle, cell, W_hh, b_hh, hseq, gates, g_out):
    """rgp_rnn_scan_backward: g_out (T, B, H) -> (dgi, dgh), both (T, B, G H)."""
    _chk64(g_out)
    G, NS = _CELL_G[cell]
    T, B, H = hseq.shape
    dgi, dgh = handle.new(T, B, G * H), handle.new(T, B, G * H)
    a = _lib.RnnScan()
    a.T, a.B, a.H, a.cell = T, B, H, cell
    a.W_hh, a.b_hh, a.hseq = W_hh.data_ptr(), b_hh.data_ptr(), hseq.data_ptr()
    a.gates = gates.data_ptr() if gates is not None else None
    a.g_out, a.dgi, a.dgh = g_out.data_ptr(), dgi.data_ptr(), dgh.data_ptr()
    handle._check(handle.lib.rgp_rnn_scan_backward(handle.h, C.byref(a)))
    return dgi, dgh


def ptr(t):
    return None if t is None else t.data_ptr()
