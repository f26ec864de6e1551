"""Host side of the MI355X recurrent-GP path: the model object that mirrors ``autoreg.DeepAutoreg`` /
``autoreg.DeepAutoreg_new`` (/root/reference/autoreg/model.py:632-740, :15-282) and drives librgp_hip.so.

What lives where
  * this file: data shift, latent initialisation, parameter layout, the top-down layer loop, optimiser loops;
    all tensors are torch fp64 device tensors (torch = storage + stream + torch.distributed only);
  * librgp_hip.so (include/rgp_hip.h): everything numerical per layer (windows, K_uu, psi statistics, M x M algebra,
    all gradients, window scatter, entropy/prior terms, Adam).

Parameter layout -- ONE flat fp64 device tensor ``param_array`` (constrained values, like paramz' param_array) and a
mirror ``grad_array``:
    [ global parameters of every layer, top layer first | ELBO slot | latent means | latent variances ]
  * the global segment (+ELBO slot) is the single all-reduce bucket of data-parallel SVI;
  * latent means / variances are grouped per SEQUENCE (for s: for hidden layer i top..1: T_{s,i} x H_i block), so the
    local parameters of a contiguous minibatch are one contiguous range.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from .capi import Handle
from ._lib import LayerIn, LayerOut

POSITIVE = ('variance', 'inv_l', 'noise', 'qU_a')


def _softplus_inv(p):
    return torch.where(p > 36.0, p, torch.log(torch.expm1(torch.clamp(p, max=36.0))))


def _softplus(x):
    return torch.where(x > 36.0, x, torch.log1p(torch.exp(torch.clamp(x, max=36.0))))


class _Kern(object):
    """``layer.kern`` with GPy's parameter names (RBF(ARD=True, inv_l=True)): ``variance``, ``inv_l`` (=1/lengthscale^2)."""

    def __init__(self, variance, inv_l):
        self.variance, self.inv_l = variance, inv_l

    @property
    def lengthscale(self):
        return 1.0 / torch.sqrt(self.inv_l)


class _Lik(object):
    def __init__(self, variance):
        self.variance = variance


class LayerView(object):
    """Named views of one layer's parameters / gradients inside the flat arrays (what ``model.layers[k]`` exposes in
    the reference: ``.kern.variance``, ``.kern.inv_l``, ``.likelihood.variance``, ``.Z``, ``.qU_mean``, ``.qU_W``, ``.qU_a``)."""

    def __init__(self, index, name, X_win, X_dim, U_win, U_dim, M, D, observed):
        self.index, self.name = index, name
        self.X_win, self.X_dim, self.U_win, self.U_dim, self.M, self.D = X_win, X_dim, U_win, U_dim, M, D
        self.Q = X_win * X_dim + U_win * U_dim
        self.X_observed = observed
        self.slices = {}
        self.qU_ratio = 1.0


class DeepAutoreg(object):
    """Deep autoregressive (recurrent) GP; constructor arguments follow autoreg.DeepAutoreg_new (model.py:28-33).

    wins[0] is the observed layer's own window (must be 0, as in the reference), wins[i>0] the hidden windows;
    ``self.layers`` is ordered top layer first like the reference's ``model.layers`` (model.py:94-109).
    Back-constraints are separate classes: ``DeepAutoreg_rnn`` (sequence encoder) and ``DeepAutoreg_mlp`` (MLP recursion)."""

    def __init__(self, wins, Y, U=None, U_win=1, nDims=None, X_variance=0.01, num_inducing=10, likelihood=None,
                 name='autoreg', kernels=None, U_pre_step=True, init='Y', inducing_init='rand', back_cstr=False,
                 MLP_dims=None, inference_method='svi', minibatch_inference=False, mb_inf_tot_data_size=None,
                 mb_inf_sample_idxes=None, device=None, seed=0, Zs=None, latent_full=False):
        if back_cstr:
            raise NotImplementedError('use DeepAutoreg_mlp (MLP recursion) or DeepAutoreg_rnn (sequence encoder) for back-constraints')
        if not U_pre_step:
            raise NotImplementedError('U_pre_step=False')
        assert wins[0] == 0, 'the observed layer cannot be autoregressive (as in the reference)'
        self.name = name
        self.handle = Handle(device)
        self.device = self.handle.device
        rng = np.random.default_rng(seed)
        Ys = [Y] if isinstance(Y, np.ndarray) else list(Y)
        Us = None if U is None else ([U] if isinstance(U, np.ndarray) else list(U))
        self.wins, self.nLayers, self.U_win = list(wins), len(wins), U_win
        self.svi = (inference_method == 'svi')
        self.mode = _lib.RGP_MODE_SVI if self.svi else _lib.RGP_MODE_VARDTC
        self.nDims = list(nDims) if nDims is not None else [Ys[0].shape[1]] + [1] * (self.nLayers - 1)
        # a1: shift (model.py:53-69)
        if Us is not None:
            assert len(Us) == len(Ys)
            Us = [np.asarray(u, dtype=np.float64)[:-1] for u in Us]
            Ys = [np.asarray(y, dtype=np.float64)[U_win:] for y in Ys]
        else:
            Ys = [np.asarray(y, dtype=np.float64) for y in Ys]
        self.nSeq = len(Ys)
        self.n_total = self.nSeq if mb_inf_tot_data_size is None else int(mb_inf_tot_data_size)
        self.Ly = np.array([y.shape[0] for y in Ys], dtype=np.int64)
        self.Dy = Ys[0].shape[1]
        self.Du = Us[0].shape[1] if Us is not None else 0
        dev = self.device
        f64 = dict(dtype=torch.float64, device=dev)
        self.Ycat = torch.tensor(np.concatenate(Ys, 0), **f64).contiguous()
        self.y_off = np.concatenate([[0], np.cumsum(self.Ly[:-1] * self.Dy)]).astype(np.int64)
        if Us is not None:
            self.Ucat = torch.tensor(np.concatenate(Us, 0), **f64).contiguous()
            self.u_len = np.array([u.shape[0] for u in Us], dtype=np.int64)
            self.u_off = np.concatenate([[0], np.cumsum(self.u_len[:-1] * self.Du)]).astype(np.int64)
        else:
            self.Ucat = None
        # ---- layers (top first) ----
        if not isinstance(num_inducing, (list, tuple)):
            num_inducing = [num_inducing] * self.nLayers
        self.order = list(range(self.nLayers - 1, -1, -1))
        self.layers = []
        for i in self.order:
            top = (i == self.nLayers - 1)
            X_win, X_dim = self.wins[i], self.nDims[i]
            if top:
                Uw, Ud = (U_win, self.Du) if Us is not None else (0, 0)
            else:
                Uw, Ud = self.wins[i + 1], self.nDims[i + 1]
            self.layers.append(LayerView(i, 'layer_%d' % i, X_win, X_dim, Uw, Ud, int(num_inducing[i]), self.nDims[i], i == 0))
        # ---- flat layout ----
        off = 0
        for L in self.layers:
            spec = [('Z', (L.M, L.Q)), ('variance', (1,)), ('inv_l', (L.Q,)), ('noise', (1,))]
            if self.svi:
                spec += [('qU_m', (L.M, L.D)), ('qU_W', (L.M, L.M)), ('qU_a', (1,))]
            for nm, shp in spec:
                n = int(np.prod(shp))
                L.slices[nm] = (off, off + n, shp)
                off += n
        self.n_global = off
        self.elbo_slot = off
        off += 1
        self.loc0 = off
        # latent blocks: element offsets relative to the start of the means region
        # latent_full: one latent row per RAW time step (latents produced by a sequence encoder, model.py:427-440)
        self.latent_full = latent_full
        self.lat_len = {i: self.Ly + ((U_win if Us is not None else 0) if latent_full else self.wins[i])
                        for i in range(1, self.nLayers)}
        self.lat_off = {i: np.zeros(self.nSeq, dtype=np.int64) for i in range(1, self.nLayers)}
        self.seq_loc_off = np.zeros(self.nSeq + 1, dtype=np.int64)
        o = 0
        for s in range(self.nSeq):
            self.seq_loc_off[s] = o
            for i in range(self.nLayers - 1, 0, -1):
                self.lat_off[i][s] = o
                o += int(self.lat_len[i][s]) * self.nDims[i]
        self.seq_loc_off[self.nSeq] = o
        self.n_local = o
        self.n_param = self.loc0 + 2 * self.n_local
        self.param_array = torch.zeros(self.n_param, **f64)
        self.grad_array = torch.zeros(self.n_param, **f64)
        pos = torch.zeros(self.n_param, dtype=torch.uint8, device=dev)
        for L in self.layers:
            for nm in POSITIVE:
                if nm in L.slices:
                    a, b, _ = L.slices[nm]
                    pos[a:b] = 1
        pos[self.loc0 + self.n_local:] = 1
        self.positive = pos
        self.means = self.param_array[self.loc0:self.loc0 + self.n_local]
        self.vars = self.param_array[self.loc0 + self.n_local:]
        self.gmeans = self.grad_array[self.loc0:self.loc0 + self.n_local]
        self.gvars = self.grad_array[self.loc0 + self.n_local:]
        # ---- a2: hidden q(X) initialisation (model.py:129-156) ----
        means = np.zeros(self.n_local)
        for i in range(1, self.nLayers):
            H, w = self.nDims[i], self.wins[i]
            for s in range(self.nSeq):
                T = int(self.lat_len[i][s])
                blk = np.zeros((T, H))
                if init == 'Y':
                    blk[T - int(self.Ly[s]):] = Ys[s][:, :H]
                elif init == 'rand':
                    blk[:] = rng.standard_normal(blk.shape)
                elif init == 'zero':
                    blk[:] = rng.standard_normal(blk.shape) * 0.01
                else:
                    raise ValueError(init)
                means[self.lat_off[i][s]:self.lat_off[i][s] + T * H] = blk.ravel()
        self.means.copy_(torch.tensor(means, **f64))
        self.vars.fill_(X_variance)
        # ---- global parameter initialisation (layers.py:1332-1345, :51-58; model.py:94-109) ----
        for L in self.layers:
            i = L.index
            Xm = self._host_X_mean(i, Ys, Us, means)
            if Zs is not None:
                Z = np.asarray(Zs[i], dtype=np.float64)
            elif inducing_init == 'kmeans':
                from sklearn.cluster import KMeans
                km = KMeans(n_clusters=L.M, n_init=1000, max_iter=100)       # layers.py:1335
                km.fit(Xm.copy())
                Z = km.cluster_centers_.copy()
            else:
                Z = rng.standard_normal((L.M, L.Q))                            # layers.py:1339
            self.set_param(L, 'Z', Z)
            if kernels is not None and kernels[i] is not None:
                kv, kil = kernels[i]
                self.set_param(L, 'variance', kv)
                self.set_param(L, 'inv_l', kil)
            else:
                self.set_param(L, 'variance', 1.0)
                self.set_param(L, 'inv_l', np.ones(L.Q))
            self.set_param(L, 'noise', 1.0 if i == 0 else 0.01)               # model.py:96-109
            if self.svi:
                from scipy.linalg import lstsq
                Ym = np.concatenate(Ys, 0) if i == 0 else self._host_Y_mean(i, means)
                W = lstsq(Xm, Ym)[0]                                          # util.py:32-38
                self.set_param(L, 'qU_m', Z.dot(W))
                self.set_param(L, 'qU_W', rng.standard_normal((L.M, L.M)) * 0.01)
                self.set_param(L, 'qU_a', 1e-3)
        self._tables()
        self._elbo_buf = torch.zeros(self.nLayers, **f64)
        self._winv = {L.index: torch.zeros(L.M, L.M, **f64) for L in self.layers}
        self._wvec = {L.index: torch.zeros(L.M, L.D, **f64) for L in self.layers}
        self.opt_array = None
        self._adam = None
        self.data_streamer = None
        self.mb_inf_sample_idxes = None
        # consecutive svi_step calls overlap the next step's M x M chains with the running step (see _svi_step_pipelined);
        # every other way of changing parameters goes through _flush_pipeline first.  Code that writes param_array directly
        # between svi_step calls must call params_changed().
        self.pipeline = os.environ.get('RGP_NO_PIPELINE') is None
        # data-parallel steps: False = each layer's global gradients are all-reduced as soon as they exist (pipelined steps,
        # one collective per layer); True = ONE all-reduce of the whole [global gradients | ELBO] bucket per step, the exchange
        # the reference's dead stub stood for (svi_vardtc.py:65-67)
        self.single_bucket = False
        self._chain_ready = False
        self._chain_n = None
        self._post_stream = None

    # ------------------------------------------------------------------------------------------------------------
    # host-side helpers used only at construction (setup, not the hot path)
    # ------------------------------------------------------------------------------------------------------------
    def _host_latent(self, i, s, means):
        T, H = int(self.lat_len[i][s]), self.nDims[i]
        return means[self.lat_off[i][s]:self.lat_off[i][s] + T * H].reshape(T, H)

    def _host_X_mean(self, i, Ys, Us, means):
        """Window design matrix of layer i on the host (layers.py:1399-1410) -- initialisation only."""
        rows = []
        w = self.wins[i]
        for s in range(self.nSeq):
            N = int(self.Ly[s]) if i == 0 else int(self.lat_len[i][s]) - w
            cols = []
            if w > 0:
                own = self._host_latent(i, s, means)[:-1]
                idx = np.arange(N)[:, None] + np.arange(w)[None, :]
                cols.append(own[idx].reshape(N, -1))
            if i == self.nLayers - 1:
                up, uw = (Us[s], self.U_win) if Us is not None else (None, 0)
            else:
                up, uw = self._host_latent(i + 1, s, means), self.wins[i + 1]
            if up is not None:
                up = up[up.shape[0] - N - uw + 1:]
                idx = np.arange(N)[:, None] + np.arange(uw)[None, :]
                cols.append(up[idx].reshape(N, -1))
            rows.append(np.hstack(cols))
        return np.vstack(rows)

    def _host_Y_mean(self, i, means):
        return np.vstack([self._host_latent(i, s, means)[self.wins[i]:] for s in range(self.nSeq)])

    def _tables(self):
        dev = self.device
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.int64, device=dev)
        self.tab = {}
        for L in self.layers:
            i = L.index
            d = {}
            if i == 0:
                d['own_off'], d['own_len'] = self.y_off, self.Ly
            else:
                d['own_off'], d['own_len'] = self.lat_off[i], self.lat_len[i]
            if i == self.nLayers - 1:
                if self.Ucat is not None:
                    d['up_off'], d['up_len'] = self.u_off, self.u_len
                else:
                    d['up_off'], d['up_len'] = np.zeros(self.nSeq, np.int64), np.zeros(self.nSeq, np.int64)
            else:
                d['up_off'], d['up_len'] = self.lat_off[i + 1], self.lat_len[i + 1]
            self.tab[i] = dict(host=d, dev={k: t(v) for k, v in d.items()})

    # ------------------------------------------------------------------------------------------------------------
    # parameter access
    # ------------------------------------------------------------------------------------------------------------
    def layer(self, i):
        return self.layers[self.nLayers - 1 - i]

    def view(self, L, name, grad=False):
        a, b, shp = L.slices[name]
        return (self.grad_array if grad else self.param_array)[a:b].view(*shp)

    def params_changed(self):
        """Call after writing ``param_array`` directly (paramz does this implicitly): invalidates work in flight."""
        self._flush_pipeline()
        self._adam = None

    def set_param(self, L, name, value):
        self._flush_pipeline()
        v = torch.as_tensor(np.asarray(value, dtype=np.float64), device=self.device)
        self.view(L, name).copy_(v.reshape(self.view(L, name).shape))

    def latent(self, i, s, var=False, grad=False):
        T, H = int(self.lat_len[i][s]), self.nDims[i]
        base = (self.gvars if var else self.gmeans) if grad else (self.vars if var else self.means)
        o = int(self.lat_off[i][s])
        return base[o:o + T * H].view(T, H)

    # ------------------------------------------------------------------------------------------------------------
    # one objective + gradient evaluation: top-down layer loop (model.py:161-187)
    # ------------------------------------------------------------------------------------------------------------
    def _seq_tables(self, L, seqs):
        tb = self.tab[L.index]
        if seqs is None:
            dev = tb['dev']
            host = tb['host']
            sl = slice(None)
        elif isinstance(seqs, slice):
            sl = seqs
            dev = {k: v[sl] for k, v in tb['dev'].items()}
            host = tb['host']
        else:
            idx = torch.as_tensor(np.asarray(seqs, dtype=np.int64), device=self.device)
            dev = {k: v.index_select(0, idx).contiguous() for k, v in tb['dev'].items()}
            host = tb['host']
            sl = np.asarray(seqs, dtype=np.int64)
        own_len = host['own_len'][sl]
        up_len = host['up_len'][sl]
        n_seq = len(own_len)
        N = int(own_len.sum() - n_seq * L.X_win)
        return dev, n_seq, N, int(own_len.sum()), int(up_len.sum())

    def evaluate(self, seqs=None, qU_ratio=None, stats_allreduce=None, world=1):
        """ELBO and all gradients for the sequences ``seqs`` (None = all, slice, or index list).  Returns the device
        scalar ELBO (also stored in ``grad_array[elbo_slot]``).  Latent gradients are written for the touched sequences
        only."""
        h = self.handle
        # chains started by the previous step of an encoder model (_restart_chains) are used instead of dropped
        keep = bool(getattr(self, '_use_chains', False)) and self._chain_ready and getattr(self, '_rows_key', None) is None \
            and stats_allreduce is None
        if not keep:
            self._flush_pipeline()
        ios, n_batch = self._build_ios(seqs, qU_ratio)
        if keep and any(lin.N > n for (lin, _), n in zip(ios, self._chain_n)):
            self._flush_pipeline()
            keep = False
        self._last_ios = ios
        try:
            return self._evaluate_ios(ios, n_batch, stats_allreduce, world, rows_only=keep)
        except Exception:
            h.discard_prepared()          # a failed call must not leave prepared layers queued for the next one
            self._chain_ready = False
            raise

    def _restart_chains(self):
        """After the global parameters of every layer have been updated on the library's side stream (adam_step(post=True)):
        start the next step's M x M chains now, so that they run under whatever the caller does before the next evaluation
        (an encoder's backward and forward passes).  Used by the encoder models' svi_step; DeepAutoreg's own pipelined step
        does the same per layer."""
        from ._lib import PREPARE_CHAIN_ONLY
        ios = getattr(self, '_last_ios', None)
        if not (self.svi and self.pipeline and ios is not None and len(ios) <= 3):
            return
        for lin, _ in ios:
            self.handle.layer_prepare(lin, PREPARE_CHAIN_ONLY)
        self._chain_ready = True
        self._chain_n = [lin.N for lin, _ in ios]

    def _evaluate_ios(self, ios, n_batch, stats_allreduce, world, rows_only=False):
        h = self.handle
        # start K_uu / factorisations / the data-independent SVI algebra of every layer now (side stream): they need only
        # the parameters, and would otherwise sit on the critical path in front of each layer's psi passes
        if self.svi and len(ios) <= 3:
            from ._lib import PREPARE_ROWS_ONLY
            for lin, _ in ios:
                h.layer_prepare(lin, PREPARE_ROWS_ONLY if rows_only else 0)
            self._chain_ready = False            # consumed by the layer steps below
        if (not self.svi) and stats_allreduce is not None:
            # data-parallel collapsed bound: one all-reduce of the sufficient statistics per layer (svi_vardtc.py:65-67)
            for (lin, lout), L in zip(ios, self.layers):
                stats = torch.empty(2 + 2 * L.M * L.M + L.D * L.M, dtype=torch.float64, device=self.device)
                ntot = torch.tensor([float(lin.N)], dtype=torch.float64, device=self.device)
                h.vardtc_stats(lin, stats)
                stats_allreduce(stats, ntot, L)
                h.vardtc_finish(lin, lout, stats, int(round(float(ntot.item()))), 1.0 / world)
        else:
            for lin, lout in ios:
                h.layer_step(lin, lout)
        h.join()                         # global gradients / ELBO of every layer are produced on the library's side streams
        elbo = self._elbo_buf.sum()
        self.grad_array[self.elbo_slot] = elbo
        self._last_n_batch = n_batch
        return elbo

    def _flush_pipeline(self):
        """Drop the early chains a pipelined svi_step left in flight (the parameters may have been changed since)."""
        if getattr(self, '_chain_ready', False):
            self.handle.discard_prepared()
            self._chain_ready = False
        self._rows_key = None

    def _build_ios(self, seqs, qU_ratio):
        """rgp_layer_in / rgp_layer_out of every layer (top first) for the sequences ``seqs``."""
        n_batch = None
        hist = getattr(self, '_keep_hist', None)            # tables of the last few calls stay alive: kernels of pipelined steps
        if hist is None:                                    # (and of a prefetched minibatch) may still read them
            hist = self._keep_hist = []
        self._keep = []
        hist.append(self._keep)
        del hist[:-6]
        ios = []
        for k, L in enumerate(self.layers):
            dev, n_seq, N, own_rows, up_rows = self._seq_tables(L, seqs)
            n_batch = n_seq
            i = L.index
            lin, lout = LayerIn(), LayerOut()
            lin.M, lin.D, lin.X_win, lin.X_dim, lin.U_win, lin.U_dim = L.M, L.D, L.X_win, L.X_dim, L.U_win, L.U_dim
            lin.n_seq, lin.mode, lin.y_uncertain = n_seq, self.mode, 0 if i == 0 else 1
            lin.N, lin.own_rows, lin.up_rows = N, own_rows, up_rows
            ratio = float(n_seq) / self.n_total if qU_ratio is None else float(qU_ratio)     # layers.py:328
            lin.qU_ratio = ratio if self.svi else 1.0
            lin.up_var_const = 1e-10                                                         # model.py:65
            self._keep.append(dev)
            lin.own_off, lin.own_len = dev['own_off'].data_ptr(), dev['own_len'].data_ptr()
            lin.up_off, lin.up_len = dev['up_off'].data_ptr(), dev['up_len'].data_ptr()
            if i == 0:
                lin.own_mean, lin.own_var = self.Ycat.data_ptr(), None
            else:
                lin.own_mean, lin.own_var = self.means.data_ptr(), self.vars.data_ptr()
            if i == self.nLayers - 1:
                lin.up_mean = self.Ucat.data_ptr() if self.Ucat is not None else None
                lin.up_var = None
            else:
                lin.up_mean, lin.up_var = self.means.data_ptr(), self.vars.data_ptr()
            for nm in ('Z', 'variance', 'inv_l', 'noise'):
                setattr(lin, nm, self.view(L, nm).data_ptr())
            for nm in ('gZ', 'gvariance', 'ginv_l', 'gnoise'):
                setattr(lout, nm, self.view(L, nm[1:], grad=True).data_ptr())
            if self.svi:
                for nm in ('qU_m', 'qU_W', 'qU_a'):
                    setattr(lin, nm, self.view(L, nm).data_ptr())
                    setattr(lout, 'g' + nm, self.view(L, nm, grad=True).data_ptr())
            lout.elbo = self._elbo_buf[k:k + 1].data_ptr()
            if i > 0:
                lout.g_own_mean, lout.g_own_var = self.gmeans.data_ptr(), self.gvars.data_ptr()
            if i < self.nLayers - 1:
                lout.g_up_mean, lout.g_up_var = self.gmeans.data_ptr(), self.gvars.data_ptr()
            lout.woodbury_inv, lout.woodbury_vector = self._winv[i].data_ptr(), self._wvec[i].data_ptr()
            ios.append((lin, lout))
        return ios, n_batch

    def log_likelihood(self, seqs=None):
        e = float(self.evaluate(seqs).item())
        self.handle.sync()
        return e

    # ------------------------------------------------------------------------------------------------------------
    # paramz-like flat access for tests / L-BFGS (host round trip; not the SVI hot loop)
    # ------------------------------------------------------------------------------------------------------------
    def _index_set(self, seqs=None):
        """Indices of param_array that belong to the globals and to the latents of ``seqs``."""
        if seqs is None:                      # every L-BFGS evaluation asks for this one: built (and copied to the device) once
            cached = getattr(self, '_index_all', None)
            if cached is None:
                cached = self._index_all = self._index_set(range(self.nSeq))
            return cached
        idx = [np.arange(self.n_global)]
        ss = range(self.nSeq) if seqs is None else seqs
        for s in ss:
            a, b = int(self.seq_loc_off[s]), int(self.seq_loc_off[s + 1])
            idx.append(self.loc0 + np.arange(a, b))
        for s in ss:
            a, b = int(self.seq_loc_off[s]), int(self.seq_loc_off[s + 1])
            idx.append(self.loc0 + self.n_local + np.arange(a, b))
        return torch.as_tensor(np.concatenate(idx), device=self.device)

    def get_optimizer_array(self, seqs=None):
        idx = self._index_set(seqs)
        p = self.param_array[idx]
        pos = self.positive[idx].bool()
        return torch.where(pos, _softplus_inv(p), p).cpu().numpy()

    def set_optimizer_array(self, x, seqs=None):
        self._flush_pipeline()
        idx = self._index_set(seqs)
        x = torch.as_tensor(np.asarray(x, dtype=np.float64), device=self.device)
        pos = self.positive[idx].bool()
        self.param_array[idx] = torch.where(pos, _softplus(x), x)

    def objective_and_grad(self, x, seqs=None):
        """(-ELBO, -dELBO/dx) in optimiser space on the host (paramz' objective_function / _grads).  With a data streamer
        attached and ``seqs`` None, each call first advances to the streamer's next minibatch (model.py:209-213); ``x`` then
        covers the globals and the latents of ALL sequences, gradients of untouched sequences are zero."""
        if seqs is None and self.data_streamer is not None:
            mb = self._next_minibatch()
            self.set_optimizer_array(x, None)
            self.grad_array[self.loc0:].zero_()
            e = self.evaluate(mb)
            idx = self._index_set(None)
            g = self.grad_array[idx]
            p = self.param_array[idx]
            pos = self.positive[idx].bool()
            g = torch.where(pos, g * (-torch.expm1(-p)), g)
            self.handle.sync()
            return -float(e.item()), -g.cpu().numpy()
        self.set_optimizer_array(x, seqs)
        e = self.evaluate(seqs)
        idx = self._index_set(seqs)
        g = self.grad_array[idx]
        p = self.param_array[idx]
        pos = self.positive[idx].bool()
        g = torch.where(pos, g * (-torch.expm1(-p)), g)
        self.handle.sync()
        return -float(e.item()), -g.cpu().numpy()

    def optimize(self, optimizer='bfgs', max_iters=100, messages=0):
        """``m.optimize('bfgs', max_iters=...)`` of the reference (paramz -> scipy L-BFGS-B), full batch."""
        from scipy.optimize import fmin_l_bfgs_b
        x0 = self.get_optimizer_array()
        x, f, d = fmin_l_bfgs_b(self.objective_and_grad, x0, maxfun=max_iters, maxiter=max_iters, iprint=1 if messages else -1)
        self.set_optimizer_array(x)
        return f, d

    # ------------------------------------------------------------------------------------------------------------
    # minibatch switch (model.py:209-282) and parameter-vector checkpoint (examples/walk_run_2_alex.py:565-576)
    # ------------------------------------------------------------------------------------------------------------
    def set_DataStreamer(self, data_streamer):
        """Attach a streamer (rgp_amd.data_streamers): every objective evaluation through ``objective_and_grad`` /
        ``optimize`` then sees the streamer's next minibatch, as the overridden ``optimizer_array`` setter does in the
        reference (model.py:209-213).  The streamer must index the sequences this model was built with."""
        assert data_streamer.total_size == self.nSeq, 'the streamer must cover the sequences of this model'
        self.data_streamer = data_streamer
        self.mb_inf_sample_idxes = list(range(self.nSeq))

    def _next_minibatch(self):
        _, idx, _, _ = self.data_streamer.next_minibatch()
        self.mb_inf_sample_idxes = [int(i) for i in idx]
        return self.mb_inf_sample_idxes

    def save(self, path):
        """Dump the flat constrained parameter vector (the reference saves ``m.param_array`` to .h5 / .mat)."""
        self.handle.sync()
        np.savez(path, param_array=self.param_array.cpu().numpy(), n_global=self.n_global, n_local=self.n_local)

    def load(self, path):
        d = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        assert int(d['n_global']) == self.n_global and int(d['n_local']) == self.n_local, 'parameter layout mismatch'
        self._flush_pipeline()
        self.param_array.copy_(torch.as_tensor(d['param_array'], device=self.device))
        self._adam = None

    # ------------------------------------------------------------------------------------------------------------
    # SVI loop: Adam on device (global bucket + the touched sequences' local parameters)
    # ------------------------------------------------------------------------------------------------------------
    def _init_adam(self):
        self.opt_array = torch.where(self.positive.bool(), _softplus_inv(self.param_array), self.param_array).contiguous()
        self._adam = dict(m=torch.zeros_like(self.param_array), v=torch.zeros_like(self.param_array),
                          t_glob=0, t_loc=np.zeros(self.nSeq, dtype=np.int64))

    def svi_step(self, seqs, lr=1e-2, allreduce=None, qU_ratio=None, update=True, lazy_elbo=False, next_seqs=None):
        """One SVI step on the sequences ``seqs`` (slice of contiguous indices or index list): ELBO + gradients,
        optional all-reduce of the global bucket (data-parallel), one Adam update of the global parameters and of the
        touched sequences' latent parameters.  Returns the device ELBO scalar (pre-update).

        lazy_elbo=True (pipelined steps only): the ELBO is summed on the library's side stream and the main stream does not
        wait for the last layer's M x M algebra, so the next step's row work starts earlier; the returned tensor may only be
        read after ``handle.join()`` / a device synchronisation.

        next_seqs (pipelined steps only): the minibatch of the FOLLOWING step when the data streamer already knows it.  If it
        shares no sequence with ``seqs`` its row terms are computed while this step still runs (RGP_PREPARE_ROWS_EARLY), so the
        next step does not start with the row preparation of its first layer; a hint that turns out wrong only costs the
        prefetch (the prepared layers are discarded)."""
        if isinstance(seqs, slice) and seqs.step not in (None, 1):
            seqs = list(range(*seqs.indices(self.nSeq)))          # a strided slice is an index list, not a contiguous range
        if self._adam is None:
            self._init_adam()
        ad, h = self._adam, self.handle
        pipelined = update and self.svi and self.pipeline and len(self.layers) <= 3 and type(self).evaluate is DeepAutoreg.evaluate \
            and not (allreduce is not None and self.single_bucket)
        if pipelined:
            try:
                elbo = self._svi_step_pipelined(seqs, lr, allreduce, qU_ratio, lazy_elbo, next_seqs)
            except Exception:
                h.discard_prepared()
                self._chain_ready = False
                self._rows_key = None
                raise
        else:
            elbo = self.evaluate(seqs, qU_ratio=qU_ratio)
            if allreduce is not None:
                allreduce(self.grad_array[:self.n_global + 1])
                elbo = self.grad_array[self.elbo_slot]
            if not update:              # debugging / kernel ablations only
                return elbo
            ad['t_glob'] += 1
            G = self.n_global
            h.adam_step(self.opt_array[:G], self.param_array[:G], self.grad_array[:G], ad['m'][:G], ad['v'][:G],
                        self.positive[:G], lr, ad['t_glob'])
        if isinstance(seqs, slice):
            r = range(*seqs.indices(self.nSeq))
            ranges = [(r.start, r.stop)] if len(r) else []
        elif seqs is None:
            ranges = [(0, self.nSeq)]
        else:
            ranges = [(int(s), int(s) + 1) for s in seqs]
        for (r0, r1) in ranges:
            ad['t_loc'][r0:r1] += 1
            # one launch per run of sequences with the same step count (the bias correction depends on it)
            s0 = r0
            while s0 < r1:
                t = int(ad['t_loc'][s0])
                s1 = s0 + 1
                while s1 < r1 and int(ad['t_loc'][s1]) == t:
                    s1 += 1
                a, b = int(self.seq_loc_off[s0]), int(self.seq_loc_off[s1])
                for base in (self.loc0, self.loc0 + self.n_local):
                    sl = slice(base + a, base + b)
                    h.adam_step(self.opt_array[sl], self.param_array[sl], self.grad_array[sl], ad['m'][sl], ad['v'][sl],
                                self.positive[sl], lr, t)
                s0 = s1
        return elbo

    @staticmethod
    def _seq_key(seqs, n):
        if seqs is None:
            return tuple(range(n))
        if isinstance(seqs, slice):
            return tuple(range(*seqs.indices(n)))
        return tuple(int(x) for x in seqs)

    def _svi_step_pipelined(self, seqs, lr, allreduce, qU_ratio, lazy_elbo=False, next_seqs=None):
        """ELBO + gradients + update of the GLOBAL parameters, layer by layer: as soon as a layer's gradients exist (library
        side stream) they are all-reduced and applied there, and the M x M chain of the NEXT step is started for that layer
        while the main stream is still busy with the psi passes of the layers below (include/rgp_hip.h, pipelining
        section).  Same arithmetic as evaluate + one Adam step; only the order of independent work changes."""
        from ._lib import PREPARE_CHAIN_ONLY, PREPARE_ROWS_ONLY, PREPARE_ROWS_EARLY
        ad, h = self._adam, self.handle
        key = self._seq_key(seqs, self.nSeq)
        rows_key = getattr(self, '_rows_key', None)
        if rows_key is not None and rows_key != (key, qU_ratio):
            self._flush_pipeline()                      # the prefetched rows belong to another minibatch
        ios, n_batch = self._build_ios(seqs, qU_ratio)
        n_prev = getattr(self, '_chain_n', None)
        if self._chain_ready and n_prev is not None and any(lin.N > n for (lin, _), n in zip(ios, n_prev)):
            self._flush_pipeline()                      # larger minibatch than the early chains were sized for
        if getattr(self, '_rows_key', None) is None:
            for lin, _ in ios:
                h.layer_prepare(lin, PREPARE_ROWS_ONLY if self._chain_ready else 0)
        self._rows_key = None
        ios_next = None
        if next_seqs is not None:
            nkey = self._seq_key(next_seqs, self.nSeq)
            if not set(nkey) & set(key):
                ios_next, _ = self._build_ios(next_seqs, qU_ratio)
                if any(ln.N > lc.N for (ln, _), (lc, _) in zip(ios_next, ios)):
                    ios_next = None                     # the chain-only prepare below is sized for this step's minibatch
        ad['t_glob'] += 1
        if (allreduce is not None or lazy_elbo) and self._post_stream is None:
            sp = h.get_stream(1)            # NULL = the default stream (RGP_NO_OVERLAP debugging runs on a handle made there)
            self._post_stream = torch.cuda.ExternalStream(sp, device=self.device) if sp else torch.cuda.default_stream(self.device)
        elbo_in_bucket = False
        done = []
        for (lin, lout), L in zip(ios, self.layers):
            h.layer_step(lin, lout)
            a = min(v[0] for v in L.slices.values())
            b = max(v[1] for v in L.slices.values())
            if allreduce is not None:
                with torch.cuda.stream(self._post_stream):
                    if L is self.layers[-1] and b == self.elbo_slot:
                        # the ELBO slot follows the last layer's parameters: it rides in that layer's all-reduce (every
                        # layer's ELBO term has been written on this stream by now), one collective fewer per step
                        self.grad_array[self.elbo_slot] = self._elbo_buf.sum()
                        allreduce(self.grad_array[a:b + 1])
                        elbo_in_bucket = True
                    else:
                        allreduce(self.grad_array[a:b])
            h.adam_step(self.opt_array[a:b], self.param_array[a:b], self.grad_array[a:b], ad['m'][a:b], ad['v'][a:b],
                        self.positive[a:b], lr, ad['t_glob'], post=True)
            h.layer_prepare(lin, PREPARE_CHAIN_ONLY)
            if ios_next is not None:                    # rows of the next minibatch, behind this layer's new chain
                h.layer_prepare(ios_next[len(done)][0], PREPARE_ROWS_ONLY | PREPARE_ROWS_EARLY)
            done.append(L)
        if ios_next is not None:
            self._rows_key = (nkey, qU_ratio)
        self._chain_ready = True
        self._chain_n = [lin.N for lin, _ in ios]
        self._last_n_batch = n_batch
        if lazy_elbo:
            # the per-layer ELBO terms are written on the side stream; summing them there keeps the main stream free of the
            # last layer's M x M algebra (the latent gradients it needs next come from main-stream kernels)
            with torch.cuda.stream(self._post_stream):
                if elbo_in_bucket:
                    return self.grad_array[self.elbo_slot].clone()
                elbo = self._elbo_buf.sum()
                if allreduce is not None:
                    allreduce(elbo)
                self.grad_array[self.elbo_slot] = elbo
            return elbo
        h.join()
        if elbo_in_bucket:
            return self.grad_array[self.elbo_slot].clone()
        elbo = self._elbo_buf.sum()
        if allreduce is not None:
            allreduce(elbo)
        self.grad_array[self.elbo_slot] = elbo
        return elbo

    # ------------------------------------------------------------------------------------------------------------
    # prediction: top-down free-run cascade with moment matching (model.py:725-740, layers.py:1537-1571)
    # ------------------------------------------------------------------------------------------------------------
    def freerun(self, U, init_Xs=None):
        """Free simulation for control signal(s) ``U`` ((T, D_u) array or a list of equal-length arrays), using the
        posterior of the last ``evaluate``/``log_likelihood`` call.  Returns, per layer (top first), a pair of device
        tensors (mean, variance) of shape (n_seq, X_win + steps, D); the last entry is the observed layer."""
        Us = [U] if isinstance(U, np.ndarray) else list(U)
        n_seq = len(Us)
        T_u = Us[0].shape[0]
        assert all(u.shape[0] == T_u for u in Us), 'batched free-run needs equal-length control sequences'
        steps = T_u - self.U_win
        con_m = torch.as_tensor(np.stack(Us), dtype=torch.float64, device=self.device).contiguous()
        con_v, con_rows, con_win, con_dim = None, T_u, self.U_win - 1, self.Du
        outs = []
        for L in self.layers:
            skip = con_win - L.U_win + 1
            im = iv = None
            if init_Xs is not None and init_Xs[L.index] is not None and L.X_win > 0:
                im = torch.as_tensor(np.stack([x[0][-L.X_win:] for x in init_Xs[L.index]]), dtype=torch.float64, device=self.device).contiguous()
                iv = torch.as_tensor(np.stack([x[1][-L.X_win:] for x in init_Xs[L.index]]), dtype=torch.float64, device=self.device).contiguous()
            Xm, Xv = self.handle.freerun(n_seq, L.M, L.D, L.X_win, L.U_win, L.U_dim, steps, self.view(L, 'Z'),
                                         self.view(L, 'variance'), self.view(L, 'inv_l'), self._winv[L.index],
                                         self._wvec[L.index], im, iv, con_m, con_v, 1e-10, con_rows, skip)
            outs.append((Xm, Xv))
            con_m, con_v, con_rows, con_win, con_dim = Xm, Xv, L.X_win + steps, L.X_win, L.D
        return outs

    # ------------------------------------------------------------------------------------------------------------
    # parity helpers
    # ------------------------------------------------------------------------------------------------------------
    def load_from_oracle(self, om):
        """Copy every parameter of an oracle.model.DeepAutoreg with the same structure (tests / smoke only)."""
        for L in self.layers:
            p = om.P[L.index]
            self.set_param(L, 'Z', p.Z)
            self.set_param(L, 'variance', p.variance)
            self.set_param(L, 'inv_l', p.inv_l)
            self.set_param(L, 'noise', p.noise)
            if self.svi:
                self.set_param(L, 'qU_m', p.qU_m)
                self.set_param(L, 'qU_W', p.qU_W)
                self.set_param(L, 'qU_a', p.qU_a)
        for i in range(1, self.nLayers):
            for s in range(self.nSeq):
                self.latent(i, s).copy_(torch.as_tensor(om.Xmean[i][s], device=self.device))
                self.latent(i, s, var=True).copy_(torch.as_tensor(om.Xvar[i][s], device=self.device))
        self._adam = None

    def grads_as_oracle_dict(self, seqs=None):
        G = {}
        for L in self.layers:
            for nm in L.slices:
                g = self.view(L, nm, grad=True).cpu().numpy()
                G[(L.index, nm)] = g if g.size > 1 else float(g.reshape(-1)[0])
        ss = range(self.nSeq) if seqs is None else seqs
        for i in range(1, self.nLayers):
            for s in ss:
                G[(i, 'Xmean', s)] = self.latent(i, s, grad=True).cpu().numpy()
                G[(i, 'Xvar', s)] = self.latent(i, s, var=True, grad=True).cpu().numpy()
        return G


DeepAutoreg_new = DeepAutoreg


class _EncoderParamsMixin(object):
    """Flat parameter access, checkpoints and L-BFGS for the models whose q(X) comes from a recognition network: the free
    parameters are the GP globals and the network's own parameters (paramz links both under the model, model.py:400-440
    and layers.py:272-296), not the latent rows, so the flat vector is [GP globals (optimiser space) | encoder parameters]."""

    def _enc_flat(self, grad=False):
        ps = list(self.encoder.parameters())
        if grad:
            return torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in ps])
        return torch.cat([q.detach().reshape(-1) for q in ps])

    def get_optimizer_array(self, seqs=None):
        G = self.n_global
        pg = self.param_array[:G]
        x = torch.where(self.positive[:G].bool(), _softplus_inv(pg), pg)
        return torch.cat((x, self._enc_flat())).cpu().numpy()

    def set_optimizer_array(self, x, seqs=None):
        self._flush_pipeline()
        G = self.n_global
        x = torch.as_tensor(np.asarray(x, dtype=np.float64), device=self.device)
        self.param_array[:G] = torch.where(self.positive[:G].bool(), _softplus(x[:G]), x[:G])
        o = G
        with torch.no_grad():
            for q in self.encoder.parameters():
                q.copy_(x[o:o + q.numel()].view_as(q))
                o += q.numel()
        assert o == x.numel(), 'parameter vector length mismatch'

    def objective_and_grad(self, x, seqs=None):
        """(-ELBO, -gradient) over [GP globals | encoder parameters]; with a data streamer attached and ``seqs`` None every
        call first advances to the streamer's next minibatch (model.py:577-578,609-615)."""
        if seqs is None and self.data_streamer is not None:
            seqs = self._next_minibatch()
        self.set_optimizer_array(x)
        e = self.evaluate(seqs)
        G = self.n_global
        g, pg = self.grad_array[:G], self.param_array[:G]
        g = torch.where(self.positive[:G].bool(), g * (-torch.expm1(-pg)), g)
        out = torch.cat((g, self._enc_flat(grad=True)))
        self.handle.sync()
        return -float(e.item()), -out.cpu().numpy()

    def save(self, path):
        """The GP parameter vector plus every tensor of the encoder's state (the reference's ``m.param_array`` holds both)."""
        self.handle.sync()
        enc = {'enc__' + k: v.detach().cpu().numpy() for k, v in self.encoder.state_dict().items()}
        np.savez(path, param_array=self.param_array.cpu().numpy(), n_global=self.n_global, n_local=self.n_local, **enc)

    def load(self, path):
        d = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        assert int(d['n_global']) == self.n_global and int(d['n_local']) == self.n_local, 'parameter layout mismatch'
        sd = self.encoder.state_dict()
        missing = [k for k in sd if 'enc__' + k not in d.files]
        assert not missing, 'checkpoint holds no encoder state for: %s' % ', '.join(missing)
        self._flush_pipeline()
        self.param_array.copy_(torch.as_tensor(d['param_array'], device=self.device))
        with torch.no_grad():
            for k, v in sd.items():
                a = torch.as_tensor(d['enc__' + k], device=v.device)
                assert a.shape == v.shape, 'encoder tensor %s has shape %s in the checkpoint, %s in the model' % (k, tuple(a.shape), tuple(v.shape))
                v.copy_(a)
        self._adam = None
        self._enc_opt = None


class DeepAutoreg_rnn(_EncoderParamsMixin, DeepAutoreg):
    """autoreg.DeepAutoreg_rnn (model.py:284-629): q(X) of every hidden layer is produced by a recurrent encoder from the
    raw sequences (equal lengths, stacked to (T, batch, D_y + D_u) as at model.py:429), so the latents are NOT free
    parameters; the trainable parameters are the GP globals and the encoder weights.  The GP layers run through
    librgp_hip.so exactly as in DeepAutoreg (latents with one row per raw time step); the encoder is PyTorch-ROCm fp64."""

    def __init__(self, wins, Y, U=None, U_win=1, nDims=None, num_inducing=10, kernels=None, rnn_type='rnn',
                 rnn_hidden_dims=None, rnn_bidirectional=False, inference_method='svi', mb_inf_tot_data_size=None,
                 device=None, seed=0, Zs=None, **kw):
        from .rnn_encoder import SeqEncoder
        Ys = [Y] if isinstance(Y, np.ndarray) else list(Y)
        Us = None if U is None else ([U] if isinstance(U, np.ndarray) else list(U))
        assert len(set(y.shape[0] for y in Ys)) == 1, 'the sequence encoder needs equal-length sequences (model.py:429)'
        DeepAutoreg.__init__(self, wins, Ys, U=Us, U_win=U_win, nDims=nDims, num_inducing=num_inducing, kernels=kernels,
                             init='Y', inference_method=inference_method, mb_inf_tot_data_size=mb_inf_tot_data_size,
                             device=device, seed=seed, Zs=Zs, latent_full=True, **kw)
        torch.manual_seed(seed)
        in_dims = list(self.nDims[:-1])
        in_dims[0] += (Us[0].shape[1] if Us is not None else 0)              # encode_include_input (model.py:409-412)
        hid = rnn_hidden_dims[0] if rnn_hidden_dims is not None else 10
        self.encoder = SeqEncoder(in_dims, self.nDims[1:], hid, rnn_type, rnn_bidirectional).to(self.device)
        for lay in self.encoder.layers:                                      # GRU time loops run as device scans
            lay.handle = self.handle
        raw = np.stack([np.hstack([y, u]) if Us is not None else y for y, u in zip(Ys, Us or Ys)], axis=1)
        self.enc_input = torch.as_tensor(raw, dtype=torch.float64, device=self.device)       # (T, n_seq, D_y + D_u)
        # flat offsets of (t, s, h) of hidden layer i inside the means / vars regions
        self._enc_index = {}
        for i in range(1, self.nLayers):
            H = self.nDims[i]
            T = int(self.lat_len[i][0])
            base = torch.as_tensor(self.lat_off[i], device=self.device).view(1, -1, 1)
            t = torch.arange(T, device=self.device).view(-1, 1, 1) * H
            hh = torch.arange(H, device=self.device).view(1, 1, -1)
            self._enc_index[i] = (base + t + hh)                                             # (T, n_seq, H)
        self._enc_out = None

    def encode(self, seqs=None):
        """Run the encoder on the sequences ``seqs`` and write q(X) of every hidden layer into the latent buffers."""
        idx = slice(None) if seqs is None else (seqs if isinstance(seqs, slice) else torch.as_tensor(list(seqs), device=self.device))
        means, variances = self.encoder(self.enc_input[:, idx, :])
        with torch.no_grad():
            for l, (m, v) in enumerate(zip(means, variances)):
                flat = self._enc_index[l + 1][:, idx, :].reshape(-1)
                self.means[flat] = m.reshape(-1)
                self.vars[flat] = v.reshape(-1)
        self._enc_out = (means, variances, idx)
        return means, variances

    def evaluate(self, seqs=None, qU_ratio=None, **kw):
        """ELBO and gradients: GP globals in grad_array, encoder weights in ``.grad`` (model.py:525-551)."""
        self.encode(seqs)
        means, variances, idx = self._enc_out
        elbo = DeepAutoreg.evaluate(self, seqs, qU_ratio=qU_ratio, **kw)
        self.encoder.zero_grad()
        outs, grads = [], []
        for l, (m, v) in enumerate(zip(means, variances)):
            flat = self._enc_index[l + 1][:, idx, :].reshape(-1)
            outs += [m, v]
            grads += [self.gmeans[flat].view(m.shape), self.gvars[flat].view(v.shape)]
        torch.autograd.backward(outs, grads)                                               # rnn_encoder.py:274
        self._enc_out = None
        return elbo

    def encoder_grad_vector(self):
        return torch.cat([p.grad.reshape(-1) for p in self.encoder.parameters()])

    def svi_step(self, seqs, lr=1e-2, allreduce=None, qU_ratio=None, update=True):
        """One SVI step: ELBO + gradients, optional all-reduce (global GP bucket and encoder-weight gradients, both
        replicated parameters), Adam on the GP globals (device kernel) and on the encoder weights (torch.optim.Adam)."""
        if self._adam is None:
            self._init_adam()
            self._enc_opt = torch.optim.Adam(self.encoder.parameters(), lr=lr, maximize=True)
        ad, h = self._adam, self.handle
        carry = allreduce is None and update        # single process: the next step's chains start right after the update
        self._use_chains = carry
        try:
            elbo = self.evaluate(seqs, qU_ratio=qU_ratio)
        finally:
            self._use_chains = False
        if allreduce is not None:
            allreduce(self.grad_array[:self.n_global + 1])
            flat = self.encoder_grad_vector()
            allreduce(flat)
            o = 0
            for p in self.encoder.parameters():
                p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            elbo = self.grad_array[self.elbo_slot]
        if not update:
            return elbo
        ad['t_glob'] += 1
        G = self.n_global
        h.adam_step(self.opt_array[:G], self.param_array[:G], self.grad_array[:G], ad['m'][:G], ad['v'][:G],
                    self.positive[:G], lr, ad['t_glob'], post=carry)
        if carry:
            self._restart_chains()
        for g in self._enc_opt.param_groups:
            g['lr'] = lr
        self._enc_opt.step()
        return elbo


class DeepAutoreg_mlp(_EncoderParamsMixin, DeepAutoreg):
    """autoreg.DeepAutoreg(..., back_cstr=True) (model.py:30-187 with layers.py:398-441,623-756): the predicted latent means
    of every hidden layer are the output of an MLP recursion over the layer's own past and the signal above; trainable are
    the GP globals, the first X_win latent rows, the latent variances and the MLP weights.  The GP layers run through
    librgp_hip.so exactly as in DeepAutoreg; the recursion and its back-propagation are PyTorch-ROCm fp64
    (rgp_amd/mlp_encoder.py).  Sequences must have equal length (they are batched through the recursion)."""

    def __init__(self, wins, Y, U=None, U_win=1, nDims=None, X_variance=0.01, num_inducing=10, kernels=None,
                 MLP_dims=None, inference_method='svi', mb_inf_tot_data_size=None, device=None, seed=0, Zs=None,
                 minibatch_inference=False, mb_inf_init_xs_means='all', mb_inf_init_xs_vars='all', mb_inf_sample_idxes=None,
                 init_MLP_dims=None, **kw):
        """minibatch_inference / mb_inf_init_xs_means / mb_inf_init_xs_vars: the reference's parameterisations of the
        first rows and of the predicted rows' variances under minibatch inference (model.py:31-33,83-88; described at
        rgp_amd.mlp_encoder.MlpBackConstraint).  All sequences of the data set are given to the constructor and stay on the
        device; ``mb_inf_sample_idxes`` names the first minibatch, a data streamer the following ones."""
        from .mlp_encoder import MlpBackConstraint
        Ys = [Y] if isinstance(Y, np.ndarray) else list(Y)
        Us = None if U is None else ([U] if isinstance(U, np.ndarray) else list(U))
        assert len(set(y.shape[0] for y in Ys)) == 1, 'the batched recursion needs equal-length sequences'
        DeepAutoreg.__init__(self, wins, Ys, U=Us, U_win=U_win, nDims=nDims, X_variance=X_variance, num_inducing=num_inducing,
                             kernels=kernels, init='Y', inference_method=inference_method,
                             mb_inf_tot_data_size=mb_inf_tot_data_size, device=device, seed=seed, Zs=Zs, **kw)
        torch.manual_seed(seed)
        specs = []
        for i in range(self.nLayers - 1, 0, -1):                                 # top hidden layer first
            L = self.layer(i)
            up_len = int(self.u_len[0]) if (i == self.nLayers - 1 and Us is not None) else \
                (int(self.lat_len[i + 1][0]) if i < self.nLayers - 1 else 0)
            specs.append(dict(index=i, X_win=L.X_win, X_dim=L.X_dim, U_win=L.U_win, U_dim=L.U_dim,
                              N=int(self.lat_len[i][0]) - L.X_win, up_len=up_len))
        low_init = None
        if minibatch_inference and mb_inf_init_xs_means == 'mlp':               # model.py:167: first rows of the shifted observations
            xw1 = self.layer(1).X_win
            shift = self.U_win if Us is not None else 0                         # the model's data shift (model.py:53-69)
            low_init = np.stack([np.asarray(y, dtype=np.float64)[shift:shift + xw1].reshape(-1) for y in Ys])
        self.encoder = MlpBackConstraint(specs, self.nSeq, X_variance=X_variance, MLP_dims=MLP_dims,
                                         minibatch_inference=minibatch_inference, init_xs_means=mb_inf_init_xs_means,
                                         init_xs_vars=mb_inf_init_xs_vars, low_init=low_init,
                                         init_MLP_dims=init_MLP_dims).to(self.device)
        self.minibatch_inference = bool(minibatch_inference)
        if mb_inf_sample_idxes is not None:
            self.mb_inf_sample_idxes = [int(i) for i in mb_inf_sample_idxes]
        self.encoder.handle = self.handle                                        # the recursion runs as a device scan
        self.controls = None if Us is None else self.Ucat.view(self.nSeq, int(self.u_len[0]), self.Du)
        self._enc_index = {}
        for sp in specs:
            i, H = sp['index'], self.nDims[sp['index']]
            T = int(self.lat_len[i][0])
            base = torch.as_tensor(self.lat_off[i], device=self.device).view(-1, 1, 1)
            t = torch.arange(T, device=self.device).view(1, -1, 1) * H
            hh = torch.arange(H, device=self.device).view(1, 1, -1)
            self._enc_index[i] = base + t + hh                                    # (n_seq, T, H)
        self._enc_out = None
        self._enc_opt = None

    def encode(self, seqs=None):
        """Run the recursions for the sequences ``seqs`` and write q(X) of every hidden layer into the latent buffers."""
        idx = slice(None) if seqs is None else (seqs if isinstance(seqs, slice) else torch.as_tensor(list(seqs), device=self.device))
        means, variances = self.encoder(self.controls, idx)
        with torch.no_grad():
            for sp, m, v in zip(self.encoder.specs, means, variances):
                flat = self._enc_index[sp['index']][idx].reshape(-1)
                self.means[flat] = m.reshape(-1)
                self.vars[flat] = v.reshape(-1)
        self._enc_out = (means, variances, idx)
        return means, variances

    def evaluate(self, seqs=None, qU_ratio=None, **kw):
        """ELBO and gradients: GP globals in grad_array, recursion parameters in ``.grad`` (layers.py:668-756)."""
        self.encode(seqs)
        means, variances, idx = self._enc_out
        elbo = DeepAutoreg.evaluate(self, seqs, qU_ratio=qU_ratio, **kw)
        self.encoder.zero_grad()
        outs, grads = [], []
        for sp, m, v in zip(self.encoder.specs, means, variances):
            flat = self._enc_index[sp['index']][idx].reshape(-1)
            outs += [m, v]
            grads += [self.gmeans[flat].view(m.shape), self.gvars[flat].view(v.shape)]
        torch.autograd.backward(outs, grads)
        self._enc_out = None
        return elbo

    def encoder_grad_vector(self):
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.encoder.parameters()])

    def svi_step(self, seqs, lr=1e-2, allreduce=None, qU_ratio=None, update=True):
        """One SVI step: ELBO + gradients, optional all-reduce of the replicated parameters' gradients (GP globals and MLP
        weights; the per-sequence latent parameters stay local), Adam on the GP globals (device kernel) and torch Adam on
        the recursion parameters."""
        from .mlp_encoder import RowAdam
        if self._adam is None:
            self._init_adam()
        if self._enc_opt is None:
            self._enc_opt = RowAdam(self.encoder)
        ad, h = self._adam, self.handle
        carry = allreduce is None and update        # single process: the next step's chains start right after the update
        self._use_chains = carry
        try:
            elbo = self.evaluate(seqs, qU_ratio=qU_ratio)
        finally:
            self._use_chains = False
        if allreduce is not None:
            allreduce(self.grad_array[:self.n_global + 1])
            for p in self.encoder.shared_parameters():
                allreduce(p.grad)
            elbo = self.grad_array[self.elbo_slot]
        if not update:
            return elbo
        ad['t_glob'] += 1
        G = self.n_global
        h.adam_step(self.opt_array[:G], self.param_array[:G], self.grad_array[:G], ad['m'][:G], ad['v'][:G],
                    self.positive[:G], lr, ad['t_glob'], post=carry)
        if carry:
            self._restart_chains()
        idx = slice(None) if seqs is None else (seqs if isinstance(seqs, slice) else torch.as_tensor(list(seqs), device=self.device))
        self._enc_opt.step(idx, lr)
        return elbo
