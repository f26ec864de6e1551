"""ORACLE (test infrastructure only) -- layer / model assembly of the recurrent GP in NumPy.

The reference's ``autoreg/layers.py`` and ``autoreg/model.py`` cannot be imported here (they need
GPy.core.SparseGP, paramz, GPy.kern, GPy.likelihoods), so this file RESTATES their semantics:

  * data shift, controls as N(U, 1e-10)             /root/reference/autoreg/model.py:657-673 (new: :53-69)
  * hidden q(X) initialisation                       model.py:691-716
  * layer wiring (top layer first)                   model.py:680-689
  * lag-window design matrix (right aligned)         layers.py:1399-1431, util.py:6-12
  * per-layer bound + gradient routing               layers.py:66-176
  * dL/dq(X)                                         layers.py:1452-1458
  * output-side gradients, entropy / prior terms     layers.py:1460-1481, variational.py:4-25
  * window scatter-add (exact adjoint of the gather) layers.py:1433-1450
  * ELBO = sum over layers, bottom-up scatter        model.py:718-723 (new: :158-187)
  * q(U) initialisation                              layers.py:51-58, util.py:32-38
  * minibatch qU_ratio                               layers.py:231-232,328-329
  * free-run with moment matching                    layers.py:1537-1571, model.py:725-740
    (+ GPy SparseGP._raw_predict, restated from upstream and pinned by quadrature in tests/test_oracle_layers_golden.py)

What pins it: the inference pieces it composes are pinned to the reference by tests/test_oracle_golden.py;
the window indexing by hand-computed cases and the exact-adjoint/finite-difference check in
tests/test_oracle_model.py; additivity over minibatches (testing/minibatch_tests.py:288-296) likewise.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy.linalg import lstsq

from . import inference as oi
from . import psi_stats as ps


def conv_1d(arr, win):
    """Rows n -> arr[n:n+win] flattened time-major then dimension (util.py:6-12 + reshape at layers.py:1406)."""
    T = arr.shape[0]
    n = T - win + 1
    idx = np.arange(n)[:, None] + np.arange(win)[None, :]
    return arr[idx].reshape(n, -1)


def softplus(x):
    return np.where(x > 36.0, x, np.log1p(np.exp(np.minimum(x, 36.0))))


def softplus_inv(p):
    return np.where(p > 36.0, p, np.log(np.expm1(np.minimum(p, 36.0))))


class LayerParams(object):
    """Global parameters of one sparse-GP layer (names follow GPy's printed names)."""

    def __init__(self, Z, variance, inv_l, noise, svi, D):
        M = Z.shape[0]
        self.Z = Z.copy()
        self.variance = float(variance)
        self.inv_l = np.array(inv_l, dtype=float).copy()
        self.noise = float(noise)
        self.svi = svi
        if svi:
            self.qU_m = np.zeros((M, D))
            self.qU_W = np.zeros((M, M))
            self.qU_a = 1e-3


class DeepAutoreg(object):
    """NumPy restatement of autoreg.DeepAutoreg / DeepAutoreg_new without back-constraints.

    Index convention: ``i`` is the index into ``wins`` (0 = observed layer, len(wins)-1 = top layer fed by the
    controls).  ``self.order`` lists layer indices top first, which is the order of the reference's
    ``model.layers`` list (model.py:682-688)."""

    def __init__(self, wins, Y, U=None, U_win=1, nDims=None, X_variance=0.01, num_inducing=10,
                 init='Y', inference_method=None, n_total=None, Zs=None, rng=None, fast_psi=True, chunk=512,
                 latent_full=False):
        rng = np.random.default_rng(0) if rng is None else rng
        Ys = [Y] if isinstance(Y, np.ndarray) else list(Y)
        Us = None if U is None else ([U] if isinstance(U, np.ndarray) else list(U))
        assert wins[0] == 0, "the reference cannot run an autoregressive window on the observed layer"
        self.wins = list(wins)
        self.nLayers = len(wins)
        self.U_win = U_win
        self.svi = (inference_method == 'svi')
        self.nDims = list(nDims) if nDims is not None else [Ys[0].shape[1]] + [1] * (self.nLayers - 1)
        self.fast_psi, self.chunk = fast_psi, chunk
        # ---- a1: shift (model.py:657-673) ----
        if Us is not None:
            assert len(Us) == len(Ys)
            self.Us = [u[:-1].copy() for u in Us]
            self.Ys = [y[U_win:].copy() for y in Ys]
        else:
            self.Us, self.Ys = None, [y.copy() for y in Ys]
        self.nSeq = len(self.Ys)
        self.n_total = self.nSeq if n_total is None else n_total
        # ---- a2: hidden q(X) (model.py:691-716).  latent_full: latents produced by a sequence encoder have one row per
        #      RAW time step (DeepAutoreg_rnn, model.py:427-440) instead of wins[i] + len(Y) rows; windows stay right aligned ----
        self.latent_full = latent_full
        self.Xmean, self.Xvar = {}, {}
        for i in range(1, self.nLayers):
            ms, vs = [], []
            for y in self.Ys:
                w, h = self.wins[i], self.nDims[i]
                if latent_full:
                    w = (U_win if Us is not None else 0)
                m = np.zeros((w + y.shape[0], h))
                if init == 'Y':
                    m[w:] = y[:, :h]
                elif init == 'rand':
                    m[:] = rng.standard_normal(m.shape)
                elif init == 'zero':
                    m[:] = rng.standard_normal(m.shape) * 0.01
                else:
                    raise ValueError(init)
                ms.append(m)
                vs.append(np.zeros_like(m) + X_variance)
            self.Xmean[i], self.Xvar[i] = ms, vs
        # ---- layers, top first (model.py:680-689); Z by the randn branch (layers.py:1339) ----
        if not isinstance(num_inducing, (list, tuple)):
            num_inducing = [num_inducing] * self.nLayers
        self.order = list(range(self.nLayers - 1, -1, -1))
        self.P = {}
        for i in self.order:
            Xm, Xv, Ym, Yv = self.build_XY(i, range(self.nSeq))
            Q = Xm.shape[1]
            Z = rng.standard_normal((num_inducing[i], Q)) if Zs is None else np.array(Zs[i], dtype=float)
            noise = 1.0 if i == 0 else 0.01                              # model.py:684-688
            p = LayerParams(Z, 1.0, np.ones(Q), noise, self.svi, Ym.shape[1])
            if self.svi:                                                 # a15 (layers.py:51-58)
                W = lstsq(Xm, Ym)[0]
                p.qU_m = Z.dot(W)
                p.qU_W = rng.standard_normal((Z.shape[0], Z.shape[0])) * 0.01
                p.qU_a = 1e-3
            self.P[i] = p

    # ------------------------------------------------------------------------------------------
    # windows (a3/a4, Appendix C)
    # ------------------------------------------------------------------------------------------
    def _inputs_of(self, i, s):
        """(mean, var, win) of what feeds layer i for sequence s: controls for the top layer, else the latent above."""
        if i == self.nLayers - 1:
            if self.Us is None:
                return None
            u = self.Us[s]
            return u, np.full(u.shape, 1e-10), self.U_win                # model.py:669
        return self.Xmean[i + 1][s], self.Xvar[i + 1][s], self.wins[i + 1]

    def _own_len(self, i, s):
        return self.Ys[s].shape[0] if i == 0 else self.Xmean[i][s].shape[0]

    def build_XY(self, i, seqs):
        X_win = self.wins[i]
        Xm, Xv, Ym, Yv = [], [], [], []
        for s in seqs:
            N = self._own_len(i, s) - X_win
            cols_m, cols_v = [], []
            if X_win > 0:                                                # own history (layers.py:1405-1407)
                cols_m.append(conv_1d(self.Xmean[i][s][:-1], X_win))
                cols_v.append(conv_1d(self.Xvar[i][s][:-1], X_win))
            inp = self._inputs_of(i, s)
            if inp is not None:                                          # inputs (layers.py:1408-1410)
                um, uv, uw = inp
                cols_m.append(conv_1d(um[um.shape[0] - N - uw + 1:], uw))
                cols_v.append(conv_1d(uv[uv.shape[0] - N - uw + 1:], uw))
            Xm.append(np.hstack(cols_m))
            Xv.append(np.hstack(cols_v))
            assert Xm[-1].shape[0] == N
            if i == 0:
                Ym.append(self.Ys[s])
            else:
                Ym.append(self.Xmean[i][s][X_win:])
                Yv.append(self.Xvar[i][s][X_win:])
        return (np.vstack(Xm), np.vstack(Xv), np.vstack(Ym), (np.vstack(Yv) if i > 0 else None))

    # ------------------------------------------------------------------------------------------
    # one objective + gradient evaluation (a5-a13)
    # ------------------------------------------------------------------------------------------
    def elbo_and_grads(self, seqs=None, want_grads=True):
        seqs = list(range(self.nSeq)) if seqs is None else list(seqs)
        ratio = float(len(seqs)) / self.n_total if self.svi else 1.0     # layers.py:328
        elbo = 0.0
        G = {}
        gXm = {i: {s: np.zeros_like(self.Xmean[i][s]) for s in seqs} for i in range(1, self.nLayers)}
        gXv = {i: {s: np.zeros_like(self.Xvar[i][s]) for s in seqs} for i in range(1, self.nLayers)}
        self.layer_elbo = {}
        for i in self.order:
            p = self.P[i]
            X_win = self.wins[i]
            Xm, Xv, Ym, Yv = self.build_XY(i, seqs)
            N, Q = Xm.shape
            D = Ym.shape[1]
            kern = ps.RBF(Q, p.variance, p.inv_l, fast=self.fast_psi, chunk=self.chunk)
            q = _Post(Xm, Xv)
            st = oi.Stats(kern.psi0(p.Z, q).sum(), kern.psi1(p.Z, q), kern.psi2(p.Z, q), Ym, Yv)
            Kuu = kern.K(p.Z)
            if self.svi:
                S = p.qU_W.dot(p.qU_W.T) + np.eye(p.Z.shape[0]) * p.qU_a                 # layers.py:71
                logL, g, winv, wvec, mid = oi.svi(Kuu, st, p.noise, p.qU_m, S)
                KL, dKLm, dKLv, dKLK = oi.kl_qU(p.qU_m, S, mid)
                logL = logL - KL * ratio                                                  # layers.py:76-79
                g['dL_dqU_mean'] = g['dL_dqU_mean'] - dKLm * ratio
                g['dL_dqU_var'] = g['dL_dqU_var'] - dKLv * ratio
                g['dL_dKmm'] = g['dL_dKmm'] - dKLK * ratio
            else:
                logL, g, winv, wvec = oi.vardtc(Kuu, st, p.noise)
            p.woodbury_inv, p.woodbury_vector = winv, wvec
            bound = float(logL)
            if i > 0:                                                     # a11 (layers.py:1466-1481)
                off = 0
                for s in seqs:
                    n_s = self._own_len(i, s) - X_win
                    m_s, v_s = self.Xmean[i][s], self.Xvar[i][s]
                    if X_win > 0:
                        bound += -oi.normal_prior_kl(m_s[:X_win], v_s[:X_win])
                    bound += -oi.normal_entropy(v_s[X_win:])
                    if want_grads:
                        gXm[i][s][X_win:] += g['dL_dYmean'][off:off + n_s]
                        dyv = g['dL_dYvar'][off:off + n_s]
                        gXv[i][s][X_win:] += dyv[:, None] if dyv.ndim == 1 else dyv     # layers.py:603-606
                        if X_win > 0:
                            dm, dv = oi.normal_prior_grads(m_s[:X_win], v_s[:X_win])
                            gXm[i][s][:X_win] += dm
                            gXv[i][s][:X_win] += dv
                        gXv[i][s][X_win:] += oi.normal_entropy_grad_var(v_s[X_win:])
                    off += n_s
            self.layer_elbo[i] = bound
            elbo += bound
            if not want_grads:
                continue
            # ---- a9: kernel / Z / likelihood / q(U) gradients (layers.py:83-176) ----
            pg = kern.psi_grads(p.Z, q, g['dL_dpsi0'], g['dL_dpsi1'], g['dL_dpsi2'])
            kv, kl, kZ = kern.K_grads(p.Z, g['dL_dKmm'])
            G[(i, 'Z')] = pg['dZ'] + kZ
            G[(i, 'variance')] = pg['dvar'] + kv
            G[(i, 'inv_l')] = pg['dinv_l'] + kl
            G[(i, 'noise')] = float(g['dL_dthetaL'])
            if self.svi:
                Gv = g['dL_dqU_var']
                G[(i, 'qU_m')] = g['dL_dqU_mean']
                G[(i, 'qU_W')] = (Gv + Gv.T).dot(p.qU_W)                  # layers.py:175
                G[(i, 'qU_a')] = float(np.trace(Gv))                      # layers.py:176
            # ---- a10 + a12: dL/dq(X) scattered back onto the latents (layers.py:1433-1458) ----
            off = 0
            Qx = X_win * self.nDims[i] if X_win > 0 else 0
            # reference_scatter: the literal per-row Python loop of the reference (CPU-baseline timing of its structure)
            sc = scatter_reference_loop if getattr(self, 'reference_scatter', False) else _scatter
            for s in seqs:
                n_s = self._own_len(i, s) - X_win
                dm, dv = pg['dmu'][off:off + n_s], pg['dS'][off:off + n_s]
                if X_win > 0:
                    sc(gXm[i][s], dm[:, :Qx], 0, X_win)
                    sc(gXv[i][s], dv[:, :Qx], 0, X_win)
                if i < self.nLayers - 1:
                    uw = self.wins[i + 1]
                    T_u = self.Xmean[i + 1][s].shape[0]
                    u_off = T_u - n_s - uw + 1                            # layers.py:1439
                    sc(gXm[i + 1][s], dm[:, Qx:], u_off, uw)
                    sc(gXv[i + 1][s], dv[:, Qx:], u_off, uw)
                off += n_s
        self.last_elbo = elbo
        if not want_grads:
            return elbo, None
        for i in range(1, self.nLayers):
            for s in seqs:
                G[(i, 'Xmean', s)] = gXm[i][s]
                G[(i, 'Xvar', s)] = gXv[i][s]
        return elbo, G

    # ------------------------------------------------------------------------------------------
    # flat parameter vector (paramz-like): positive parameters go through softplus (Logexp)
    # ------------------------------------------------------------------------------------------
    def param_list(self, seqs=None):
        seqs = list(range(self.nSeq)) if seqs is None else list(seqs)
        out = []
        for i in self.order:
            out += [(i, 'Z'), (i, 'variance'), (i, 'inv_l'), (i, 'noise')]
            if self.svi:
                out += [(i, 'qU_m'), (i, 'qU_W'), (i, 'qU_a')]
        for i in range(self.nLayers - 1, 0, -1):
            for s in seqs:
                out += [(i, 'Xmean', s), (i, 'Xvar', s)]
        return out

    POSITIVE = ('variance', 'inv_l', 'noise', 'qU_a', 'Xvar')

    def get(self, key):
        i, name = key[0], key[1]
        if name == 'Xmean':
            return self.Xmean[i][key[2]]
        if name == 'Xvar':
            return self.Xvar[i][key[2]]
        return np.asarray(getattr(self.P[i], name), dtype=float)

    def set(self, key, val):
        i, name = key[0], key[1]
        if name == 'Xmean':
            self.Xmean[i][key[2]][...] = val.reshape(self.Xmean[i][key[2]].shape)
        elif name == 'Xvar':
            self.Xvar[i][key[2]][...] = val.reshape(self.Xvar[i][key[2]].shape)
        elif name in ('variance', 'noise', 'qU_a'):
            setattr(self.P[i], name, float(np.asarray(val).reshape(-1)[0]))
        else:
            getattr(self.P[i], name)[...] = val.reshape(getattr(self.P[i], name).shape)

    def get_param_array(self, seqs=None):
        return np.concatenate([self.get(k).ravel() for k in self.param_list(seqs)])

    def set_param_array(self, x, seqs=None):
        o = 0
        for k in self.param_list(seqs):
            n = self.get(k).size
            self.set(k, np.asarray(x[o:o + n]))
            o += n

    def get_optimizer_array(self, seqs=None):
        return np.concatenate([(softplus_inv(self.get(k)) if k[1] in self.POSITIVE else self.get(k)).ravel()
                               for k in self.param_list(seqs)])

    def set_optimizer_array(self, x, seqs=None):
        o = 0
        for k in self.param_list(seqs):
            n = self.get(k).size
            v = np.asarray(x[o:o + n])
            self.set(k, softplus(v) if k[1] in self.POSITIVE else v)
            o += n

    def flat_grads(self, G, seqs=None, optimizer_space=False):
        out = []
        for k in self.param_list(seqs):
            g = np.asarray(G[k], dtype=float).ravel()
            if optimizer_space and k[1] in self.POSITIVE:
                p = self.get(k).ravel()
                g = g * (-np.expm1(-p))                     # d softplus(x)/dx = 1 - exp(-p)
            out.append(g)
        return np.concatenate(out)

    def objective_and_grad(self, x, seqs=None):
        self.set_optimizer_array(x, seqs)
        e, G = self.elbo_and_grads(seqs)
        return -e, -self.flat_grads(G, seqs, optimizer_space=True)

    # ------------------------------------------------------------------------------------------
    # prediction (a14): GPy SparseGP._raw_predict restated + moment-matching free-run
    # ------------------------------------------------------------------------------------------
    def _raw_predict(self, i, xm, xv):
        p = self.P[i]
        kern = ps.RBF(xm.shape[1], p.variance, p.inv_l, fast=False)
        q = _Post(xm, xv)
        psi1 = kern.psi1(p.Z, q)                                          # (n,M)
        psi2n = kern.psi2n(p.Z, q)                                        # (n,M,M)
        mean = psi1.dot(p.woodbury_vector)                                # (n,D)
        wv = p.woodbury_vector
        t = psi2n - psi1[:, :, None] * psi1[:, None, :]
        var = np.einsum('md,nmk,kd->nd', wv, t, wv) + (p.variance - np.einsum('mk,nmk->n', p.woodbury_inv, psi2n))[:, None]
        return mean, np.clip(var, 1e-15, np.inf)

    def layer_freerun(self, i, step, U_mean=None, U_var=None, init=None):
        """Moment-matching branch of Layer.freerun (layers.py:1551-1571)."""
        X_win, X_dim = self.wins[i], self.nDims[i]
        has_in = U_mean is not None
        uw = (self.U_win if i == self.nLayers - 1 else self.wins[i + 1]) if has_in else 0
        Qx = X_win * X_dim
        Q = Qx + (uw * U_mean.shape[1] if has_in else 0)
        Xm = np.empty((X_win + step, X_dim))
        Xv = np.ones((X_win + step, X_dim))
        if X_win > 0:
            if init is None:
                Xm[:X_win], Xv[:X_win] = 0.0, 1.0                          # layers.py:1545
            else:
                Xm[:X_win], Xv[:X_win] = init[0][-X_win:], init[1][-X_win:]
        xin_m, xin_v = np.empty((1, Q)), np.full((1, Q), 1e-10)
        for n in range(step):
            if X_win > 0:
                xin_m[0, :Qx] = Xm[n:n + X_win].ravel()
                xin_v[0, :Qx] = Xv[n:n + X_win].ravel()
            if has_in:
                xin_m[0, Qx:] = U_mean[n:n + uw].ravel()
                if U_var is not None:
                    xin_v[0, Qx:] = U_var[n:n + uw].ravel()
            m, v = self._raw_predict(i, xin_m, xin_v)
            v = np.where(v <= 0.0, 1e-10, v)                              # layers.py:1570
            Xm[X_win + n], Xv[X_win + n] = m[0], v[0]
        return Xm, Xv

    def freerun(self, U, init_Xs=None):
        """Top-down cascade (model.py:725-740). Returns list of (mean, var) per layer, top first."""
        step = U.shape[0] - self.U_win
        con_m, con_v = U, None
        con_win = self.U_win - 1
        outs = []
        for i in self.order:
            uw = self.U_win if i == self.nLayers - 1 else self.wins[i + 1]
            con_m = con_m[con_win - uw + 1:]
            con_v = None if con_v is None else con_v[con_win - uw + 1:]
            Xm, Xv = self.layer_freerun(i, step, con_m, con_v, None if init_Xs is None else init_Xs[i])
            outs.append((Xm, Xv))
            con_m, con_v, con_win = Xm, Xv, self.wins[i]
        return outs


class _Post(object):
    def __init__(self, mean, variance):
        self.mean, self.variance = mean, variance


def _scatter(target, rows, offset, win):
    """target[offset+n : offset+n+win] += rows[n].reshape(win, dim) for every n (layers.py:1443-1448),
    written as ``win`` shifted slice-adds so the summation order is fixed."""
    n, dim = rows.shape[0], target.shape[1]
    r = rows.reshape(n, win, dim)
    for j in range(win):
        target[offset + j: offset + j + n] += r[:, j, :]


def scatter_reference_loop(target, rows, offset, win):
    """Literal per-row loop of the reference (layers.py:1440-1448); used by tests and the 'reference-faithful' CPU timing."""
    dim = target.shape[1]
    for n in range(rows.shape[0]):
        target[offset + n: offset + n + win] += rows[n].reshape(-1, dim)
