#!/usr/bin/env python3
"""Generate tests/golden/layer_*.npz by running the REFERENCE's own layer code (autoreg/layers.py, autoreg/util.py) on a
two-hidden-layer toy model in the build container.

    python tests/golden/make_golden_layers.py          (build container only: /root/reference is not on the GPU box)

What runs unmodified from the reference, as unbound functions on duck-typed layer objects (the classes themselves need
GPy's SparseGP / paramz parameter tree, which is not installed):
    Layer(_new)._update_conv / _init_XY / _update_X      window construction           (layers.py:1399-1431, :476-553)
    SparseGP_MPI._inference_vardtc                        bound incl. KL * qU_ratio, gradient routing (layers.py:66-176)
    Layer(_new)._update_qX_gradients / _prepare_gradients  dL/dq(X), output-side terms, entropy / prior (layers.py:1452-1481)
    Layer(_new).update_latent_gradients                   window scatter-add, bottom-up  (layers.py:1433-1450, model.py:723)
    VarDTC / SVI_VarDTC (inference) and NormalEntropy / NormalPrior (variational), as in make_golden.py
What is NOT the reference's: the kernel object.  GPy is absent, so kern.K / psi* / their gradients come from
oracle/psi_stats.py behind GPy's method names (pinned separately by quadrature and finite differences).  The fixtures
therefore pin windows, row order, bound assembly, gradient routing and scatter -- everything of the layer / model code
around the kernel.  No reference source text is stored: the fixtures hold numeric inputs and outputs only.
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from make_golden import STANDIN  # noqa: E402

EXTRA = {
    'GPy/__init__.py': 'from . import likelihoods, kern\n',
    'GPy/likelihoods.py': '',
    'GPy/kern.py': '',
    'GPy/core/__init__.py': (
        'class SparseGP(object):\n'
        '    pass\n'
        'class Param(object):\n'
        '    pass\n'),
    'GPy/core/parameterization/transformations.py': (
        'class Logexp(object):\n'
        '    pass\n'),
    # arrays that carry .gradient and .values like a paramz Param; posteriors that slice into views of themselves
    'GPy/core/parameterization/variational.py': (
        'import numpy as np\n'
        'class GA(np.ndarray):\n'
        '    def __new__(cls, a):\n'
        '        o = np.asarray(a, dtype=float).view(cls)\n'
        '        o.gradient = np.zeros(o.shape)\n'
        '        return o\n'
        '    def __array_finalize__(self, obj):\n'
        '        if not hasattr(self, "gradient"):\n'
        '            self.gradient = None\n'
        '    @property\n'
        '    def values(self):\n'
        '        return self.view(np.ndarray)\n'
        'class VariationalPosterior(object):\n'
        '    def __init__(self, mean, variance, name=None):\n'
        '        self.mean = mean if (isinstance(mean, GA) and mean.gradient is not None) else GA(mean)\n'
        '        self.variance = variance if (isinstance(variance, GA) and variance.gradient is not None) else GA(variance)\n'
        '        self.num_data, self.input_dim = self.mean.shape\n'
        '        self.shape = self.mean.shape\n'
        '    def __getitem__(self, s):\n'
        '        p = object.__new__(type(self))\n'
        '        for nm in ("mean", "variance"):\n'
        '            a = getattr(self, nm)\n'
        '            v = a.values[s].view(GA)\n'
        '            v.gradient = a.gradient[s]\n'
        '            setattr(p, nm, v)\n'
        '        p.num_data, p.input_dim = p.mean.shape\n'
        '        p.shape = p.mean.shape\n'
        '        return p\n'
        'class NormalPosterior(VariationalPosterior):\n'
        '    pass\n'
        'class NormalPrior(object):\n'
        '    pass\n'),
    'paramz/__init__.py': '',
    'paramz/transformations.py': (
        'class Transformation(object):\n'
        '    pass\n'
        '_lim_val = 36.0\n'
        'epsilon = 2.220446049250313e-16\n'),
}


def import_reference_layers():
    tmp = tempfile.mkdtemp(prefix='gpy_standin_layers_')
    files = dict(STANDIN)
    files.update(EXTRA)
    for rel, src in files.items():
        p = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, 'w') as f:
            f.write(src)
    sys.path.insert(0, tmp)
    sys.dont_write_bytecode = True
    pkg = types.ModuleType('autoreg')
    pkg.__path__ = [os.path.join(REF, 'autoreg')]
    sys.modules['autoreg'] = pkg
    layers = importlib.import_module('autoreg.layers')
    inference = importlib.import_module('autoreg.inference')
    from GPy.core.parameterization.variational import GA, NormalPosterior
    return layers, inference, GA, NormalPosterior


class Lik(object):
    def __init__(self, v):
        self.variance = np.array([float(v)])
        self.gradient = None

    def update_gradients(self, g):
        self.gradient = np.asarray(g, dtype=float).reshape(-1).copy()


def make_kern_class(ps):
    class Kern(ps.RBF):
        """oracle/psi_stats.RBF behind the GPy kernel method names the reference's layer code calls."""

        def update_gradients_expectations(self, variational_posterior, Z, dL_dpsi0, dL_dpsi1, dL_dpsi2):
            g = self.psi_grads(np.asarray(Z), variational_posterior, dL_dpsi0, dL_dpsi1, dL_dpsi2)
            self.gradient = np.hstack([np.atleast_1d(g['dvar']), g['dinv_l']])

        def gradients_Z_expectations(self, dL_dpsi0, dL_dpsi1, dL_dpsi2, Z, variational_posterior):
            return self.psi_grads(np.asarray(Z), variational_posterior, dL_dpsi0, dL_dpsi1, dL_dpsi2)['dZ']

        def gradients_qX_expectations(self, variational_posterior, Z, dL_dpsi0, dL_dpsi1, dL_dpsi2):
            g = self.psi_grads(np.asarray(Z), variational_posterior, dL_dpsi0, dL_dpsi1, dL_dpsi2)
            return g['dmu'], g['dS']

        def update_gradients_full(self, dL_dK, X, X2=None):
            assert X2 is None
            kv, kl, _ = self.K_grads(np.asarray(X), dL_dK)
            self.gradient = np.hstack([np.atleast_1d(kv), kl])

        def gradients_X(self, dL_dK, X, X2=None):
            assert X2 is None
            return self.K_grads(np.asarray(X), dL_dK)[2]
    return Kern


def build_and_run(svi, seed):
    from oracle import psi_stats as ps
    layers, inference, GA, NormalPosterior = import_reference_layers()
    Kern = make_kern_class(ps)
    cls = layers.Layer_new if svi else layers.Layer

    class Duck(object):
        pass
    for nm in ('_update_conv', '_init_XY', '_update_X', 'update_latent_gradients', '_update_qX_gradients', '_prepare_gradients'):
        setattr(Duck, nm, getattr(cls, nm))
    Duck._inference_vardtc = layers.SparseGP_MPI._inference_vardtc

    rng = np.random.default_rng(seed)
    wins, nDims, U_win, u_dim, M = [0, 3, 2], [3, 2, 3], 2, 2, 4
    lens = [9, 12, 10]                                   # raw lengths (ragged)
    n_seq, n_total = len(lens), 4
    U_raw = [rng.standard_normal((t, u_dim)) for t in lens]
    Y_raw = [np.tanh(np.cumsum(rng.standard_normal((t, nDims[0])), 0) * 0.3) for t in lens]
    # data shift (model.py:657-673): U <- U[:-1] as N(U, 1e-10), Y <- Y[U_win:]
    Us = [NormalPosterior(u[:-1].copy(), np.full(u[:-1].shape, 1e-10)) for u in U_raw]
    Ys = [y[U_win:].copy() for y in Y_raw]
    # hidden latents (model.py:691-716): wins[i] + len(Y) rows
    Xs = {}
    for i in (1, 2):
        Xs[i] = [NormalPosterior(rng.standard_normal((wins[i] + y.shape[0], nDims[i])) * 0.7,
                                 0.02 + 0.2 * rng.random((wins[i] + y.shape[0], nDims[i]))) for y in Ys]
    out = {'lens': np.array(lens), 'wins': np.array(wins), 'nDims': np.array(nDims), 'U_win': np.array(U_win), 'M': np.array(M),
           'n_total': np.array(n_total), 'svi': np.array(int(svi))}
    for s in range(n_seq):
        out['in_U_%d' % s], out['in_Y_%d' % s] = U_raw[s], Y_raw[s]
        for i in (1, 2):
            out['in_Xmean_%d_%d' % (i, s)] = np.array(Xs[i][s].mean.values)
            out['in_Xvar_%d_%d' % (i, s)] = np.array(Xs[i][s].variance.values)
    # layers top first (model.py:680-689): layer 2 fed by the controls, layer 1 by layer 2, layer 0 (observed) by layer 1
    objs = {}
    for i in (2, 1, 0):
        o = Duck()
        o.layer_upper = None if i == 2 else objs[i + 1]
        o.Xs_flat = Ys if i == 0 else Xs[i]
        o.nSeq = n_seq
        o.X_win, o.X_dim = wins[i], nDims[i]
        o.X_observed = (i == 0)
        o.withControl = True
        o.Us_flat = Us if i == 2 else Xs[i + 1]
        o.U_win = U_win if i == 2 else wins[i + 1]
        o.U_dim = u_dim if i == 2 else nDims[i + 1]
        o.back_cstr = False
        o.svi = svi
        o.mpi_comm, o.mpi_root, o.psicov = None, 0, False
        o.qU_ratio = float(n_seq) / n_total if svi else 1.0
        o.Y_metadata = None
        o.normalPrior, o.normalEntropy = layers.NormalPrior(), layers.NormalEntropy()
        o._init_XY()
        Q = o.X.mean.shape[1]
        o.Z = GA(rng.standard_normal((M, Q)))
        o.kern = Kern(Q, 0.6 + rng.random(), 0.3 + rng.random(Q), fast=False)
        o.likelihood = Lik(0.1 + 0.3 * rng.random())
        o.inference_method = inference.SVI_VarDTC() if svi else inference.VarDTC()
        if svi:
            D = nDims[i]
            o.qU_mean = GA(rng.standard_normal((M, D)))
            o.qU_W = GA(rng.standard_normal((M, M)) * 0.3)
            o.qU_a = GA(np.array([0.2 + rng.random()]))
        objs[i] = o
        out['in_Z_%d' % i] = np.array(o.Z.values)
        out['in_variance_%d' % i], out['in_inv_l_%d' % i] = np.array(o.kern.variance), np.array(o.kern.inv_l)
        out['in_noise_%d' % i] = np.array(o.likelihood.variance)
        if svi:
            out['in_qU_m_%d' % i], out['in_qU_W_%d' % i], out['in_qU_a_%d' % i] = (np.array(o.qU_mean.values), np.array(o.qU_W.values),
                                                                                     np.array(o.qU_a.values))
    # one objective + gradient evaluation: layers top-down (update_layer, layers.py:617-621 / parameters_changed :1483-1488),
    # then the bottom-up scatter (model.py:721-723)
    total = 0.0
    for i in (2, 1, 0):
        o = objs[i]
        o._update_X()
        o._inference_vardtc()
        o._update_qX_gradients()
        o._prepare_gradients()
        total += float(np.asarray(o._log_marginal_likelihood).reshape(-1)[0])
        out['X_mean_%d' % i], out['X_var_%d' % i] = np.array(o.X.mean.values), np.array(o.X.variance.values)
        out['Y_mean_%d' % i] = np.array(o.Y if i == 0 else o.Y.mean.values)
        out['bound_%d' % i] = np.array(float(np.asarray(o._log_marginal_likelihood).reshape(-1)[0]))
        out['g_kern_%d' % i] = np.array(o.kern.gradient)             # [variance, inv_l...]
        out['g_Z_%d' % i] = np.array(o.Z.gradient)
        out['g_noise_%d' % i] = np.array(o.likelihood.gradient)
        if svi:
            out['g_qU_m_%d' % i], out['g_qU_W_%d' % i] = np.array(o.qU_mean.gradient), np.array(o.qU_W.gradient)
            out['g_qU_a_%d' % i] = np.array(o.qU_a.gradient)
        out['gX_mean_rows_%d' % i], out['gX_var_rows_%d' % i] = np.array(o.X.mean.gradient), np.array(o.X.variance.gradient)
    for i in (0, 1, 2):
        objs[i].update_latent_gradients()
    out['elbo'] = np.array(total)
    for s in range(n_seq):
        for i in (1, 2):
            out['g_Xmean_%d_%d' % (i, s)] = np.array(Xs[i][s].mean.gradient)
            out['g_Xvar_%d_%d' % (i, s)] = np.array(Xs[i][s].variance.gradient)
        out['g_Umean_%d' % s] = np.array(Us[s].mean.gradient)
    return out


def main():
    for svi, name, seed in ((True, 'svi', 301), (False, 'vardtc', 302)):
        out = build_and_run(svi, seed)
        path = os.path.join(HERE, 'layer_%s.npz' % name)
        np.savez_compressed(path, **out)
        print('wrote', path, 'elbo=%.12g' % out['elbo'])
        for m in [k for k in sys.modules if k == 'autoreg' or k.startswith('autoreg.') or k == 'GPy' or k.startswith('GPy.')
                  or k == 'paramz' or k.startswith('paramz.')]:
            del sys.modules[m]


if __name__ == '__main__':
    main()
