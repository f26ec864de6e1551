#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own inference code in the build container.

Run (build container only -- /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

What it does:
  1. writes a ~70-line stand-in for the eleven GPy symbols the reference's
     ``autoreg/inference/{vardtc,svi_vardtc}.py`` and ``autoreg/variational.py`` import
     (thin scipy.linalg.lapack wrappers + two record classes) into a temporary directory;
     GPy itself is an un-vendored, un-pinned third-party dependency that is not installed here;
  2. registers an empty ``autoreg`` package whose ``__path__`` points at /root/reference/autoreg so that the
     three modules import UNMODIFIED (``autoreg/__init__.py`` is not executed: it needs real GPy);
  3. feeds them seeded inputs plus psi-statistics computed by oracle/psi_stats.py (the psi closed forms are
     GPy's and are pinned separately by quadrature, see tests/test_oracle_psi.py) and stores inputs and every
     output as small .npz fixtures.
No reference source text is copied: fixtures are numeric inputs/outputs only.
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

STANDIN = {
    'GPy/__init__.py': '',
    'GPy/util/__init__.py': 'from . import linalg, diag\n',
    'GPy/util/diag.py': (
        'import numpy as np\n'
        'def add(A, b):\n'
        '    A[np.diag_indices(A.shape[0])] += b\n'
        '    return A\n'),
    'GPy/util/linalg.py': (
        'import numpy as np\n'
        'from scipy.linalg import lapack\n'
        'def jitchol(A, maxtries=5):\n'
        '    A = np.ascontiguousarray(A)\n'
        '    L, info = lapack.dpotrf(A, lower=1)\n'
        '    if info != 0:\n'
        '        raise np.linalg.LinAlgError("not positive definite (stand-in does not add jitter)")\n'
        '    return np.tril(L)\n'
        'def dtrtrs(A, B, lower=1, trans=0, unitdiag=0):\n'
        '    A = np.asfortranarray(A)\n'
        '    return lapack.dtrtrs(A, B, lower=lower, trans=trans, unitdiag=unitdiag)\n'
        'def dtrtri(L):\n'
        '    return lapack.dtrtri(np.asfortranarray(L), lower=1)[0]\n'
        'def dpotri(A, lower=1):\n'
        '    R, info = lapack.dpotri(np.asfortranarray(A), lower=lower)\n'
        '    R = np.tril(R) + np.tril(R, -1).T if lower else np.triu(R) + np.triu(R, 1).T\n'
        '    return R, info\n'
        'def tdot(A):\n'
        '    return A.dot(A.T)\n'
        'def backsub_both_sides(L, X, transpose="left"):\n'
        '    if transpose == "left":\n'
        '        tmp, _ = dtrtrs(L, X, lower=1, trans=1)\n'
        '        return dtrtrs(L, tmp.T, lower=1, trans=1)[0].T\n'
        '    tmp, _ = dtrtrs(L, X, lower=1, trans=0)\n'
        '    return dtrtrs(L, tmp.T, lower=1, trans=0)[0].T\n'
        'def pdinv(A):\n'
        '    L = jitchol(A)\n'
        '    Li = dtrtri(L)\n'
        '    Ai, _ = dpotri(L, lower=1)\n'
        '    return Ai, L, Li, 2.0 * np.sum(np.log(np.diag(L)))\n'),
    'GPy/core/__init__.py': '',
    'GPy/core/parameterization/__init__.py': '',
    'GPy/core/parameterization/variational.py': (
        'class VariationalPosterior(object):\n'
        '    def __init__(self, mean, variance):\n'
        '        self.mean, self.variance = mean, variance\n'
        '        self.num_data, self.input_dim = mean.shape\n'
        '        self.shape = mean.shape\n'
        'class NormalPosterior(VariationalPosterior):\n'
        '    pass\n'),
    'GPy/inference/__init__.py': '',
    'GPy/inference/latent_function_inference/__init__.py': (
        'class LatentFunctionInference(object):\n'
        '    pass\n'),
    'GPy/inference/latent_function_inference/posterior.py': (
        'class Posterior(object):\n'
        '    def __init__(self, woodbury_inv=None, woodbury_vector=None, K=None, mean=None, cov=None, K_chol=None):\n'
        '        self.woodbury_inv, self.woodbury_vector = woodbury_inv, woodbury_vector\n'
        '        self.K, self.mean, self.cov, self.K_chol = K, mean, cov, K_chol\n'),
}


def import_reference():
    tmp = tempfile.mkdtemp(prefix='gpy_standin_')
    for rel, src in STANDIN.items():
        p = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, 'w') as f:
            f.write(src)
    sys.path.insert(0, tmp)
    sys.dont_write_bytecode = True
    pkg = types.ModuleType('autoreg')
    pkg.__path__ = [os.path.join(REF, 'autoreg')]
    sys.modules['autoreg'] = pkg
    inference = importlib.import_module('autoreg.inference')
    variational = importlib.import_module('autoreg.variational')
    from GPy.core.parameterization.variational import NormalPosterior
    return inference, variational, NormalPosterior


class GradArray(np.ndarray):
    """ndarray carrying a ``.gradient`` like a paramz Param (what variational.py mutates)."""
    def __new__(cls, a):
        o = np.array(a, dtype=float).view(cls)
        o.gradient = np.zeros(o.shape)
        return o


class Lik(object):
    def __init__(self, v):
        self.variance = np.array([float(v)])


CASES = [
    # name, N, M, Q, D, n_ctrl_cols (variance 1e-10), uncertain_y, seed
    ('a', 13, 5, 3, 2, 0, False, 101),
    ('b', 13, 5, 3, 2, 0, True, 102),
    ('c', 40, 7, 6, 1, 2, False, 103),
    ('d', 40, 7, 6, 3, 2, True, 104),
    ('e', 64, 10, 8, 16, 4, False, 105),
    ('f', 25, 4, 2, 1, 1, True, 106),
]


def make_inputs(N, M, Q, D, nctrl, unc_y, seed):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, Q))
    S = 0.02 + 0.3 * rng.random((N, Q))
    if nctrl:
        S[:, Q - nctrl:] = 1e-10                    # observed controls (model.py:669)
    Z = rng.standard_normal((M, Q)) * 1.2
    variance = 0.5 + rng.random()
    inv_l = 0.3 + rng.random(Q)
    noise = 0.05 + 0.5 * rng.random()
    Ym = rng.standard_normal((N, D))
    Yv = (0.01 + 0.2 * rng.random((N, D))) if unc_y else None
    qm = rng.standard_normal((M, D))
    W = rng.standard_normal((M, M)) * 0.3
    a = 0.1 + rng.random()
    return dict(mu=mu, S=S, Z=Z, variance=variance, inv_l=inv_l, noise=noise, Ymean=Ym, Yvar=Yv,
                qU_mean=qm, qU_W=W, qU_a=a)


def main():
    sys.path.insert(0, ROOT)
    from oracle import psi_stats
    inference, variational, NormalPosterior = import_reference()
    for name, N, M, Q, D, nctrl, unc_y, seed in CASES:
        inp = make_inputs(N, M, Q, D, nctrl, unc_y, seed)
        kern = psi_stats.RBF(Q, inp['variance'], inp['inv_l'], fast=False)
        X = NormalPosterior(inp['mu'], inp['S'])
        Y = NormalPosterior(inp['Ymean'], inp['Yvar']) if unc_y else inp['Ymean']
        out = {}
        for k, v in inp.items():
            if v is not None:
                out['in_' + k] = np.asarray(v)
        out['psi0'] = kern.psi0(inp['Z'], X)
        out['psi1'] = kern.psi1(inp['Z'], X)
        out['psi2'] = kern.psi2(inp['Z'], X)
        out['Kuu'] = kern.K(inp['Z'])
        # ---- collapsed bound (vardtc.py:88) ----
        post, logL, gd = inference.VarDTC().inference(kern, X, inp['Z'], Lik(inp['noise']), Y)
        out['vardtc_logL'] = np.asarray(logL).reshape(())
        out['vardtc_woodbury_inv'] = post.woodbury_inv
        out['vardtc_woodbury_vector'] = post.woodbury_vector
        for k, v in gd.items():
            out['vardtc_' + k] = np.asarray(v)
        # ---- SVI bound (svi_vardtc.py:70) + KL (:197) ----
        qU_var = inp['qU_W'].dot(inp['qU_W'].T) + np.eye(M) * inp['qU_a']       # layers.py:71
        svi = inference.SVI_VarDTC()
        post, logL, gd = svi.inference(kern, X, inp['Z'], Lik(inp['noise']), Y, inp['qU_mean'], qU_var)
        out['svi_logL'] = np.asarray(logL).reshape(())
        out['svi_woodbury_inv'] = post.woodbury_inv
        out['svi_woodbury_vector'] = post.woodbury_vector
        for k, v in gd.items():
            out['svi_' + k] = np.asarray(v)
        KL, dm, dv, dK = svi.comp_KL_qU(inp['qU_mean'], qU_var)
        out['kl'] = np.asarray(KL)
        out['kl_dmean'], out['kl_dvar'], out['kl_dKuu'] = dm, dv, dK
        # ---- latent entropy / prior terms (variational.py) ----
        vp = NormalPosterior(GradArray(inp['mu']), GradArray(inp['S']))
        ent, pri = variational.NormalEntropy(), variational.NormalPrior()
        out['entropy_value'] = np.asarray(ent.comp_value(vp))
        ent.update_gradients(vp)
        out['entropy_dvar'] = np.array(vp.variance.gradient)
        vp = NormalPosterior(GradArray(inp['mu']), GradArray(inp['S']))
        out['prior_value'] = np.asarray(pri.comp_value(vp))
        pri.update_gradients(vp)
        out['prior_dmean'] = np.array(vp.mean.gradient)
        out['prior_dvar'] = np.array(vp.variance.gradient)
        path = os.path.join(HERE, 'inference_%s.npz' % name)
        np.savez_compressed(path, **out)
        print('wrote', path, 'vardtc logL=%.12g svi logL=%.12g KL=%.12g' % (out['vardtc_logL'], out['svi_logL'], out['kl']))


if __name__ == '__main__':
    main()
