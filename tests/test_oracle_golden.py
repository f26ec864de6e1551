"""Pin oracle/inference.py to the outputs of the reference's own autoreg.inference / autoreg.variational
(fixtures made by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import inference as oi
from oracle import psi_stats as ps

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'inference_*.npz')))
RTOL = 1e-10


def _close(a, b, rtol=RTOL):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.abs(b).max(), 1e-300)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= rtol * scale, (np.abs(a - b).max(), scale)


def _stats(g):
    Yv = g['in_Yvar'] if 'in_Yvar' in g.files else None
    return oi.Stats(g['psi0'].sum(), g['psi1'], g['psi2'], g['in_Ymean'], Yv)


def test_have_fixtures():
    assert len(CASES) >= 6


@pytest.mark.parametrize('path', CASES)
def test_psi_fixture_matches_oracle(path):
    g = np.load(path)
    v, il, Z, mu, S = float(g['in_variance']), g['in_inv_l'], g['in_Z'], g['in_mu'], g['in_S']
    fast = ps.PsiStatsFast(chunk=7)
    _close(ps.psi1_direct(v, il, Z, mu, S), g['psi1'], 1e-13)
    _close(ps.psi2_direct(v, il, Z, mu, S), g['psi2'], 1e-13)
    _close(fast.psi1(v, il, Z, mu, S), g['psi1'], 1e-11)
    _close(fast.psi2(v, il, Z, mu, S), g['psi2'], 1e-11)
    _close(ps.rbf_K_exact(v, il, Z), g['Kuu'], 1e-14)
    _close(ps.rbf_K(v, il, Z), g['Kuu'], 1e-11)


@pytest.mark.parametrize('path', CASES)
def test_vardtc_matches_reference(path):
    g = np.load(path)
    logL, gd, winv, wvec = oi.vardtc(g['Kuu'], _stats(g), float(g['in_noise']))
    _close(logL, g['vardtc_logL'], 1e-12)
    _close(winv, g['vardtc_woodbury_inv'])
    _close(wvec, g['vardtc_woodbury_vector'])
    for k in ('dL_dKmm', 'dL_dpsi0', 'dL_dpsi1', 'dL_dpsi2', 'dL_dthetaL'):
        _close(np.asarray(gd[k]).reshape(g['vardtc_' + k].shape), g['vardtc_' + k])
    if 'in_Yvar' in g.files:
        _close(gd['dL_dYmean'], g['vardtc_dL_dYmean'])
        _close(gd['dL_dYvar'], g['vardtc_dL_dYvar'])
        assert g['vardtc_dL_dYvar'].ndim == 1          # (N,) -- layers.py:1474 broadcasts it


@pytest.mark.parametrize('path', CASES)
def test_svi_and_kl_match_reference(path):
    g = np.load(path)
    M = g['in_Z'].shape[0]
    S = g['in_qU_W'].dot(g['in_qU_W'].T) + np.eye(M) * float(g['in_qU_a'])
    logL, gd, winv, wvec, mid = oi.svi(g['Kuu'], _stats(g), float(g['in_noise']), g['in_qU_mean'], S)
    _close(logL, g['svi_logL'], 1e-12)
    _close(winv, g['svi_woodbury_inv'])
    _close(wvec, g['svi_woodbury_vector'])
    for k in ('dL_dKmm', 'dL_dpsi0', 'dL_dpsi1', 'dL_dpsi2', 'dL_dthetaL', 'dL_dqU_mean', 'dL_dqU_var'):
        _close(np.asarray(gd[k]).reshape(g['svi_' + k].shape), g['svi_' + k])
    if 'in_Yvar' in g.files:
        _close(gd['dL_dYmean'], g['svi_dL_dYmean'])
        _close(gd['dL_dYvar'], g['svi_dL_dYvar'])
        assert g['svi_dL_dYvar'].ndim == 2             # (N,D) -- layers.py:606
    KL, dm, dv, dK = oi.kl_qU(g['in_qU_mean'], S, mid)
    _close(KL, g['kl'], 1e-12)
    _close(dm, g['kl_dmean'])
    _close(dv, g['kl_dvar'])
    _close(dK, g['kl_dKuu'])


@pytest.mark.parametrize('path', CASES)
def test_latent_terms_match_reference(path):
    g = np.load(path)
    mu, S = g['in_mu'], g['in_S']
    _close(oi.normal_entropy(S), g['entropy_value'], 1e-14)
    _close(oi.normal_entropy_grad_var(S), g['entropy_dvar'], 1e-14)
    _close(oi.normal_prior_kl(mu, S), g['prior_value'], 1e-14)
    dm, dv = oi.normal_prior_grads(mu, S)
    _close(dm, g['prior_dmean'], 1e-14)
    _close(dv, g['prior_dvar'], 1e-14)
