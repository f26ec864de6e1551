"""GPU parity tests of the unit pieces of the C ABI against the oracle (fp64; tolerances stated per test)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    import torch
    from rgp_amd.capi import Handle
    assert torch.cuda.is_available()
    return Handle()


def dev(H, a):
    import torch
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=H.device)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make(N, M, Q, seed, nctrl=0):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, Q)) + 0.5
    S = 0.02 + 0.3 * rng.random((N, Q))
    if nctrl:
        S[:, Q - nctrl:] = 1e-10
    Z = rng.standard_normal((M, Q)) * 1.2 + 0.5
    return mu, S, Z, 0.5 + rng.random(), 0.3 + rng.random(Q)


SHAPES = [(13, 5, 3, 0), (70, 7, 6, 2), (257, 33, 20, 5), (300, 100, 40, 20), (129, 64, 8, 0), (64, 128, 57, 3)]


@pytest.mark.parametrize('N,M,Q,nctrl', SHAPES)
def test_kuu_and_psi_forward(H, N, M, Q, nctrl):
    from oracle import psi_stats as ps
    mu, S, Z, var, il = make(N, M, Q, 11 + N, nctrl)
    K = H.kuu(dev(H, Z), dev(H, [var]), dev(H, il)).cpu().numpy()
    assert rel(K, ps.rbf_K_exact(var, il, Z) + 1e-6 * np.eye(M)) < 1e-13
    p1, p2 = H.psi_forward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    H.sync()
    assert rel(p1.cpu().numpy(), ps.psi1_direct(var, il, Z, mu, S)) < 1e-12
    ref2 = ps.PsiStatsFast(64).psi2(var, il, Z, mu, S) if N * M * M * Q > 5e7 else ps.psi2_direct(var, il, Z, mu, S)
    assert rel(p2.cpu().numpy(), ref2) < 1e-11
    assert np.array_equal(p2.cpu().numpy(), p2.cpu().numpy().T)       # exactly symmetric by construction


@pytest.mark.parametrize('N,M,Q,nctrl', SHAPES)
def test_psi_backward(H, N, M, Q, nctrl):
    from oracle import psi_stats as ps
    mu, S, Z, var, il = make(N, M, Q, 23 + N, nctrl)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(N)
    w1 = rng.standard_normal((N, M))
    w2 = rng.standard_normal((M, M))                                   # not symmetric on purpose
    ref = ps.PsiStatsFast(64).grads(var, il, Z, mu, S, w0, w1, w2)
    g = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), w0.sum(), dev(H, w1), dev(H, w2))
    H.sync()
    tol = 1e-10
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g[k].cpu().numpy().reshape(np.shape(ref[k])), ref[k]) < tol, k
    # psi1-only and psi2-only paths
    ref1 = ps.PsiStatsFast(64).grads(var, il, Z, mu, S, 0 * w0, w1, None)
    g1 = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), 0.0, dev(H, w1), None)
    ref2 = ps.PsiStatsFast(64).grads(var, il, Z, mu, S, 0 * w0, None, w2)
    g2 = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), 0.0, None, dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g1[k].cpu().numpy().reshape(np.shape(ref1[k])), ref1[k]) < 1e-10, k
        assert rel(g2[k].cpu().numpy().reshape(np.shape(ref2[k])), ref2[k]) < tol, k


GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'inference_*.npz')))


@pytest.mark.parametrize('path', GOLD)
def test_svi_chain_against_reference_fixture(H, path):
    """The device M x M chain reproduces the REFERENCE's SVI_VarDTC.inference + comp_KL_qU outputs (golden fixtures)."""
    g = np.load(path)
    M, D = g['in_qU_mean'].shape
    N = g['in_mu'].shape[0]
    ratio = 0.37
    yry = (g['in_Ymean'] ** 2).sum() + (g['in_Yvar'].sum() if 'in_Yvar' in g.files else 0.0)
    o = H.svi_chain(N, dev(H, g['Kuu'] + 1e-6 * np.eye(M)), dev(H, [float(g['in_noise'])]), dev(H, g['in_qU_mean']),
                    dev(H, g['in_qU_W']), dev(H, [float(g['in_qU_a'])]), dev(H, [g['psi0'].sum()]),
                    dev(H, g['in_Ymean'].T.dot(g['psi1'])), dev(H, g['psi2']), dev(H, [yry]), qU_ratio=ratio)
    H.sync()
    c = lambda k: o[k].cpu().numpy()
    assert abs(c('bound')[0] - (g['svi_logL'] - ratio * g['kl'])) < 1e-10 * abs(g['svi_logL'])
    assert rel(c('dL_dKmm'), g['svi_dL_dKmm'] - ratio * g['kl_dKuu']) < 1e-9
    assert rel(c('dL_dpsi2'), g['svi_dL_dpsi2']) < 1e-9
    assert rel(c('Kinv_mu'), g['svi_woodbury_vector']) < 1e-9
    assert rel(c('woodbury_inv'), g['svi_woodbury_inv']) < 1e-9
    assert rel(c('dL_dthetaL'), g['svi_dL_dthetaL'].reshape(1)) < 1e-9
    assert rel(c('gqU_m'), g['svi_dL_dqU_mean'] - ratio * g['kl_dmean']) < 1e-9
    Gv = g['svi_dL_dqU_var'] - ratio * g['kl_dvar']
    assert rel(c('gqU_W'), (Gv + Gv.T).dot(g['in_qU_W'])) < 1e-9
    assert rel(c('gqU_a'), np.trace(Gv).reshape(1)) < 1e-9


@pytest.mark.parametrize('spread,tol', [(1.0, 1e-11), (0.3, 1e-8)])
def test_svi_chain_m100_against_oracle(H, spread, tol):
    """M = 100 (the benchmark's size) with a well and an ill conditioned K_uu (cond ~ 4e3 / 5e6): the device chain evaluates
    the reference's expressions in a reduced form (rgp_chain.hip header), so it is checked against the solve-based oracle
    where rounding differences would show; the tolerance scales with eps * cond(K_uu)."""
    from oracle import inference as inf, psi_stats as ps
    rng = np.random.default_rng(7)
    M, Q, D, N = 100, 12, 16, 600
    Z = rng.standard_normal((M, Q)) * spread
    il, var, noise = np.full(Q, 0.3), 1.3, 0.05
    mu = rng.standard_normal((N, Q)) * spread
    S = 0.05 + 0.1 * rng.random((N, Q))
    P = ps.PsiStatsFast(64)
    psi1, psi2 = P.psi1(var, il, Z, mu, S), P.psi2(var, il, Z, mu, S)
    Kuu = var * np.exp(-0.5 * (((Z[:, None] - Z[None]) * il) ** 2).sum(-1))
    Y = rng.standard_normal((N, D))
    qm, W, a = rng.standard_normal((M, D)), 0.3 * rng.standard_normal((M, M)), 0.5
    qS = W.dot(W.T) + a * np.eye(M)
    st = inf.Stats(N * var, psi1, psi2, Y)
    logL, g, winv, wvec, mid = inf.svi(Kuu, st, noise, qm, qS)
    KL, dKLm, dKLv, dKLK = inf.kl_qU(qm, qS, mid)
    ratio = 0.37
    o = H.svi_chain(N, dev(H, Kuu + 1e-6 * np.eye(M)), dev(H, [noise]), dev(H, qm), dev(H, W), dev(H, [a]), dev(H, [N * var]),
                    dev(H, Y.T.dot(psi1)), dev(H, psi2), dev(H, [(Y ** 2).sum()]), qU_ratio=ratio)
    H.sync()
    c = lambda k: o[k].cpu().numpy()
    Gv = g['dL_dqU_var'] - ratio * dKLv
    assert abs(c('bound')[0] - (logL - ratio * KL)) < tol * abs(logL)
    assert rel(c('dL_dKmm'), g['dL_dKmm'] - ratio * dKLK) < tol
    assert rel(c('dL_dpsi2'), g['dL_dpsi2']) < tol
    assert rel(c('Kinv_mu'), wvec) < tol
    assert rel(c('woodbury_inv'), winv) < tol
    assert rel(c('dL_dthetaL'), np.asarray(g['dL_dthetaL']).reshape(1)) < tol
    assert rel(c('gqU_m'), g['dL_dqU_mean'] - ratio * dKLm) < tol
    assert rel(c('gqU_W'), (Gv + Gv.T).dot(W)) < tol
    assert rel(c('gqU_a'), np.trace(Gv).reshape(1)) < tol


def test_not_pd_is_reported(H):
    from rgp_amd._lib import RgpError
    M, D = 4, 1
    K = -np.eye(M)
    with pytest.raises(RgpError):
        H.svi_chain(10, dev(H, K), dev(H, [0.1]), dev(H, np.zeros((M, D))), dev(H, np.eye(M)), dev(H, [0.1]),
                    dev(H, [1.0]), dev(H, np.zeros((D, M))), dev(H, np.eye(M)), dev(H, [1.0]))
        H.sync()


def test_adam_step_matches_numpy(H):
    import torch
    rng = np.random.default_rng(0)
    n = 1000
    x = rng.standard_normal(n); g = rng.standard_normal(n)
    pos = (rng.random(n) < 0.5).astype(np.uint8)
    sp = lambda v: np.log1p(np.exp(v))
    p = np.where(pos, sp(x), x)
    m = np.zeros(n); v = np.zeros(n)
    xd, pd_, gd, md, vd = dev(H, x), dev(H, p), dev(H, g), dev(H, m), dev(H, v)
    posd = torch.tensor(pos, device=H.device)
    for step in (1, 2, 3):
        H.adam_step(xd, pd_, gd, md, vd, posd, 1e-2, step)
        gi = np.where(pos, g * (1 - np.exp(-p)), g)
        m = 0.9 * m + 0.1 * gi; v = 0.999 * v + 0.001 * gi * gi
        x = x + 1e-2 * (m / (1 - 0.9 ** step)) / (np.sqrt(v / (1 - 0.999 ** step)) + 1e-8)
        p = np.where(pos, sp(x), x)
    H.sync()
    assert rel(xd.cpu().numpy(), x) < 1e-13 and rel(pd_.cpu().numpy(), p) < 1e-13
