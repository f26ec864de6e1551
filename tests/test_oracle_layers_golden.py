"""Pin oracle/model.py (windows, row order, bound assembly incl. KL * qU_ratio, gradient routing, scatter) to the outputs of
the reference's own autoreg/layers.py run on a two-hidden-layer toy model (fixtures made by tests/golden/make_golden_layers.py:
Layer_new + SVI_VarDTC and Layer + VarDTC, ragged sequences, qU_ratio = 3/4)."""
import os

import numpy as np
import pytest

from oracle import model as om

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load_fixture(name):
    g = np.load(os.path.join(GOLD, 'layer_%s.npz' % name))
    n_seq = len(g['lens'])
    U = [g['in_U_%d' % s] for s in range(n_seq)]
    Y = [g['in_Y_%d' % s] for s in range(n_seq)]
    return g, U, Y, n_seq


def oracle_from_fixture(g, U, Y, n_seq):
    svi = bool(int(g['svi']))
    o = om.DeepAutoreg([int(w) for w in g['wins']], Y, U=U, U_win=int(g['U_win']), nDims=[int(d) for d in g['nDims']],
                       num_inducing=int(g['M']), init='Y', inference_method='svi' if svi else None, n_total=int(g['n_total']),
                       rng=np.random.default_rng(0), fast_psi=False)
    for i in (0, 1, 2):
        p = o.P[i]
        p.Z[...] = g['in_Z_%d' % i]
        p.inv_l[...] = g['in_inv_l_%d' % i]
        p.variance = float(g['in_variance_%d' % i].reshape(-1)[0])
        p.noise = float(g['in_noise_%d' % i].reshape(-1)[0])
        if svi:
            p.qU_m[...] = g['in_qU_m_%d' % i]
            p.qU_W[...] = g['in_qU_W_%d' % i]
            p.qU_a = float(g['in_qU_a_%d' % i].reshape(-1)[0])
    for i in (1, 2):
        for s in range(n_seq):
            o.Xmean[i][s][...] = g['in_Xmean_%d_%d' % (i, s)]
            o.Xvar[i][s][...] = g['in_Xvar_%d_%d' % (i, s)]
    return o, svi


def close(a, b, rtol=1e-10):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.size == b.size, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-300)
    assert np.abs(a.reshape(b.shape) - b).max() <= rtol * scale, (np.abs(a.reshape(b.shape) - b).max(), scale)


@pytest.mark.parametrize('name', ['svi', 'vardtc'])
def test_windows_match_reference_layer_code(name):
    """Layer._update_conv / _init_XY (layers.py:1399-1431): stacked lag windows and targets, row for row."""
    g, U, Y, n_seq = load_fixture(name)
    o, _ = oracle_from_fixture(g, U, Y, n_seq)
    for i in (0, 1, 2):
        Xm, Xv, Ym, Yv = o.build_XY(i, range(n_seq))
        assert np.array_equal(Xm, g['X_mean_%d' % i])
        assert np.array_equal(Xv, g['X_var_%d' % i])
        assert np.array_equal(Ym, g['Y_mean_%d' % i])


@pytest.mark.parametrize('name', ['svi', 'vardtc'])
def test_bound_and_gradients_match_reference_layer_code(name):
    """SparseGP_MPI._inference_vardtc, Layer._prepare_gradients, Layer.update_latent_gradients (layers.py:66-176,1433-1481)
    and the model's sum / bottom-up scatter (model.py:721-723)."""
    g, U, Y, n_seq = load_fixture(name)
    o, svi = oracle_from_fixture(g, U, Y, n_seq)
    elbo, G = o.elbo_and_grads()
    assert abs(elbo - float(g['elbo'])) <= 1e-12 * abs(float(g['elbo']))
    for i in (0, 1, 2):
        assert abs(o.layer_elbo[i] - float(g['bound_%d' % i])) <= 1e-11 * abs(float(g['bound_%d' % i]))
        close(G[(i, 'Z')], g['g_Z_%d' % i])
        close(np.hstack([np.atleast_1d(G[(i, 'variance')]), G[(i, 'inv_l')]]), g['g_kern_%d' % i])
        close(G[(i, 'noise')], g['g_noise_%d' % i])
        if svi:
            close(G[(i, 'qU_m')], g['g_qU_m_%d' % i])
            close(G[(i, 'qU_W')], g['g_qU_W_%d' % i])
            close(G[(i, 'qU_a')], g['g_qU_a_%d' % i])
    for i in (1, 2):
        for s in range(n_seq):
            close(G[(i, 'Xmean', s)], g['g_Xmean_%d_%d' % (i, s)])
            close(G[(i, 'Xvar', s)], g['g_Xvar_%d_%d' % (i, s)])


def test_raw_predict_is_the_moment_match_of_the_sparse_gp_predictor():
    """Definition-level pin of the oracle's restatement of GPy's SparseGP._raw_predict under an uncertain input (the
    call at layers.py:1568; GPy itself is not installed): with f(x) = k(x, Z) w and s(x) = k(x, x) - k(x, Z) W k(Z, x),
    and x ~ N(mu, diag(S)), the predictor returns  E[f(x)]  and  Var[f(x)] + E[s(x)]  (law of total variance).  Both are
    integrated here by tensor Gauss-Hermite quadrature from the kernel's definition alone."""
    rng = np.random.default_rng(5)
    Q, M, D = 2, 5, 3
    U = [rng.standard_normal((8, 1))]
    Y = [rng.standard_normal((8, D))]
    o = om.DeepAutoreg([0, 2], Y, U=U, U_win=1, nDims=[D, 1], num_inducing=M, init='Y', inference_method='svi',
                       rng=np.random.default_rng(1), fast_psi=False)
    p = o.P[0]
    assert p.Z.shape == (M, Q)
    p.Z[...] = rng.standard_normal((M, Q))
    p.inv_l[...] = 0.5 + rng.random(Q)
    p.variance = 1.3
    p.woodbury_vector = rng.standard_normal((M, D))
    A = rng.standard_normal((M, M))
    p.woodbury_inv = 0.05 * (A + A.T)
    mu = rng.standard_normal((3, Q))
    S = 0.05 + 0.4 * rng.random((3, Q))
    mean, var = o._raw_predict(0, mu, S)

    x, w = np.polynomial.hermite_e.hermegauss(60)                  # weight exp(-x^2 / 2)
    w = w / np.sqrt(2 * np.pi)
    for n in range(3):
        X = np.stack(np.meshgrid(mu[n, 0] + np.sqrt(S[n, 0]) * x, mu[n, 1] + np.sqrt(S[n, 1]) * x, indexing='ij'), -1).reshape(-1, Q)
        W = np.outer(w, w).reshape(-1)
        d = X[:, None, :] - p.Z[None, :, :]
        Kxz = p.variance * np.exp(-0.5 * (d * d * p.inv_l).sum(-1))       # inv_l = 1 / lengthscale^2 (GPy RBF, inv_l=True)
        f = Kxz.dot(p.woodbury_vector)
        s = p.variance - np.einsum('am,mk,ak->a', Kxz, p.woodbury_inv, Kxz)
        Ef = W.dot(f)
        Vf = W.dot(f * f) - Ef * Ef
        close(mean[n], Ef, rtol=1e-9)
        close(var[n], Vf + W.dot(s), rtol=1e-9)
