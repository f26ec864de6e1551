"""Pin oracle/psi_stats.py: psi values against the DEFINITION (Gaussian expectations of the ARD-RBF kernel, by
Gauss-Hermite quadrature) and every gradient against central finite differences.  GPy (where these closed forms
live) is not under /root/reference, so this is what stands in for a reference fixture."""
import itertools

import numpy as np
import pytest

from oracle import psi_stats as ps


def _setup(N=5, M=4, Q=2, seed=0):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, Q))
    S = 0.05 + 0.5 * rng.random((N, Q))
    Z = rng.standard_normal((M, Q))
    return 1.3, 0.4 + rng.random(Q), Z, mu, S


def test_psi_against_quadrature():
    var, il, Z, mu, S = _setup()
    N, Q = mu.shape
    M = Z.shape[0]
    x, w = np.polynomial.hermite_e.hermegauss(40)          # weight exp(-x^2/2)
    w = w / np.sqrt(2 * np.pi)
    p1 = np.zeros((N, M))
    p2 = np.zeros((N, M, M))
    for idx in itertools.product(range(len(x)), repeat=Q):
        xi = np.array([x[i] for i in idx])
        wi = np.prod([w[i] for i in idx])
        pts = mu + np.sqrt(S) * xi[None, :]                # (N,Q) one node per data row
        k = np.stack([ps.rbf_K_exact(var, il, pts[n:n + 1], Z)[0] for n in range(N)])    # (N,M)
        p1 += wi * k
        p2 += wi * k[:, :, None] * k[:, None, :]
    np.testing.assert_allclose(ps.psi1_direct(var, il, Z, mu, S), p1, rtol=1e-10)
    np.testing.assert_allclose(ps.psi2n_direct(var, il, Z, mu, S), p2, rtol=1e-10)
    np.testing.assert_allclose(ps.psi0_direct(var, mu), var)


def test_certain_input_limit():
    var, il, Z, mu, S = _setup(seed=3)
    S0 = np.full_like(S, 1e-13)
    Kfu = ps.rbf_K_exact(var, il, mu, Z)
    np.testing.assert_allclose(ps.psi1_direct(var, il, Z, mu, S0), Kfu, rtol=1e-9)
    np.testing.assert_allclose(ps.psi2_direct(var, il, Z, mu, S0), Kfu.T.dot(Kfu), rtol=1e-9)


def _objective(var, il, Z, mu, S, w0, w1, w2):
    return ((w0 * ps.psi0_direct(var, mu)).sum() + (w1 * ps.psi1_direct(var, il, Z, mu, S)).sum()
            + (w2 * ps.psi2_direct(var, il, Z, mu, S)).sum())


@pytest.mark.parametrize('impl', ['direct', 'fast'])
def test_psi_gradients_fd(impl):
    var, il, Z, mu, S = _setup(N=6, M=5, Q=3, seed=1)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(mu.shape[0])
    w1 = rng.standard_normal((mu.shape[0], Z.shape[0]))
    w2 = rng.standard_normal((Z.shape[0], Z.shape[0]))      # deliberately NOT symmetric
    if impl == 'direct':
        g = ps.psi_grads_direct(var, il, Z, mu, S, w0, w1, w2)
    else:
        g = ps.PsiStatsFast(chunk=4).grads(var, il, Z, mu, S, w0, w1, w2)
    eps = 1e-6

    def fd(arr, i):
        a = arr.copy(); a.flat[i] += eps
        b = arr.copy(); b.flat[i] -= eps
        return a, b
    for name, arr in (('dZ', Z), ('dmu', mu), ('dS', S), ('dinv_l', il)):
        num = np.zeros(arr.size)
        for i in range(arr.size):
            a, b = fd(arr, i)
            args_a = dict(Z=Z, mu=mu, S=S, il=il); args_b = dict(args_a)
            key = {'dZ': 'Z', 'dmu': 'mu', 'dS': 'S', 'dinv_l': 'il'}[name]
            args_a[key] = a; args_b[key] = b
            num[i] = (_objective(var, args_a['il'], args_a['Z'], args_a['mu'], args_a['S'], w0, w1, w2)
                      - _objective(var, args_b['il'], args_b['Z'], args_b['mu'], args_b['S'], w0, w1, w2)) / (2 * eps)
        np.testing.assert_allclose(np.asarray(g[name]).ravel(), num, rtol=2e-6, atol=1e-7)
    num = (_objective(var + eps, il, Z, mu, S, w0, w1, w2) - _objective(var - eps, il, Z, mu, S, w0, w1, w2)) / (2 * eps)
    np.testing.assert_allclose(g['dvar'], num, rtol=1e-6)


def test_fast_equals_direct():
    var, il, Z, mu, S = _setup(N=33, M=7, Q=5, seed=2)
    rng = np.random.default_rng(7)
    w0 = rng.standard_normal(33); w1 = rng.standard_normal((33, 7)); w2 = rng.standard_normal((7, 7))
    a = ps.psi_grads_direct(var, il, Z, mu, S, w0, w1, w2)
    b = ps.PsiStatsFast(chunk=8).grads(var, il, Z, mu, S, w0, w1, w2)
    for k in a:
        np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-11)


def test_K_gradients_fd():
    var, il, Z, _, _ = _setup(M=6, Q=3, seed=4)
    rng = np.random.default_rng(9)
    W = rng.standard_normal((6, 6))
    dvar, dil, dZ = ps.rbf_K_grads(var, il, Z, W)
    f = lambda v, l, z: (W * ps.rbf_K_exact(v, l, z)).sum()
    eps = 1e-6
    num = np.zeros(Z.size)
    for i in range(Z.size):
        a = Z.copy(); a.flat[i] += eps; b = Z.copy(); b.flat[i] -= eps
        num[i] = (f(var, il, a) - f(var, il, b)) / (2 * eps)
    np.testing.assert_allclose(dZ.ravel(), num, rtol=1e-6, atol=1e-8)
    num = np.zeros(il.size)
    for i in range(il.size):
        a = il.copy(); a[i] += eps; b = il.copy(); b[i] -= eps
        num[i] = (f(var, a, Z) - f(var, b, Z)) / (2 * eps)
    np.testing.assert_allclose(dil, num, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(dvar, (f(var + eps, il, Z) - f(var - eps, il, Z)) / (2 * eps), rtol=1e-6)
