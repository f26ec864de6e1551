"""Pin oracle/model.py (restated autoreg.layers / autoreg.model semantics): hand-computed window alignment,
scatter == the reference's per-row loop, finite-difference gradient check (the reference's
own ``checkgrad`` gate, testing/minibatch_tests.py:98-100) and additivity / permutation over minibatches
(testing/minibatch_tests.py:281-296)."""
import numpy as np
import pytest

from oracle import model as om


def make_data(n_seq, T, u_dim=2, y_dim=3, seed=0):
    rng = np.random.default_rng(seed)
    U = [rng.standard_normal((T, u_dim)) for _ in range(n_seq)]
    Y = [np.tanh(np.cumsum(rng.standard_normal((T, y_dim)), 0) * 0.3) + 0.1 * rng.standard_normal((T, y_dim)) for _ in range(n_seq)]
    return U, Y


def make_model(inference, n_seq=3, T=14, n_total=None, seed=0, wins=(0, 3, 3), nDims=(3, 2, 3), M=4):
    U, Y = make_data(n_seq, T, seed=seed)
    m = om.DeepAutoreg(list(wins), Y, U=U, U_win=2, nDims=list(nDims), X_variance=0.05, num_inducing=M,
                       init='Y', inference_method=inference, n_total=n_total, rng=np.random.default_rng(seed + 1),
                       fast_psi=False)
    rng = np.random.default_rng(seed + 2)
    for i in m.order:                       # move away from the symmetric init so gradients are generic
        p = m.P[i]
        p.inv_l[:] = 0.3 + rng.random(p.inv_l.shape)
        p.variance = 0.7 + rng.random()
        p.noise = 0.1 + 0.3 * rng.random()
        if inference == 'svi':
            p.qU_W[:] = rng.standard_normal(p.qU_W.shape) * 0.2
            p.qU_a = 0.3
    return m


def test_window_alignment_by_hand():
    # one sequence, T=8 raw steps, U_win=2, wins=[0,3]: after the shift U has 7 rows, Y has 6 rows,
    # hidden latent has 3+6=9 rows.
    T = 8
    U = [np.arange(T, dtype=float)[:, None] * 10.0]                       # u_t = 10 t
    Y = [np.arange(T, dtype=float)[:, None] + 100.0]                      # y_t = 100 + t
    m = om.DeepAutoreg([0, 3], Y, U=U, U_win=2, nDims=[1, 1], num_inducing=2, init='Y', rng=np.random.default_rng(0))
    assert m.Us[0].shape == (7, 1) and m.Ys[0].shape == (6, 1)
    np.testing.assert_array_equal(m.Ys[0][:, 0], 100.0 + np.arange(2, 8))
    lat = m.Xmean[1][0]
    assert lat.shape == (9, 1)
    np.testing.assert_array_equal(lat[:3, 0], 0.0)
    np.testing.assert_array_equal(lat[3:, 0], m.Ys[0][:, 0])
    lat[:, 0] = np.arange(9) + 0.5                                        # x_j = j + .5
    # top layer (i=1): N = 9-3 = 6 rows; row n = [x_n, x_n+1, x_n+2 | u_n, u_n+1], target x_{n+3}
    Xm, Xv, Ym, Yv = m.build_XY(1, [0])
    assert Xm.shape == (6, 5)
    for n in range(6):
        np.testing.assert_array_equal(Xm[n], [n + .5, n + 1.5, n + 2.5, 10. * n, 10. * (n + 1)])
        assert Ym[n, 0] == n + 3.5
    np.testing.assert_array_equal(Xv[:, 3:], 1e-10)
    # observed layer (i=0): X_win=0, inputs = latent above with window wins[1]=3, U_off = 9-6-3+1 = 1:
    # row n = [x_{n+1}, x_{n+2}, x_{n+3}] (current step included), target y row n.
    Xm, Xv, Ym, Yv = m.build_XY(0, [0])
    assert Xm.shape == (6, 3) and Yv is None
    for n in range(6):
        np.testing.assert_array_equal(Xm[n], [n + 1.5, n + 2.5, n + 3.5])


def test_scatter_is_reference_loop_and_adjoint():
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((11, 3 * 2))
    a = np.zeros((15, 2)); b = np.zeros((15, 2))
    om._scatter(a, rows, 2, 3)
    om.scatter_reference_loop(b, rows, 2, 3)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-15)
    # adjoint of conv_1d: <conv(x), rows> == <x, scatter(rows)>
    x = rng.standard_normal((13, 2))
    lhs = (om.conv_1d(x, 3) * rows).sum()
    t = np.zeros_like(x); om._scatter(t, rows, 0, 3)
    np.testing.assert_allclose(lhs, (x * t).sum(), rtol=1e-13)


@pytest.mark.parametrize('inference', [None, 'svi'])
def test_checkgrad(inference):
    m = make_model(inference, n_seq=2, T=10)
    x0 = m.get_optimizer_array()
    f0, g0 = m.objective_and_grad(x0)
    rng = np.random.default_rng(11)
    idx = rng.choice(x0.size, size=min(60, x0.size), replace=False)
    # always include one entry of every parameter block
    o = 0
    for k in m.param_list():
        idx = np.append(idx, o); o += m.get(k).size
    eps = 1e-6
    for j in np.unique(idx):
        xp = x0.copy(); xp[j] += eps
        xm = x0.copy(); xm[j] -= eps
        num = (m.objective_and_grad(xp)[0] - m.objective_and_grad(xm)[0]) / (2 * eps)
        assert abs(num - g0[j]) <= 1e-5 * max(1.0, abs(num)), (j, num, g0[j])


def test_svi_minibatch_additivity_and_permutation():
    m = make_model('svi', n_seq=6, T=12, n_total=6)
    glob = [k for k in m.param_list() if len(k) == 2]
    e_all, G_all = m.elbo_and_grads(range(6))
    e_perm, G_perm = m.elbo_and_grads([4, 2, 0, 5, 1, 3])
    assert np.isclose(e_perm, e_all, rtol=1e-13, atol=0)
    e1, G1 = m.elbo_and_grads([0, 1, 2])
    e2, G2 = m.elbo_and_grads([3, 4, 5])
    assert np.isclose(e1 + e2, e_all, rtol=1e-13, atol=0)
    for k in glob:
        np.testing.assert_allclose(np.asarray(G1[k]) + np.asarray(G2[k]), G_all[k], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(G_perm[k], G_all[k], rtol=1e-9, atol=1e-11)
    # local (per-sequence) gradients are untouched by what else is in the minibatch except through the shared globals
    for s in (0, 1, 2):
        np.testing.assert_allclose(G1[(1, 'Xmean', s)], G_all[(1, 'Xmean', s)], rtol=1e-9, atol=1e-11)


def test_fast_psi_path_matches_direct():
    a = make_model('svi', n_seq=2, T=10)
    ea, Ga = a.elbo_and_grads()
    a.fast_psi, a.chunk = True, 5
    eb, Gb = a.elbo_and_grads()
    assert np.isclose(ea, eb, rtol=1e-12)
    for k in Ga:
        np.testing.assert_allclose(Gb[k], Ga[k], rtol=1e-8, atol=1e-10)


def test_freerun_shapes_and_determinism():
    m = make_model(None, n_seq=2, T=12)
    m.elbo_and_grads()
    U = np.random.default_rng(5).standard_normal((15, 2))
    outs = m.freerun(U)
    assert len(outs) == 3
    assert outs[-1][0].shape == (15 - 2, 3)         # observed layer: X_win=0 -> step rows, y_dim cols
    assert np.all(outs[-1][1] > 0)
    outs2 = m.freerun(U)
    np.testing.assert_array_equal(outs[-1][0], outs2[-1][0])
