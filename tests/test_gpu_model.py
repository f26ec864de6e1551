"""GPU parity of the fused layer step / whole-model ELBO and gradients against the oracle model, plus the reference's
own invariants (checkgrad, additivity and permutation over minibatches: testing/minibatch_tests.py:98-100,281-296)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_data(n_seq, T, u_dim=2, y_dim=3, seed=0, ragged=False):
    rng = np.random.default_rng(seed)
    U, Y = [], []
    for s in range(n_seq):
        t = T + (3 * s if ragged else 0)
        U.append(rng.standard_normal((t, u_dim)))
        Y.append(np.tanh(np.cumsum(rng.standard_normal((t, y_dim)), 0) * 0.3) + 0.1 * rng.standard_normal((t, y_dim)))
    return U, Y


def build_pair(n_seq=3, T=14, wins=(0, 3, 3), nDims=(3, 2, 3), M=4, U_win=2, seed=0, ragged=False, n_total=None,
               u_dim=2, y_dim=3, inference='svi', fast_psi=False, chunk=16, data=None):
    from oracle import model as om
    from rgp_amd.model import DeepAutoreg
    U, Y = data if data is not None else make_data(n_seq, T, u_dim, y_dim, seed, ragged)
    n_seq = len(Y)
    o = om.DeepAutoreg(list(wins), Y, U=U, U_win=U_win, nDims=list(nDims), X_variance=0.05, num_inducing=M, init='Y',
                       inference_method=inference, n_total=n_total, rng=np.random.default_rng(seed + 1), fast_psi=fast_psi,
                       chunk=chunk)
    rng = np.random.default_rng(seed + 2)
    for i in o.order:
        p = o.P[i]
        p.inv_l[:] = 0.3 + rng.random(p.inv_l.shape)
        p.variance = 0.7 + rng.random()
        p.noise = 0.1 + 0.3 * rng.random()
        if inference == 'svi':
            p.qU_W[:] = rng.standard_normal(p.qU_W.shape) * 0.2
            p.qU_a = 0.3
    for i in range(1, len(wins)):
        for s in range(n_seq):
            o.Xvar[i][s][:] = 0.02 + 0.1 * rng.random(o.Xvar[i][s].shape)
            o.Xmean[i][s][:] += 0.1 * rng.standard_normal(o.Xmean[i][s].shape)
    g = DeepAutoreg(list(wins), Y, U=U, U_win=U_win, nDims=list(nDims), X_variance=0.05, num_inducing=M, init='Y',
                    inference_method=inference, mb_inf_tot_data_size=n_total, seed=seed)
    g.load_from_oracle(o)
    return o, g


def worst_cond(o):
    """Largest 2-norm condition number of K_uu + 1e-6 I over the layers of an oracle model."""
    from oracle import psi_stats as ps
    worst = 1.0
    for i in o.order:
        p = o.P[i]
        ev = np.linalg.eigvalsh(ps.rbf_K_exact(p.variance, p.inv_l, p.Z) + 1e-6 * np.eye(p.Z.shape[0]))
        worst = max(worst, ev[-1] / ev[0])
    return worst


def rounding_sensitivity(o, seqs=None):
    """How far one unit in the last place of every parameter moves each gradient group of the ORACLE (relative to the group's
    largest entry).  Two correct fp64 evaluations that order their operations differently can differ by about this much, however
    each is written: it measures the conditioning of the whole map from parameters to gradients (the inverse of K_uu, the
    cancellation in dL/dZ), of which cond(K_uu) alone is only one factor."""
    rng = np.random.default_rng(12345)
    eps = np.finfo(float).eps
    _, G0 = o.elbo_and_grads(seqs)
    G0 = {k: np.array(v, dtype=float) for k, v in G0.items()}
    saved = []

    def nudge(a):
        saved.append((a, a.copy()))
        a *= 1.0 + eps * rng.choice([-1.0, 1.0], size=a.shape)

    for i in o.order:
        for name in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            a = getattr(o.P[i], name, None)
            if isinstance(a, np.ndarray):
                nudge(a)
    for i in range(1, len(o.Xmean)):
        for s in range(len(o.Xmean[i])):
            nudge(o.Xmean[i][s]); nudge(o.Xvar[i][s])
    _, G1 = o.elbo_and_grads(seqs)
    for a, b in saved:
        a[...] = b
    return {k: np.abs(np.asarray(G1[k], dtype=float) - v).max() / max(np.abs(v).max(), 1e-12) for k, v in G0.items()}


def compare(o, g, seqs=None, rtol=1e-9, sens=None):
    import torch
    eo, Go = o.elbo_and_grads(seqs)
    eg = float(g.evaluate(seqs).item())
    g.handle.sync()
    assert abs(eg - eo) <= 1e-11 * abs(eo), (eg, eo)
    Gg = g.grads_as_oracle_dict(seqs if seqs is not None else None)
    for k, v in Go.items():
        a, b = np.asarray(Gg[k], dtype=float), np.asarray(v, dtype=float)
        scale = max(np.abs(b).max(), 1e-12)
        tol = rtol if sens is None else max(rtol, 20.0 * sens[k])      # sens: rounding_sensitivity(o)
        assert np.abs(a.reshape(b.shape) - b).max() <= tol * scale, (k, np.abs(a.reshape(b.shape) - b).max(), scale, tol)
    return eo


@pytest.mark.parametrize('cfg', [
    dict(),                                                        # the reference tests' 2-hidden-layer shape
    dict(wins=(0, 3), nDims=(3, 2), n_seq=2),                      # 1 hidden layer
    dict(ragged=True, n_seq=4),                                    # sequences of different length
    dict(wins=(0, 5, 4, 3), nDims=(2, 1, 2, 1), n_seq=2, T=16, M=6, y_dim=2),   # 3 hidden layers, mixed windows
    dict(wins=(0, 20, 20), nDims=(16, 1, 1), n_seq=2, T=64, M=100, U_win=20, u_dim=1, y_dim=16),   # config-4 geometry
])
def test_elbo_and_gradients_match_oracle(cfg):
    o, g = build_pair(**cfg)
    compare(o, g)


@pytest.mark.parametrize('cfg', [
    dict(),
    dict(wins=(0, 3), nDims=(3, 2), n_seq=2),
    dict(ragged=True, n_seq=4),
    dict(wins=(0, 10), nDims=(1, 1), n_seq=1, T=120, M=20, U_win=10, u_dim=1, y_dim=1),      # Actuator-like (config 1 geometry)
    dict(wins=(0, 20, 20), nDims=(1, 1, 1), n_seq=1, T=150, M=50, U_win=20, u_dim=1, y_dim=1),   # Ballbeam-like (config 2)
])
def test_collapsed_bound_matches_oracle(cfg):
    """VarDTC (autoreg/inference/vardtc.py) on the device: ELBO and every gradient vs the oracle."""
    o, g = build_pair(inference=None, **cfg)
    compare(o, g)


def test_collapsed_bound_checkgrad_device():
    o, g = build_pair(n_seq=2, T=10, inference=None)
    x0 = g.get_optimizer_array()
    f0, g0 = g.objective_and_grad(x0)
    rng = np.random.default_rng(4)
    eps = 1e-6
    for j in rng.choice(x0.size, size=30, replace=False):
        xp = x0.copy(); xp[j] += eps
        xm = x0.copy(); xm[j] -= eps
        num = (g.objective_and_grad(xp)[0] - g.objective_and_grad(xm)[0]) / (2 * eps)
        assert abs(num - g0[j]) <= 1e-5 * max(1.0, abs(num)), (j, num, g0[j])


def test_minibatch_subset_permutation_additivity():
    o, g = build_pair(n_seq=6, T=12, n_total=6)
    e_all = compare(o, g)
    Gall = {k: np.array(v) for k, v in g.grads_as_oracle_dict().items()}
    compare(o, g, [4, 2, 0, 5, 1, 3])                               # permutation
    Gperm = g.grads_as_oracle_dict()
    for k in Gall:
        if len(k) == 2:
            assert np.allclose(Gperm[k], Gall[k], rtol=1e-11, atol=0) or np.abs(np.asarray(Gperm[k]) - Gall[k]).max() < 1e-11 * np.abs(Gall[k]).max()
    e1 = float(g.evaluate(slice(0, 3)).item())                      # contiguous slices (StdMemoryDataStreamer)
    G1 = {k: np.array(v) for k, v in g.grads_as_oracle_dict([0, 1, 2]).items()}
    e2 = float(g.evaluate(slice(3, 6)).item())
    G2 = {k: np.array(v) for k, v in g.grads_as_oracle_dict([3, 4, 5]).items()}
    assert np.isclose(e1 + e2, e_all, rtol=1e-13, atol=0)           # testing/minibatch_tests.py:288-296
    for k in Gall:
        if len(k) == 2:
            s = np.asarray(G1[k]) + np.asarray(G2[k])
            assert np.abs(s - Gall[k]).max() <= 1e-10 * max(np.abs(Gall[k]).max(), 1e-12), k
    compare(o, g, [1, 4])                                           # arbitrary subset with qU_ratio = 2/6


def test_checkgrad_device():
    o, g = build_pair(n_seq=2, T=10)
    x0 = g.get_optimizer_array()
    f0, g0 = g.objective_and_grad(x0)
    rng = np.random.default_rng(3)
    idx = rng.choice(x0.size, size=40, replace=False)
    eps = 1e-6
    for j in idx:
        xp = x0.copy(); xp[j] += eps
        xm = x0.copy(); xm[j] -= eps
        num = (g.objective_and_grad(xp)[0] - g.objective_and_grad(xm)[0]) / (2 * eps)
        assert abs(num - g0[j]) <= 1e-5 * max(1.0, abs(num)), (j, num, g0[j])


def test_svi_steps_track_oracle_adam():
    """A few Adam steps on the device follow the same trajectory as NumPy Adam on the oracle's gradients."""
    o, g = build_pair(n_seq=4, T=12, n_total=4)
    lr = 1e-2
    x = o.get_optimizer_array(); m = np.zeros_like(x); v = np.zeros_like(x)
    for step in (1, 2, 3):
        f, gr = o.objective_and_grad(x)
        gr = -gr
        m = 0.9 * m + 0.1 * gr; v = 0.999 * v + 0.001 * gr * gr
        x = x + lr * (m / (1 - 0.9 ** step)) / (np.sqrt(v / (1 - 0.999 ** step)) + 1e-8)
        eg = float(g.svi_step(slice(0, 4), lr=lr).item())
        assert abs(eg + f) <= 1e-9 * abs(f), (step, eg, f)
    o.set_optimizer_array(x)
    eo = o.elbo_and_grads()[0]
    eg = g.log_likelihood()
    assert abs(eg - eo) <= 1e-8 * abs(eo)


@pytest.mark.parametrize('inference', ['svi', None])
def test_freerun_matches_oracle(inference):
    """Predictive mean / variance of the moment-matching free-run (layers.py:1551-1571) vs the oracle, all layers."""
    o, g = build_pair(n_seq=3, T=16, inference=inference, M=6)
    compare(o, g)                                  # also fills the posteriors (woodbury_inv / woodbury_vector)
    rng = np.random.default_rng(8)
    U = rng.standard_normal((30, 2))
    want = o.freerun(U)
    got = g.freerun(U)
    g.handle.sync()
    for (wm, wv), (gm, gv) in zip(want, got):
        np.testing.assert_allclose(gm[0].cpu().numpy(), wm, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(gv[0].cpu().numpy(), wv, rtol=1e-7, atol=1e-12)
    # batched: two different control sequences at once give the same as one by one
    U2 = rng.standard_normal((30, 2))
    both = g.freerun([U, U2])
    single = g.freerun(U2)
    g.handle.sync()
    np.testing.assert_array_equal(both[-1][0][1].cpu().numpy(), single[-1][0][0].cpu().numpy())
    np.testing.assert_array_equal(both[-1][0][0].cpu().numpy(), got[-1][0][0].cpu().numpy())


def test_freerun_padded_width_20():
    """Free-run with a window of Q = 20 columns (the padded width 20 code path, as in the top layer of config 4)."""
    o, g = build_pair(n_seq=2, T=40, wins=(0, 10), nDims=(2, 1), M=12, U_win=10, u_dim=1, y_dim=2)
    compare(o, g)
    U = np.random.default_rng(3).standard_normal((50, 1))
    want, got = o.freerun(U), g.freerun(U)
    g.handle.sync()
    for (wm, wv), (gm, gv) in zip(want, got):
        np.testing.assert_allclose(gm[0].cpu().numpy(), wm, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(gv[0].cpu().numpy(), wv, rtol=1e-7, atol=1e-12)


def test_no_controls_and_wide_outputs():
    """U=None (the top layer sees only its own history, model.py:67-69) and a wide observed layer (D_y = 20, H = 2)."""
    from oracle import model as om
    from rgp_amd.model import DeepAutoreg
    rng = np.random.default_rng(5)
    Y = [np.tanh(np.cumsum(rng.standard_normal((18, 20)), 0) * 0.2) for _ in range(2)]
    for inference in ('svi', None):
        o = om.DeepAutoreg([0, 3, 2], Y, U=None, nDims=[20, 2, 1], X_variance=0.05, num_inducing=5, init='Y',
                           inference_method=inference, rng=np.random.default_rng(2), fast_psi=False)
        for i in o.order:
            o.P[i].inv_l[:] = 0.4 + rng.random(o.P[i].inv_l.shape)
            o.P[i].noise = 0.2
            if inference == 'svi':
                o.P[i].qU_W[:] = rng.standard_normal(o.P[i].qU_W.shape) * 0.2
                o.P[i].qU_a = 0.2
        g = DeepAutoreg([0, 3, 2], Y, U=None, nDims=[20, 2, 1], X_variance=0.05, num_inducing=5, init='Y',
                        inference_method=inference, seed=0)
        g.load_from_oracle(o)
        compare(o, g)


@pytest.mark.parametrize('n', [8, 4])
def test_full_size_additivity_config4(n):
    """BASELINE config-4 sizes (T=4096, D_y=16, M=100, wins=[0,20,20]): the size-independent identities the reference
    tests assert (testing/minibatch_tests.py:281-296) hold on the device: ELBO and global gradients are additive over a
    split of the minibatch and invariant under a permutation of it.  With 4 sequences the whole minibatch (16304 rows) runs
    the pair form of the psi2 pass and each half (8152 rows) the sweep kernels, so the split also checks one against the other;
    with 8 everything runs the pair form (short and long passes)."""
    from rgp_amd.synthetic import CONFIG4 as cfg, generate_sequences
    from rgp_amd.model import DeepAutoreg
    Us, Ys = generate_sequences(n, cfg['T'], cfg['D_y'], cfg['D_u'])
    m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                    init='Y', inference_method='svi', mb_inf_tot_data_size=n, seed=3)
    for L in m.layers:
        m.set_param(L, 'inv_l', np.full(L.Q, 0.25))
        m.set_param(L, 'noise', 0.05)
    G = m.n_global
    e_all = float(m.evaluate(slice(0, n)).item()); g_all = m.grad_array[:G].cpu().numpy().copy()
    e_a = float(m.evaluate(slice(0, n // 2)).item()); g_a = m.grad_array[:G].cpu().numpy().copy()
    e_b = float(m.evaluate(slice(n // 2, n)).item()); g_b = m.grad_array[:G].cpu().numpy().copy()
    e_p = float(m.evaluate(list(range(n))[::-1]).item()); g_p = m.grad_array[:G].cpu().numpy().copy()
    m.handle.sync()
    assert np.isfinite(e_all)
    assert abs((e_a + e_b) - e_all) <= 1e-12 * abs(e_all)
    assert abs(e_p - e_all) <= 1e-12 * abs(e_all)
    scale = np.abs(g_all).max()
    assert np.abs(g_a + g_b - g_all).max() <= 1e-9 * scale
    assert np.abs(g_p - g_all).max() <= 1e-9 * scale


def test_config4_geometry_matches_oracle_elbo():
    """Config-4 geometry (M=100, wins=[0,20,20], D_y=16) on a T=768 sequence: ELBO within 1e-9 of the CPU oracle
    (BASELINE.json asks for 1e-4) and the top layer's dL/dZ within 1e-7."""
    import os
    os.environ.setdefault('OMP_NUM_THREADS', '8')
    from rgp_amd.synthetic import CONFIG4 as cfg, generate_sequences
    from rgp_amd.model import DeepAutoreg
    from oracle import model as om
    Us, Ys = generate_sequences(1, 768, cfg['D_y'], cfg['D_u'])
    m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                    init='Y', inference_method='svi', mb_inf_tot_data_size=4, seed=3)
    for L in m.layers:
        m.set_param(L, 'inv_l', np.full(L.Q, 0.25))
        m.set_param(L, 'noise', 0.05)
    o = om.DeepAutoreg(cfg['wins'], [Ys[0]], U=[Us[0]], U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05,
                       num_inducing=cfg['M'], init='Y', inference_method='svi', n_total=4, rng=np.random.default_rng(0),
                       fast_psi=True, chunk=128)
    for L in m.layers:
        p = o.P[L.index]
        for nm in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            getattr(p, nm)[...] = m.view(L, nm).cpu().numpy()
        p.variance = float(m.view(L, 'variance').item()); p.noise = float(m.view(L, 'noise').item())
        p.qU_a = float(m.view(L, 'qU_a').item())
    e_o, G_o = o.elbo_and_grads()
    e_g = float(m.evaluate([0]).item())
    m.handle.sync()
    assert abs(e_g - e_o) <= 1e-9 * abs(e_o), (e_g, e_o)
    gz = m.view(m.layers[0], 'Z', grad=True).cpu().numpy()
    assert np.abs(gz - G_o[(m.layers[0].index, 'Z')]).max() <= 1e-7 * np.abs(gz).max()


def test_streamer_and_checkpoint(tmp_path):
    """Minibatch switch through a data streamer (model.py:209-267) and save/load of the flat parameter vector."""
    from rgp_amd.data_streamers import StdMemoryDataStreamer, TrivialDataStreamer
    o, g = build_pair(n_seq=4, T=12, n_total=4)
    x0 = g.get_optimizer_array()
    f_full, _ = g.objective_and_grad(x0)
    # trivial streamer: same likelihood as no streamer (testing/minibatch_tests.py:102-107)
    U, Y = make_data(4, 12)
    g.set_DataStreamer(TrivialDataStreamer(Y, U))
    f_triv, g_triv = g.objective_and_grad(x0)
    assert f_triv == f_full
    # two half minibatches add up to the full objective (KL is pre-scaled by qU_ratio)
    g.set_DataStreamer(StdMemoryDataStreamer(Y, U, 2))
    f1, g1 = g.objective_and_grad(x0)
    f2, g2 = g.objective_and_grad(x0)
    assert g.mb_inf_sample_idxes == [2, 3]
    assert abs((f1 + f2) - f_full) <= 1e-12 * abs(f_full)
    np.testing.assert_allclose(g1 + g2, g_triv, rtol=1e-8, atol=1e-9)
    # checkpoint round trip
    path = str(tmp_path / 'ckpt.npz')
    g.data_streamer = None
    g.save(path)
    g.param_array.mul_(1.01)
    assert g.objective_and_grad(g.get_optimizer_array())[0] != f_full
    g.load(path)
    assert g.objective_and_grad(g.get_optimizer_array())[0] == f_full


@pytest.mark.parametrize('M', [4, 140])
def test_collapsed_bound_data_parallel_two_shards(M):
    """Collapsed-bound data parallelism: two shards exchange only the sufficient statistics
    [psi0, yry, psi2, psi1Y, PsiR] per layer (the reference's dead allReduceArrays stub, svi_vardtc.py:65-67) and one sum of
    [ELBO, global gradients] reproduces the single-device collapsed bound.  Two handles emulate the two ranks.  M = 140 takes
    the blocked kernels and the blocked chain."""
    import torch
    from oracle import model as om
    from rgp_amd.model import DeepAutoreg
    U, Y = make_data(4, 13, seed=3)
    kw = dict(U_win=2, nDims=[3, 2, 3], X_variance=0.05, num_inducing=M, init='Y', inference_method=None)
    full = DeepAutoreg([0, 3, 3], Y, U=U, seed=0, **kw)
    A = DeepAutoreg([0, 3, 3], Y[:2], U=U[:2], seed=0, **kw)
    B = DeepAutoreg([0, 3, 3], Y[2:], U=U[2:], seed=0, **kw)
    rng = np.random.default_rng(1)
    for L in full.layers:
        full.set_param(L, 'inv_l', 0.3 + rng.random(L.Q))
        full.set_param(L, 'noise', 0.2)
    for m in (A, B):
        m.param_array[:m.n_global].copy_(full.param_array[:full.n_global])       # identical globals (same Z via the seed)
    e_full = float(full.evaluate().item())
    g_full = full.grad_array[:full.n_global].cpu().numpy().copy()
    # pass 1: record every layer's local statistics on both shards (a layer's statistics do not depend on other layers)
    rec = {}
    def recorder(tag):
        def f(stats, ntot, L):
            rec[(tag, L.index)] = (stats.clone(), ntot.clone())
        return f
    A.evaluate(stats_allreduce=recorder('A'), world=2)
    B.evaluate(stats_allreduce=recorder('B'), world=2)
    # pass 2: the "all-reduce" hands both shards the summed statistics
    def summed(stats, ntot, L):
        stats.copy_(rec[('A', L.index)][0] + rec[('B', L.index)][0])
        ntot.copy_(rec[('A', L.index)][1] + rec[('B', L.index)][1])
    eA = float(A.evaluate(stats_allreduce=summed, world=2).item()); gA = A.grad_array[:A.n_global].cpu().numpy().copy()
    eB = float(B.evaluate(stats_allreduce=summed, world=2).item()); gB = B.grad_array[:B.n_global].cpu().numpy().copy()
    A.handle.sync(); B.handle.sync(); full.handle.sync()
    assert abs((eA + eB) - e_full) <= 1e-11 * abs(e_full), (eA + eB, e_full)
    np.testing.assert_allclose(gA + gB, g_full, rtol=1e-8, atol=1e-9)
    # local (latent) gradients need no exchange
    np.testing.assert_allclose(A.latent(1, 0, grad=True).cpu().numpy(), full.latent(1, 0, grad=True).cpu().numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(B.latent(2, 1, var=True, grad=True).cpu().numpy(), full.latent(2, 3, var=True, grad=True).cpu().numpy(), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('inference', ['svi', None])
@pytest.mark.parametrize('cfg', [
    dict(n_seq=1, T=6, wins=(0, 3), nDims=(1, 1), M=1, U_win=2, u_dim=1, y_dim=1),          # one sequence, one inducing point, N = 4 / 1 rows
    dict(n_seq=2, T=5, wins=(0, 2, 2), nDims=(2, 1, 1), M=2, U_win=1, u_dim=1, y_dim=2),     # shortest usable sequences, U_win = 1
    dict(n_seq=1, T=40, wins=(0, 8), nDims=(4, 4), M=128, U_win=8, u_dim=4, y_dim=4),         # maximum sizes: M = 128, Q = 64
    dict(n_seq=3, T=9, wins=(0, 1, 1), nDims=(1, 1, 1), M=3, U_win=1, u_dim=1, y_dim=1),      # windows of one step: Q = 2 / 1
    dict(n_seq=2, T=30, wins=(0, 9, 3), nDims=(2, 2, 1), M=20, U_win=10, u_dim=1, y_dim=2),   # Q = 18 / 21 / 13: padded widths 20, 24, 16
])
def test_edge_geometries(cfg, inference):
    """Minimal, degenerate and maximal geometries of the layer step (both bounds)."""
    o, g = build_pair(inference=inference, **cfg)
    compare(o, g, rtol=1e-8)


def test_unsupported_sizes_fail_loudly():
    from rgp_amd._lib import RgpError
    from rgp_amd.model import DeepAutoreg
    U, Y = make_data(1, 40, u_dim=1, y_dim=1)
    m = DeepAutoreg([0, 3], Y, U=U, U_win=2, nDims=[1, 1], num_inducing=1025, inference_method='svi')   # M > 1024
    with pytest.raises(RgpError):
        m.evaluate()
    # Q = 70 > 64 is fine for both bounds and the unit entry points (tests/test_gpu_wide_windows.py, test_gpu_big.py); the entry
    # point that cannot allocate stops at 64
    import torch
    from rgp_amd.capi import Handle
    hd = Handle()
    z = lambda *sh: torch.ones(*sh, dtype=torch.float64, device='cuda')
    with pytest.raises(RgpError):
        hd.psi_forward_static(z(5, 70), z(5, 70), z(3, 70), z(1), z(70), z(5, 3), z(3, 3))
    U3, Y3 = make_data(1, 1200, u_dim=1, y_dim=1)
    m3 = DeepAutoreg([0, 1030], Y3, U=U3, U_win=30, nDims=[1, 1], num_inducing=4, inference_method='svi')        # Q = 1060 > 1024
    with pytest.raises(RgpError):
        m3.evaluate()


def test_deterministic_mode_is_bit_reproducible():
    """rgp_set_deterministic(1): the rotating-team psi2 kernel adds in a fixed order, so two evaluations agree bit for bit;
    the default packed-team kernel (fp64 LDS atomics in hardware order) agrees with it to rounding."""
    import torch
    o, g = build_pair(n_seq=4, T=300, wins=(0, 6, 6), nDims=(3, 1, 1), M=100, U_win=6, u_dim=1, y_dim=3, seed=9)
    g.handle.set_deterministic(True)
    try:
        e1 = float(g.evaluate().item()); g.handle.sync()
        G1 = g.grad_array.clone()
        e2 = float(g.evaluate().item()); g.handle.sync()
        G2 = g.grad_array.clone()
        assert e1 == e2 and torch.equal(G1, G2)
    finally:
        g.handle.set_deterministic(False)
    e3 = float(g.evaluate().item()); g.handle.sync()
    G3 = g.grad_array.clone()
    assert abs(e3 - e1) <= 1e-12 * abs(e1)
    assert float((G3 - G1).abs().max()) <= 1e-10 * float(G1.abs().max())


def test_pipelined_svi_steps_equal_plain_steps():
    """svi_step overlaps the next step's M x M chains with the running step and updates each layer's global parameters on the
    side stream (DeepAutoreg._svi_step_pipelined); the parameters after several steps must equal those of the plain
    evaluate -> Adam sequence, and evaluate / set_param in between must see consistent state."""
    _, a = build_pair(n_seq=4, T=40, M=6, seed=5, n_total=4)
    _, b = build_pair(n_seq=4, T=40, M=6, seed=5, n_total=4)
    b.pipeline = False
    for k in range(4):
        mb = slice(0, 2) if k % 2 == 0 else slice(2, 4)
        ea = float(a.svi_step(mb, lr=2e-2).item())
        eb = float(b.svi_step(mb, lr=2e-2).item())
        assert abs(ea - eb) <= 1e-11 * abs(eb), (k, ea, eb)
    ea = float(a.evaluate().item()); eb = float(b.evaluate().item())      # flushes the chains left in flight
    a.handle.sync(); b.handle.sync()
    assert abs(ea - eb) <= 1e-11 * abs(eb)
    assert float((a.param_array - b.param_array).abs().max()) <= 1e-11 * float(b.param_array.abs().max())
    L = a.layers[0]
    a.set_param(L, 'noise', 0.123); b.set_param(L, 'noise', 0.123)         # a parameter change between pipelined steps
    ea = float(a.svi_step(slice(0, 4), lr=1e-2).item()); eb = float(b.svi_step(slice(0, 4), lr=1e-2).item())
    a.handle.sync(); b.handle.sync()
    assert abs(ea - eb) <= 1e-11 * abs(eb)
    # lazy ELBO: no main-stream wait for the last layer's algebra inside the loop; values are read after a synchronisation
    import torch
    ea_l = [a.svi_step(slice(0, 4), lr=1e-2, lazy_elbo=True) for _ in range(3)]
    eb_l = [float(b.svi_step(slice(0, 4), lr=1e-2).item()) for _ in range(3)]
    torch.cuda.synchronize()
    assert abs(float(ea_l[-1].item()) - eb_l[-1]) <= 1e-11 * abs(eb_l[-1])
    a.handle.sync(); b.handle.sync()
    assert float((a.param_array - b.param_array).abs().max()) <= 1e-11 * float(b.param_array.abs().max())


def test_pipelined_ragged_smaller_minibatch():
    """Pipelined steps whose minibatch has FEWER rows than the one the early M x M chains were sized for (ragged sequences,
    a short trailing minibatch): the chain-only prepare and the layer step carve the workspace with different N, and every
    buffer the chain writes must sit at an N-independent address.  Parameters after the steps must equal the unpipelined
    ones (they depend on every gradient, not only on the ELBO)."""
    _, a = build_pair(n_seq=5, T=30, M=6, seed=11, n_total=5, ragged=True)
    _, b = build_pair(n_seq=5, T=30, M=6, seed=11, n_total=5, ragged=True)
    b.pipeline = False
    mbs = [slice(3, 5), slice(0, 2), slice(2, 3), slice(3, 5), slice(1, 2), [4, 2], [0]]
    for k, mb in enumerate(mbs):
        ea = float(a.svi_step(mb, lr=2e-2).item())
        eb = float(b.svi_step(mb, lr=2e-2).item())
        assert abs(ea - eb) <= 1e-11 * abs(eb), (k, ea, eb)
        a.handle.sync(); b.handle.sync()
        assert float((a.param_array - b.param_array).abs().max()) <= 1e-11 * float(b.param_array.abs().max()), k


def test_prefetched_next_minibatch_equals_plain_steps():
    """svi_step(next_seqs=...) prepares the following minibatch's row terms while the running step still owns the main stream
    (RGP_PREPARE_ROWS_EARLY); results must not change -- including a wrong hint, an overlapping hint and a hint followed by
    evaluate()."""
    import torch
    _, a = build_pair(n_seq=6, T=40, M=6, seed=7, n_total=6)
    _, b = build_pair(n_seq=6, T=40, M=6, seed=7, n_total=6)
    b.pipeline = False
    mbs = [slice(0, 2), slice(2, 4), slice(4, 6)]
    ea, eb = [], []
    for k in range(7):
        ea.append(a.svi_step(mbs[k % 3], lr=2e-2, lazy_elbo=True, next_seqs=mbs[(k + 1) % 3]))
        eb.append(float(b.svi_step(mbs[k % 3], lr=2e-2).item()))
    # wrong hint (the step that follows uses other sequences), then an overlapping hint (ignored), then a list hint
    ea.append(a.svi_step(mbs[1], lr=2e-2, lazy_elbo=True, next_seqs=mbs[0]))
    eb.append(float(b.svi_step(mbs[1], lr=2e-2).item()))
    ea.append(a.svi_step(mbs[2], lr=2e-2, lazy_elbo=True, next_seqs=slice(3, 5)))
    eb.append(float(b.svi_step(mbs[2], lr=2e-2).item()))
    ea.append(a.svi_step([0, 1], lr=2e-2, lazy_elbo=True, next_seqs=[4, 5]))
    eb.append(float(b.svi_step([0, 1], lr=2e-2).item()))
    ea.append(a.svi_step([4, 5], lr=2e-2, lazy_elbo=True, next_seqs=[2, 3]))
    eb.append(float(b.svi_step([4, 5], lr=2e-2).item()))
    a.handle.sync(); b.handle.sync()
    torch.cuda.synchronize()
    for k, (x, y) in enumerate(zip(ea, eb)):
        assert abs(float(x.item()) - y) <= 1e-11 * abs(y), (k, float(x.item()), y)
    e1 = float(a.evaluate().item()); e2 = float(b.evaluate().item())        # drops the prefetched rows and chains
    a.handle.sync(); b.handle.sync()
    assert abs(e1 - e2) <= 1e-11 * abs(e2)
    assert float((a.param_array - b.param_array).abs().max()) <= 1e-11 * float(b.param_array.abs().max())


def test_latent_adam_uneven_overlap_and_strided_slice():
    """Overlapping slices give sequences different step counts: each run of equal counts gets its own bias correction, and a
    strided slice updates exactly the sequences it names.  Checked against per-sequence index lists (one launch each)."""
    import torch
    o, a = build_pair(n_seq=6, T=12, n_total=6)
    o, b = build_pair(n_seq=6, T=12, n_total=6)
    a.pipeline = b.pipeline = False
    plan = [slice(0, 3), slice(2, 5), slice(0, 6, 2), slice(1, 6)]
    for sl in plan:
        a.svi_step(sl, lr=1e-2)
        b.svi_step(list(range(*sl.indices(6))), lr=1e-2)
    a.handle.sync(); b.handle.sync()
    np.testing.assert_allclose(a.param_array.cpu().numpy(), b.param_array.cpu().numpy(), rtol=1e-9, atol=1e-12)
    assert list(a._adam['t_loc']) == [2, 2, 4, 2, 3, 1]
