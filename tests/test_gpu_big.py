"""GPU parity of the blocked path for M > 128 inducing points (BASELINE config 5: M = 500, D_y = 64, three hidden layers):
unit pieces and whole-model ELBO / gradients against the oracle, at M in {129, 256, 300, 500, 600}, Q up to 57, D = 64.
Tolerances: ELBO 1e-9 relative, gradients 1e-8 of the largest entry of the group (fp64 throughout)."""
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
    return mu, S, Z, 0.5 + rng.random(), (0.3 + rng.random(Q)) * (8.0 / max(Q, 8))


# (N, M, Q, control columns): one inducing block with idle lanes (129), exactly one full block (256), two blocks (300, 500),
# three blocks (600); Q = 57 takes the one-column-per-wave forward kernel
SHAPES = [(70, 129, 8, 2), (150, 256, 40, 20), (40, 300, 57, 3), (90, 500, 40, 20), (33, 600, 20, 0),
          (60, 20, 80, 10), (45, 150, 130, 0)]      # windows wider than 64 columns (the pair-indexed GEMM forms), M below and above 128


@pytest.mark.parametrize('N,M,Q,nctrl', SHAPES)
def test_big_kuu_and_psi_forward(H, N, M, Q, nctrl):
    from oracle import psi_stats as ps
    mu, S, Z, var, il = make(N, M, Q, 11 + N, nctrl)
    K = H.kuu(dev(H, Z), dev(H, [var]), dev(H, il)).cpu().numpy()
    assert rel(K, ps.rbf_K_exact(var, il, Z) + 1e-6 * np.eye(M)) < 1e-13
    p1, p2 = H.psi_forward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    H.sync()
    assert rel(p1.cpu().numpy(), ps.psi1_direct(var, il, Z, mu, S)) < 1e-12
    ref2 = ps.PsiStatsFast(16).psi2(var, il, Z, mu, S)
    assert rel(p2.cpu().numpy(), ref2) < 1e-11
    assert np.array_equal(p2.cpu().numpy(), p2.cpu().numpy().T)


@pytest.mark.parametrize('N,M,Q,nctrl', SHAPES)
def test_big_psi_backward(H, N, M, Q, nctrl):
    from oracle import psi_stats as ps
    mu, S, Z, var, il = make(N, M, Q, 23 + N, nctrl)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(N)
    w1 = rng.standard_normal((N, M))
    w2 = rng.standard_normal((M, M))                                   # not symmetric on purpose
    ref = ps.PsiStatsFast(16).grads(var, il, Z, mu, S, w0, w1, w2)
    g = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), w0.sum(), dev(H, w1), dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g[k].cpu().numpy().reshape(np.shape(ref[k])), ref[k]) < 1e-10, k
    ref1 = ps.PsiStatsFast(16).grads(var, il, Z, mu, S, 0 * w0, w1, None)
    g1 = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), 0.0, dev(H, w1), None)
    ref2 = ps.PsiStatsFast(16).grads(var, il, Z, mu, S, 0 * w0, None, w2)
    g2 = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), 0.0, None, dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g1[k].cpu().numpy().reshape(np.shape(ref1[k])), ref1[k]) < 1e-10, k
        assert rel(g2[k].cpu().numpy().reshape(np.shape(ref2[k])), ref2[k]) < 1e-10, k


@pytest.mark.parametrize('M,D', [(129, 3), (300, 64), (500, 64), (640, 16)])
def test_big_svi_chain_against_oracle(H, M, D):
    """Blocked Cholesky / inverse factor / MFMA GEMM chain against the solve-based oracle (SVI_VarDTC.inference + comp_KL_qU)."""
    from oracle import inference as inf, psi_stats as ps
    rng = np.random.default_rng(M)
    Q, N = 12, 400
    Z = rng.standard_normal((M, Q)) * 1.5
    il, var, noise = np.full(Q, 0.5), 1.3, 0.05
    mu = rng.standard_normal((N, Q))
    S = 0.05 + 0.1 * rng.random((N, Q))
    P = ps.PsiStatsFast(16)
    psi1, psi2 = P.psi1(var, il, Z, mu, S), P.psi2(var, il, Z, mu, S)
    Kuu = ps.rbf_K_exact(var, il, Z)
    Y = rng.standard_normal((N, D))
    qm, W, a = rng.standard_normal((M, D)), 0.3 * rng.standard_normal((M, M)) / np.sqrt(M / 100.0), 0.5
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
    tol = 1e-9
    assert abs(c('bound')[0] - (logL - ratio * KL)) < tol * abs(logL)
    assert rel(c('dL_dKmm'), g['dL_dKmm'] - ratio * dKLK) < tol
    assert rel(c('dL_dpsi2'), g['dL_dpsi2']) < tol
    assert rel(c('Kinv_mu'), wvec) < tol
    assert rel(c('woodbury_inv'), winv) < tol
    assert rel(c('dL_dthetaL'), np.asarray(g['dL_dthetaL']).reshape(1)) < tol
    assert rel(c('gqU_m'), g['dL_dqU_mean'] - ratio * dKLm) < tol
    assert rel(c('gqU_W'), (Gv + Gv.T).dot(W)) < tol
    assert rel(c('gqU_a'), np.trace(Gv).reshape(1)) < tol


def test_big_not_pd_reports_global_pivot(H):
    from rgp_amd._lib import RgpError
    M, D = 200, 2
    K = np.eye(M)
    K[150, 150] = -1.0
    with pytest.raises(RgpError) as ei:
        H.svi_chain(10, dev(H, K), dev(H, [0.1]), dev(H, np.zeros((M, D))), dev(H, np.eye(M)), dev(H, [0.1]),
                    dev(H, [1.0]), dev(H, np.zeros((D, M))), dev(H, np.eye(M)), dev(H, [1.0]))
        H.sync()
    assert 'column 150' in str(ei.value)


def _pair(**kw):
    from test_gpu_model import build_pair
    return build_pair(fast_psi=True, chunk=16, **kw)


@pytest.mark.parametrize('cfg', [
    dict(wins=(0, 3, 3), nDims=(3, 2, 3), n_seq=3, T=14, M=129),                                              # toy geometry, just over the limit
    dict(wins=(0, 20, 20), nDims=(64, 1, 1), n_seq=2, T=56, M=129, U_win=20, u_dim=1, y_dim=64),               # Q = 40 / 40 / 20, D = 64
    dict(wins=(0, 20, 20), nDims=(64, 1, 1), n_seq=2, T=56, M=256, U_win=20, u_dim=1, y_dim=64, ragged=True),
    dict(wins=(0, 20, 20, 20), nDims=(64, 1, 1, 1), n_seq=1, T=70, M=500, U_win=20, u_dim=1, y_dim=64),        # config-5 geometry: 4 GP layers
])
def test_big_elbo_and_gradients_match_oracle(cfg):
    from test_gpu_model import compare
    o, g = _pair(**cfg)
    compare(o, g, rtol=1e-8)


def test_big_minibatch_additivity_and_steps():
    """M = 160: additivity over a split of the minibatch (testing/minibatch_tests.py:288-296) and pipelined = plain SVI steps
    (the blocked layer step runs on the main stream; parameter updates on the side stream must be ordered behind it)."""
    o, g = _pair(n_seq=4, T=16, M=160, n_total=4)
    _, g2 = _pair(n_seq=4, T=16, M=160, n_total=4)
    g2.pipeline = False
    G = g.n_global
    e_all = float(g.evaluate().item()); g_all = g.grad_array[:G].cpu().numpy().copy()
    e_a = float(g.evaluate(slice(0, 2)).item()); g_a = g.grad_array[:G].cpu().numpy().copy()
    e_b = float(g.evaluate(slice(2, 4)).item()); g_b = g.grad_array[:G].cpu().numpy().copy()
    g.handle.sync()
    assert abs((e_a + e_b) - e_all) <= 1e-12 * abs(e_all)
    assert np.abs(g_a + g_b - g_all).max() <= 1e-9 * np.abs(g_all).max()
    for k in range(3):
        mb = slice(0, 2) if k % 2 == 0 else slice(2, 4)
        ea = float(g.svi_step(mb, lr=1e-2).item())
        eb = float(g2.svi_step(mb, lr=1e-2).item())
        assert abs(ea - eb) <= 1e-10 * abs(eb), (k, ea, eb)
    g.handle.sync(); g2.handle.sync()
    assert float((g.param_array - g2.param_array).abs().max()) <= 1e-10 * float(g2.param_array.abs().max())


def test_big_workspace_bytes_and_limits():
    from rgp_amd import _lib
    from rgp_amd._lib import RgpError
    from rgp_amd.model import DeepAutoreg
    from test_gpu_model import make_data
    lib = _lib.load()
    assert lib.rgp_workspace_bytes(65516, 500, 40, 64) > 3 * 65516 * 500 * 8
    U, Y = make_data(1, 40, u_dim=1, y_dim=1)
    m = DeepAutoreg([0, 3], Y, U=U, U_win=2, nDims=[1, 1], num_inducing=1025, inference_method='svi')   # M > 1024
    with pytest.raises(RgpError):
        m.evaluate()
    m = DeepAutoreg([0, 3], Y, U=U, U_win=2, nDims=[1, 1], num_inducing=1025, inference_method=None)   # collapsed bound likewise
    with pytest.raises(RgpError):
        m.evaluate()
    U, Y = make_data(1, 30, u_dim=1, y_dim=2049)                       # more output columns than the blocked step takes
    for inference in ('svi', None):
        m = DeepAutoreg([0, 3], Y, U=U, U_win=2, nDims=[2049, 1], num_inducing=129, inference_method=inference)
        with pytest.raises(RgpError, match='D must be between'):
            m.evaluate()


@pytest.mark.parametrize('cfg', [
    dict(wins=(0, 3, 3), nDims=(3, 2, 3), n_seq=3, T=14, M=129),                                              # toy geometry, just over the limit
    dict(wins=(0, 20, 20), nDims=(5, 1, 1), n_seq=2, T=56, M=200, U_win=20, u_dim=1, y_dim=5, ragged=True),    # Q = 40 / 40 / 20
    dict(wins=(0, 10), nDims=(64, 2), n_seq=1, T=60, M=300, U_win=10, u_dim=1, y_dim=64),                      # D = 64 observed, two-dimensional hidden layer
])
def test_big_collapsed_bound_matches_oracle(cfg):
    """VarDTC (autoreg/inference/vardtc.py:88-208) with more than 128 inducing points: blocked psi kernels around the blocked
    chain (Cholesky of K_uu and of I + A, their inverses, every product an MFMA GEMM)."""
    from test_gpu_model import compare
    o, g = _pair(inference=None, **cfg)
    compare(o, g, rtol=1e-8)


@pytest.mark.parametrize('inference', ['svi', None])
@pytest.mark.parametrize('M', [129, 600])
def test_big_freerun_matches_oracle(M, inference):
    """Moment-matching free-run (layers.py:1551-1571) with M > 128: the m' sweep split into parts (M <= 256) and inducing rows
    in blocks of 512 (M > 512)."""
    from test_gpu_model import compare
    o, g = _pair(n_seq=2, T=16, M=M, inference=inference)
    # Hundreds of inducing points in 6 to 15 input dimensions: K_uu + 1e-6 I is ill conditioned (cond ~ 1e5 here), and the two
    # sides factor it in different orders (blocked Cholesky and explicit inverse factor here, one triangular solve per use in the
    # oracle).  Both keep dL/dK_uu in the congruence form Linv^T (...) Linv, whose error cancels in the sum that yields dL/dZ
    # (tools/chain_accuracy.py compares both with an extended-precision evaluation: each is good to a small multiple of
    # eps * cond).  The gate is 50 eps cond, with the condition number computed here, and never tighter than the 1e-8 of the
    # file header.
    from test_gpu_model import worst_cond, rounding_sensitivity
    cond = worst_cond(o)
    sens = rounding_sensitivity(o)
    rtol = max(1e-8, 50.0 * np.finfo(float).eps * cond)
    print('M = %d %s: cond(K_uu + 1e-6 I) = %.3g, gradient gate %.2g; one ulp of the inputs moves the oracle\'s gradients by up to %.2g'
          % (M, inference, cond, rtol, max(sens.values())))
    assert cond < 1e7                                   # beyond that the gate would say little
    compare(o, g, rtol=rtol)
    U = np.random.default_rng(8).standard_normal((12, 2))
    want, got = o.freerun(U), g.freerun(U)
    g.handle.sync()
    for (wm, wv), (gm, gv) in zip(want, got):
        np.testing.assert_allclose(gm[0].cpu().numpy(), wm, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(gv[0].cpu().numpy(), wv, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize('N,M,Q,nctrl', [(150, 256, 40, 20), (90, 500, 40, 20), (33, 600, 20, 0), (40, 300, 57, 3)])
def test_big_psi2_forward_on_the_matrix_pipe(H, monkeypatch, N, M, Q, nctrl):
    """The alternative psi2 forward kernel (v_mfma_f64_16x16x4 over 32 x 32 blocks of unordered pairs, RGP_PSI2_FWD_MFMA=1)
    against the oracle and against the default kernel."""
    from oracle import psi_stats as ps
    mu, S, Z, var, il = make(N, M, Q, 31 + N, nctrl)
    args = (dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    _, p2_default = H.psi_forward(*args)
    H.sync()
    monkeypatch.setenv('RGP_PSI2_FWD_MFMA', '1')
    _, p2 = H.psi_forward(*args)
    H.sync()
    ref2 = ps.PsiStatsFast(16).psi2(var, il, Z, mu, S)
    assert rel(p2.cpu().numpy(), ref2) < 1e-11
    assert rel(p2.cpu().numpy(), p2_default.cpu().numpy()) < 1e-12
    assert np.array_equal(p2.cpu().numpy(), p2.cpu().numpy().T)


def test_full_size_additivity_config5():
    """BASELINE config-5 sizes (T = 65536, D_y = 64, M = 500, wins = [0, 20, 20, 20]) on two sequences: the row-chunk x
    inducing-block grids and the multi-GB workspace of the real size.  The size-independent identities the reference tests
    assert (testing/minibatch_tests.py:281-296) hold: ELBO and global gradients are additive over a split of the minibatch
    and invariant under a permutation of it."""
    from rgp_amd.synthetic import CONFIG5 as cfg, generate_sequences
    from rgp_amd.model import DeepAutoreg
    n = 2
    Us, Ys = generate_sequences(n, cfg['T'], cfg['D_y'], cfg['D_u'])
    m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                    init='Y', inference_method='svi', mb_inf_tot_data_size=n, seed=3)
    for L in m.layers:
        m.set_param(L, 'inv_l', np.full(L.Q, 0.25))
        m.set_param(L, 'noise', 0.05)
    G = m.n_global
    e_all = float(m.evaluate(slice(0, n)).item()); g_all = m.grad_array[:G].cpu().numpy().copy()
    e_a = float(m.evaluate(slice(0, 1)).item()); g_a = m.grad_array[:G].cpu().numpy().copy()
    e_b = float(m.evaluate(slice(1, 2)).item()); g_b = m.grad_array[:G].cpu().numpy().copy()
    e_p = float(m.evaluate([1, 0]).item()); g_p = m.grad_array[:G].cpu().numpy().copy()
    m.handle.sync()
    scale = np.abs(g_all).max()
    print('config 5, 2 x 65536 rows: ELBO %.6e, split %.2e, permutation %.2e; gradients split %.2e, permutation %.2e of the largest'
          % (e_all, abs(e_a + e_b - e_all) / abs(e_all), abs(e_p - e_all) / abs(e_all), np.abs(g_a + g_b - g_all).max() / scale,
             np.abs(g_p - g_all).max() / scale))
    assert np.isfinite(e_all) and np.isfinite(g_all).all()
    assert abs((e_a + e_b) - e_all) <= 1e-11 * abs(e_all)
    assert abs(e_p - e_all) <= 1e-11 * abs(e_all)
    assert np.abs(g_a + g_b - g_all).max() <= 1e-8 * scale
    assert np.abs(g_p - g_all).max() <= 1e-8 * scale


def test_config5_geometry_matches_oracle():
    """Config-5 geometry (M = 500, D_y = 64, four GP layers, windows of 20) on one T = 276 sequence against the CPU oracle."""
    from test_gpu_model import compare
    o, g = _pair(wins=(0, 20, 20, 20), nDims=(64, 1, 1, 1), n_seq=1, T=276, M=500, U_win=20, u_dim=1, y_dim=64)
    compare(o, g, rtol=1e-8)
