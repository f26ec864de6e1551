"""SVI layer steps with windows wider than 64 columns (the GEMM-shaped path of rgp_wideq_impl.h), against the oracle:
ELBO to 1e-11, every gradient group to 1e-9 of its largest entry, as for the narrow windows in test_gpu_model."""
import pytest

from test_gpu_model import build_pair, compare, worst_cond

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cfg', [
    # every layer wide: Q = 80 (top hidden), 120 (middle), 80 (observed); one chunk of window columns each
    dict(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=12, U_win=40, u_dim=1, y_dim=2),
    # several chunks: Q = 150 + 2 * 75 = 300 in the middle layer, 225 on top, 150 below; ragged sequences, a minibatch
    dict(wins=(0, 75, 75), nDims=(3, 2, 2), n_seq=3, T=110, M=9, U_win=75, u_dim=1, y_dim=3, ragged=True, n_total=5),
    # a wide layer between narrow ones (Q = 8 / 66 / 30) and M = 100
    dict(wins=(0, 30, 4), nDims=(2, 2, 1), n_seq=2, T=50, M=100, U_win=4, u_dim=1, y_dim=2),
])
def test_wide_windows_match_oracle(cfg):
    o, g = build_pair(fast_psi=True, chunk=64, **cfg)
    compare(o, g)


def test_wide_window_svi_steps_run():
    """Adam steps through the wide path (pipelining falls back to whole layer steps on the main stream)."""
    _, g = build_pair(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=4, T=70, M=12, U_win=40, u_dim=1, y_dim=2, n_total=4)
    e0 = float(g.evaluate().item())
    for k in range(5):
        g.svi_step(slice(0, 4), lr=1e-3)
    g.handle.sync()
    e1 = float(g.evaluate().item())
    assert e1 > e0


@pytest.mark.parametrize('cfg', [
    dict(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=12, U_win=40, u_dim=1, y_dim=2),       # Q = 80 / 120 / 80
    dict(wins=(0, 20, 20), nDims=(16, 16, 16), n_seq=1, T=45, M=100, U_win=20, u_dim=1, y_dim=16),  # Q = 340 / 640 / 320, D = 16
    dict(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=150, U_win=40, u_dim=1, y_dim=2),      # two tiles of inducing rows
    dict(wins=(0, 30, 4), nDims=(2, 2, 1), n_seq=2, T=50, M=260, U_win=4, u_dim=1, y_dim=2),        # three tiles; Q = 8 / 66 / 30
])
def test_wide_window_freerun_matches_oracle(cfg):
    """Moment-matching free-run through layers with wide windows (layers.py:1537-1571): the generic-Q kernel vs the oracle."""
    import numpy as np
    o, g = build_pair(fast_psi=True, chunk=8, **cfg)
    # many inducing points in few input dimensions: the gate is 1e-9 / 1e-8, or 50 eps cond(K_uu + 1e-6 I) where that is more
    # (test_gpu_big.test_big_freerun_matches_oracle)
    cond = worst_cond(o)
    rtol = max(1e-9 if cfg['M'] <= 128 else 1e-8, 50.0 * np.finfo(float).eps * cond)
    print('M = %d: cond(K_uu + 1e-6 I) = %.3g, gradient gate %.2g' % (cfg['M'], cond, rtol))
    assert cond < 1e7
    compare(o, g, rtol=rtol)                                    # also fills the posteriors
    U = np.random.default_rng(9).standard_normal((cfg['U_win'] + 25, cfg['u_dim']))
    want, got = o.freerun(U), g.freerun(U)
    g.handle.sync()
    loose = cfg['M'] > 128        # hundreds of inducing points in a few input dimensions: cond(K_uu) is at the jitter's limit
    for (wm, wv), (gm, gv) in zip(want, got):
        np.testing.assert_allclose(gm[0].cpu().numpy(), wm, rtol=1e-7 if loose else 1e-8, atol=1e-9 if loose else 1e-10)
        np.testing.assert_allclose(gv[0].cpu().numpy(), wv, rtol=1e-6 if loose else 1e-7, atol=1e-10 if loose else 1e-12)


@pytest.mark.parametrize('cfg', [
    # more than 128 inducing points with wide windows (blocked chain + pair products): Q = 80 / 120 / 80, M = 150
    dict(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=150, U_win=40, u_dim=1, y_dim=2),
    dict(wins=(0, 30, 4), nDims=(2, 2, 1), n_seq=2, T=50, M=260, U_win=4, u_dim=1, y_dim=2),      # Q = 8 / 66 / 30, M = 260
])
def test_wide_windows_with_many_inducing_points(cfg):
    o, g = build_pair(fast_psi=True, chunk=16, **cfg)
    compare(o, g, rtol=1e-8)


def test_wide_windows_in_row_chunks(monkeypatch):
    """The N x P weights held for 37 data rows at a time (psi2 and C accumulate over the chunks, the last chunk is ragged)."""
    monkeypatch.setenv('RGP_WQ_CHUNK_ROWS', '37')
    o, g = build_pair(fast_psi=True, chunk=64, wins=(0, 75, 75), nDims=(3, 2, 2), n_seq=3, T=110, M=9, U_win=75, u_dim=1, y_dim=3,
                      ragged=True, n_total=5)
    compare(o, g)
    o, g = build_pair(fast_psi=True, chunk=16, wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=150, U_win=40, u_dim=1, y_dim=2)
    compare(o, g, rtol=1e-8)


@pytest.mark.parametrize('cfg', [
    dict(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=12, U_win=40, u_dim=1, y_dim=2),        # Q = 80 / 120 / 80
    dict(wins=(0, 30, 4), nDims=(2, 2, 1), n_seq=2, T=50, M=100, U_win=4, u_dim=1, y_dim=2),         # a wide layer between narrow ones
    dict(wins=(0, 40, 40), nDims=(2, 2, 1), n_seq=2, T=70, M=150, U_win=40, u_dim=1, y_dim=2),       # with the blocked chain
])
def test_wide_windows_collapsed_bound(cfg):
    """VarDTC with windows wider than 64 columns: the pair-indexed Gram + exp pass runs once for psi2 before the chain and the
    full set of products after it."""
    o, g = build_pair(fast_psi=True, chunk=16, inference=None, **cfg)
    compare(o, g, rtol=1e-8)


def test_full_size_wide_windows_cross_the_weight_chunk():
    """Config-4 sizes with 16-dimensional hidden layers (windows of 340 / 640 / 320 columns, M = 100): 40 sequences of
    T = 4096 are 163040 data rows, more than the 159470 rows whose N x P weights the 6 GiB buffer holds at once, so the
    whole minibatch runs in two row chunks while each half runs in one.  ELBO and global gradients are additive over the
    split and invariant under a permutation (testing/minibatch_tests.py:281-296)."""
    import numpy as np
    from rgp_amd.synthetic import CONFIG4W as cfg, generate_sequences
    from rgp_amd.model import DeepAutoreg
    n = 40
    Us, Ys = generate_sequences(n, cfg['T'], cfg['D_y'], cfg['D_u'])
    m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                    init='Y', inference_method='svi', mb_inf_tot_data_size=n, seed=3)
    for L in m.layers:
        m.set_param(L, 'inv_l', np.full(L.Q, 4.0 / L.Q))
        m.set_param(L, 'noise', 0.05)
    G = m.n_global
    e_all = float(m.evaluate(slice(0, n)).item()); g_all = m.grad_array[:G].cpu().numpy().copy()
    e_a = float(m.evaluate(slice(0, n // 2)).item()); g_a = m.grad_array[:G].cpu().numpy().copy()
    e_b = float(m.evaluate(slice(n // 2, n)).item()); g_b = m.grad_array[:G].cpu().numpy().copy()
    e_p = float(m.evaluate(list(range(n))[::-1]).item()); g_p = m.grad_array[:G].cpu().numpy().copy()
    m.handle.sync()
    scale = np.abs(g_all).max()
    print('config 4 wide, 40 x 4096 rows: ELBO %.6e, split %.2e, permutation %.2e; gradients split %.2e, permutation %.2e of the largest'
          % (e_all, abs(e_a + e_b - e_all) / abs(e_all), abs(e_p - e_all) / abs(e_all), np.abs(g_a + g_b - g_all).max() / scale,
             np.abs(g_p - g_all).max() / scale))
    assert np.isfinite(e_all) and np.isfinite(g_all).all()
    assert abs((e_a + e_b) - e_all) <= 1e-11 * abs(e_all)
    assert abs(e_p - e_all) <= 1e-11 * abs(e_all)
    assert np.abs(g_a + g_b - g_all).max() <= 1e-9 * scale
    assert np.abs(g_p - g_all).max() <= 1e-9 * scale
