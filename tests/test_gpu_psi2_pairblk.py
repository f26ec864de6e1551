"""The pair form of the psi2 term for M > 128 inducing points (blocked form of rgp_amd/csrc/rgp_psi2_pair.hip: pair groups over
16 x 16 blocks of inducing rows, entered from the blocked layer step and the unit entry point) against the oracle and against
the blocked sweep kernels.

`Handle.set_psi2_pair(2)` forces it wherever its LDS image fits (padded Q <= 48; the default picks it from 2048 rows), so the
small shapes of test_gpu_big can be run through it.  Tolerances as there: unit gradients 1e-10 of the largest entry, ELBO 1e-11,
model gradients 1e-8."""
import numpy as np
import pytest

from test_gpu_big import dev, make, rel, _pair

pytestmark = pytest.mark.gpu


@pytest.fixture()
def HP():
    import torch
    from rgp_amd.capi import Handle
    assert torch.cuda.is_available()
    h = Handle()
    h.set_psi2_pair(2)
    return h


# (N, M, Q, controls): M just over the limit (9 column blocks, the last one with a single row), whole blocks (256), config-5's
# M = 500 (the last block has 4 rows), three sweep blocks (600); ragged last tile of 16 data rows; Q = 20 takes five pair tiles per
# wave, Q = 17 and 6 are padded, Q = 48 is the widest window whose LDS image fits; Q = 57 (padded 64) falls back to the sweeps
SHAPES = [(70, 129, 8, 2), (150, 256, 40, 20), (90, 500, 40, 20), (33, 600, 20, 0), (257, 130, 17, 0), (40, 300, 48, 3),
          (19, 145, 6, 0), (40, 300, 57, 3),
          (24, 1024, 20, 0), (50, 513, 12, 2)]     # the largest M (64 column blocks); a last block of ONE inducing row


@pytest.mark.parametrize('N,M,Q,nctrl', SHAPES)
def test_psi_backward_blocked_pair_form(HP, N, M, Q, nctrl):
    from oracle import psi_stats as ps
    H = HP
    mu, S, Z, var, il = make(N, M, Q, 57 + N, nctrl)
    rng = np.random.default_rng(8)
    w0 = rng.standard_normal(N)
    w1 = rng.standard_normal((N, M))
    w2 = rng.standard_normal((M, M))                                   # not symmetric on purpose
    args = (dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    ref = ps.PsiStatsFast(16).grads(var, il, Z, mu, S, w0, w1, w2)
    g = H.psi_backward(*args, w0.sum(), dev(H, w1), dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g[k].cpu().numpy().reshape(np.shape(ref[k])), ref[k]) < 1e-10, k
    ref2 = ps.PsiStatsFast(16).grads(var, il, Z, mu, S, 0 * w0, None, w2)
    g2 = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g2[k].cpu().numpy().reshape(np.shape(ref2[k])), ref2[k]) < 1e-10, k
    H.set_psi2_pair(0)                                                 # the blocked sweep kernels on the same handle
    g0 = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    H.set_psi2_pair(2)
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g2[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-11, k


def test_many_steps_and_row_ranges_per_task():
    """9000 rows: every (pair group, row range) task walks some 280 steps of 16 rows (both record buffers, all three T tiles, a
    ragged last tile), two row ranges add into the same C and pair sums, and a workgroup runs several tasks one after the other."""
    import torch
    from rgp_amd.capi import Handle
    from oracle import psi_stats as ps
    H = Handle()
    N, M, Q = 9000, 130, 20
    mu, S, Z, var, il = make(N, M, Q, 5, 4)
    w2 = np.random.default_rng(2).standard_normal((M, M))
    args = (dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    H.set_psi2_pair(2)
    g2 = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    ref2 = ps.PsiStatsFast(64).grads(var, il, Z, mu, S, np.zeros(N), None, w2)
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g2[k].cpu().numpy().reshape(np.shape(ref2[k])), ref2[k]) < 1e-10, k
    H.set_psi2_pair(0)
    g0 = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g2[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-11, k


@pytest.mark.parametrize('cfg', [
    dict(wins=(0, 3, 3), nDims=(3, 2, 3), n_seq=3, T=14, M=129),                                              # toy geometry, just over the limit
    dict(wins=(0, 20, 20), nDims=(64, 1, 1), n_seq=2, T=56, M=256, U_win=20, u_dim=1, y_dim=64, ragged=True),  # Q = 40 / 40 / 20, D = 64
    dict(wins=(0, 20, 20, 20), nDims=(64, 1, 1, 1), n_seq=1, T=70, M=500, U_win=20, u_dim=1, y_dim=64),        # config-5 geometry: 4 GP layers
])
def test_model_through_the_blocked_pair_form(cfg):
    """ELBO (the forward sum of psi2 enters it) and every gradient of the SVI bound with the pair form forced."""
    from test_gpu_model import compare
    o, g = _pair(**cfg)
    g.handle.set_psi2_pair(2)
    assert g.handle.psi2_pair_would_run(64, cfg['M'], 40)
    compare(o, g, rtol=1e-8)


def test_would_run_query_and_fallbacks(HP):
    """rgp_psi2_pair_would_run for the blocked layer step: mode 2 wherever the LDS image fits (padded Q <= 48), mode 1 from 2048
    rows, never in the deterministic mode."""
    H = HP
    assert H.psi2_pair_would_run(64, 500, 40) and H.psi2_pair_would_run(64, 1024, 20)
    assert not H.psi2_pair_would_run(64, 500, 57)            # padded to 64 columns: the LDS image does not fit
    H.set_psi2_pair(1)
    assert not H.psi2_pair_would_run(2047, 500, 40) and H.psi2_pair_would_run(2048, 500, 40)
    H.set_psi2_pair(0)
    assert not H.psi2_pair_would_run(100000, 500, 40)
    H.set_psi2_pair(2)
    H.set_deterministic(True)
    assert not H.psi2_pair_would_run(100000, 500, 40)
    H.set_deterministic(False)


def test_nan_input_reaches_the_gradients(HP):
    """A NaN in one window mean is not dropped by the exponential's clamp: the gradients of that row and the sums over the
    rows come out NaN, the other rows' gradients stay finite."""
    H = HP
    N, M, Q = 40, 130, 8
    mu, S, Z, var, il = make(N, M, Q, 3)
    mu[7, 2] = np.nan
    w2 = np.random.default_rng(1).standard_normal((M, M))
    g = H.psi_backward(dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il), 0.0, None, dev(H, w2))
    H.sync()
    assert np.isnan(g['dmu'].cpu().numpy()[7]).all()
    assert np.isnan(g['dZ'].cpu().numpy()).any()
    rest = np.delete(g['dmu'].cpu().numpy(), 7, axis=0)
    assert np.isfinite(rest).all()
