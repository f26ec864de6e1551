"""The pair form of the fused psi2 pass (fp64 matrix pipe, rgp_amd/csrc/rgp_psi2_pair.hip) against the oracle.

`Handle.set_psi2_pair(2)` forces it wherever its LDS image fits (the default picks it only for M >= 48 and >= 4000 rows), so the
small shapes of the other tests can be run through it.  Tolerances as in test_gpu_kernels / test_gpu_model: gradients of the
unit entry point 1e-10 of the largest entry, ELBO 1e-11, model gradients 1e-9."""
import numpy as np
import pytest

from test_gpu_kernels import dev, make, rel
from test_gpu_model import build_pair, compare

pytestmark = pytest.mark.gpu


@pytest.fixture()
def HP():
    import torch
    from rgp_amd.capi import Handle
    assert torch.cuda.is_available()
    h = Handle()
    h.set_psi2_pair(2)
    return h


# (N, M, Q, controls): ragged last tile of 16 rows, odd and even M (the half-length last diagonal), one pair group and several,
# padded windows (Q = 6 -> 8, 57 -> 64), Q = 20 as it stands, the largest M the LDS image takes
SHAPES = [(13, 5, 3, 0), (70, 7, 6, 2), (257, 33, 20, 5), (300, 100, 40, 20), (129, 64, 8, 0), (131, 112, 57, 3),
          (1000, 50, 40, 20), (48, 101, 17, 0)]


@pytest.mark.parametrize('N,M,Q,nctrl', SHAPES)
def test_psi_backward_pair_form(HP, N, M, Q, nctrl):
    from oracle import psi_stats as ps
    H = HP
    mu, S, Z, var, il = make(N, M, Q, 41 + N, nctrl)
    rng = np.random.default_rng(6)
    w0 = rng.standard_normal(N)
    w1 = rng.standard_normal((N, M))
    w2 = rng.standard_normal((M, M))                                   # not symmetric on purpose
    args = (dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    ref = ps.PsiStatsFast(64).grads(var, il, Z, mu, S, w0, w1, w2)
    g = H.psi_backward(*args, w0.sum(), dev(H, w1), dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g[k].cpu().numpy().reshape(np.shape(ref[k])), ref[k]) < 1e-10, k
    ref2 = ps.PsiStatsFast(64).grads(var, il, Z, mu, S, 0 * w0, None, w2)
    g2 = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g2[k].cpu().numpy().reshape(np.shape(ref2[k])), ref2[k]) < 1e-10, k
    # and against the sweep kernels on the same handle
    H.set_psi2_pair(0)
    g0 = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    H.set_psi2_pair(2)
    for k in ('dvar', 'dinv_l', 'dZ', 'dmu', 'dS'):
        assert rel(g2[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-11, k


@pytest.mark.parametrize('cfg', [
    dict(),                                                        # the reference tests' 2-hidden-layer shape (M = 4)
    dict(ragged=True, n_seq=4),
    dict(wins=(0, 5, 4, 3), nDims=(2, 1, 2, 1), n_seq=2, T=16, M=6, y_dim=2),
    dict(wins=(0, 20, 20), nDims=(16, 1, 1), n_seq=2, T=64, M=100, U_win=20, u_dim=1, y_dim=16),   # config-4 geometry
    dict(wins=(0, 20, 20), nDims=(1, 1, 1), n_seq=2, T=300, M=50, U_win=20, u_dim=1, y_dim=1),
])
def test_model_through_the_pair_form(cfg):
    """ELBO (the forward sum of psi2 enters it) and every gradient of the SVI bound with the pair form forced."""
    o, g = build_pair(**cfg)
    g.handle.set_psi2_pair(2)
    compare(o, g)


def test_many_steps_per_workgroup_and_minibatch():
    """A long sequence: every workgroup walks many steps of 16 rows (both record buffers, both output tiles, a ragged last
    tile); then a minibatch of two of three ragged sequences (qU_ratio = 2/3)."""
    o, g = build_pair(wins=(0, 20, 20), nDims=(16, 1, 1), n_seq=1, T=7303, M=100, U_win=20, u_dim=1, y_dim=16, fast_psi=True,
                      chunk=512)
    g.handle.set_psi2_pair(2)
    compare(o, g)
    o, g = build_pair(wins=(0, 6, 6), nDims=(3, 2, 2), n_seq=3, T=4001, M=12, U_win=6, ragged=True, n_total=3, fast_psi=True,
                      chunk=512)
    g.handle.set_psi2_pair(2)
    compare(o, g, seqs=[0, 2])


@pytest.mark.parametrize('mode', [0, 2])
def test_nan_input_reaches_psi2_and_its_gradients(HP, mode):
    """A NaN in one window mean is not dropped by the exponential's clamp, in the sweep kernels (mode 0) and in the pair form
    (mode 2): psi2 and the gradients of that row come out NaN."""
    H = HP
    H.set_psi2_pair(mode)
    N, M, Q = 40, 20, 8
    mu, S, Z, var, il = make(N, M, Q, 3)
    mu[7, 2] = np.nan
    w2 = np.random.default_rng(1).standard_normal((M, M))
    args = (dev(H, mu), dev(H, S), dev(H, Z), dev(H, [var]), dev(H, il))
    _, p2 = H.psi_forward(*args)
    g = H.psi_backward(*args, 0.0, None, dev(H, w2))
    H.sync()
    H.set_psi2_pair(2)
    assert np.isnan(p2.cpu().numpy()).all()
    assert np.isnan(g['dmu'].cpu().numpy()[7]).all()
    assert np.isnan(g['dZ'].cpu().numpy()).any()
    assert np.isfinite(g['dmu'].cpu().numpy()[:7]).all()
