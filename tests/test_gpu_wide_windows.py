"""SVI layer steps with windows wider than 64 columns (the GEMM-shaped path of rgp_wideq_impl.h), against the oracle:
ELBO to 1e-11, every gradient group to 1e-9 of its largest entry, as for the narrow windows in test_gpu_model."""
import pytest

from test_gpu_model import build_pair, compare

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
