"""BASELINE configs 1 and 2 at their own geometry and on their own data (tests/golden/sysid_data.npz, written by
tests/golden/make_sysid_data.py from the reference's data files): device against the oracle.
  config 1: Actuator, one hidden layer, wins [0, 10], U_win 10, M = 20, the 512-sample training split, full batch, collapsed bound
            (examples/IdentificationTraining_Actuator.ipynb; split and windows: autoreg/benchmark/tasks.py:141-158)
  config 2: Ballbeam, two hidden layers, wins [0, 20, 20], U_win 20, M = 50, the 500-sample training split
            (examples/IdentificationTraining_Ballbeam.ipynb overrides the task's window of 10 to 20)
ELBO to 1e-11, every gradient group to 1e-9 of its largest entry; free-run over the held-out split: means to 1e-8, variances to
1e-7 (layers.py:1537-1571)."""
import os

import numpy as np
import pytest

from test_gpu_model import build_pair, compare

pytestmark = pytest.mark.gpu


def _series(name):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sysid_data.npz'))
    u, y, k = z[name + '_u'], z[name + '_y'], int(z[name + '_split'])
    mu_u, sd_u, mu_y, sd_y = u[:k].mean(), u[:k].std(), y[:k].mean(), y[:k].std()      # normalised by the training split
    u, y = (u - mu_u) / sd_u, (y - mu_y) / sd_y
    return (u[:k], y[:k]), (u[k:], y[k:])


CONFIGS = {
    'actuator': dict(wins=(0, 10), nDims=(1, 1), M=20, U_win=10),
    'ballbeam': dict(wins=(0, 20, 20), nDims=(1, 1, 1), M=50, U_win=20),
}


@pytest.mark.parametrize('name', ['actuator', 'ballbeam'])
@pytest.mark.parametrize('inference', [None, 'svi'])
def test_config_geometry_on_its_data(name, inference):
    (u, y), (u_test, y_test) = _series(name)
    o, g = build_pair(data=([u], [y]), inference=inference, fast_psi=True, chunk=128, seed=5, **CONFIGS[name])
    compare(o, g)
    # free-run over the held-out controls from the posterior the step above left behind
    want, got = o.freerun(u_test), g.freerun(u_test)
    g.handle.sync()
    for (wm, wv), (gm, gv) in zip(want, got):
        np.testing.assert_allclose(gm[0].cpu().numpy(), wm, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(gv[0].cpu().numpy(), wv, rtol=1e-7, atol=1e-12)
    assert np.isfinite(got[-1][0][0].cpu().numpy()).all() and len(got[-1][0][0]) > 0 and y_test.shape[0] > 0


def test_actuator_lbfgs_trajectory_and_heldout_rmse():
    """Config 1 end to end, as the reference's notebook does it (m.optimize('bfgs') then free-run on the held-out split,
    autoreg/benchmark/run.py:51-58): L-BFGS-B on the device objective follows L-BFGS-B on the oracle's objective (same start,
    same iteration limit) to 1e-6 of the final objective, and the held-out free-run RMSE of the two agrees to 1e-4."""
    from scipy.optimize import fmin_l_bfgs_b
    (u, y), (u_test, y_test) = _series('actuator')
    o, g = build_pair(data=([u], [y]), inference=None, fast_psi=True, chunk=128, seed=5, **CONFIGS['actuator'])
    x0 = o.get_optimizer_array()
    np.testing.assert_allclose(g.get_optimizer_array(), x0, rtol=0, atol=1e-12)
    f0 = o.objective_and_grad(x0)[0]
    xo, fo, _ = fmin_l_bfgs_b(o.objective_and_grad, x0, maxfun=30, maxiter=25)
    xg, fg, _ = fmin_l_bfgs_b(g.objective_and_grad, x0, maxfun=30, maxiter=25)
    assert fo < f0 and fg < f0
    assert abs(fg - fo) <= 1e-6 * abs(fo), (fg, fo)
    o.set_optimizer_array(xo); o.elbo_and_grads()               # posteriors at the optimised parameters
    g.set_optimizer_array(xg); g.evaluate(); g.handle.sync()
    U_win = CONFIGS['actuator']['U_win']
    pm_o = o.freerun(u_test)[-1][0]
    pm_g = g.freerun(u_test)[-1][0][0].cpu().numpy()
    yt = y_test[-pm_o.shape[0]:] if pm_o.shape[0] <= y_test.shape[0] else y_test
    k = min(len(yt), pm_o.shape[0])
    rmse_o = float(np.sqrt(np.mean((pm_o[-k:] - yt[-k:]) ** 2)))
    rmse_g = float(np.sqrt(np.mean((pm_g[-k:] - yt[-k:]) ** 2)))
    assert np.isfinite(rmse_g) and abs(rmse_g - rmse_o) <= 1e-4 * rmse_o, (rmse_g, rmse_o)
