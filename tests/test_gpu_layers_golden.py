"""The device model against fixtures produced by the reference's own autoreg/layers.py (tests/golden/layer_*.npz, made by
tests/golden/make_golden_layers.py): per-layer windows, bounds and every routed / scattered gradient, SVI and collapsed."""
import numpy as np
import pytest

from test_oracle_layers_golden import close, load_fixture, oracle_from_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['svi', 'vardtc'])
def test_device_model_matches_reference_layer_fixtures(name):
    from rgp_amd.model import DeepAutoreg
    g, U, Y, n_seq = load_fixture(name)
    o, svi = oracle_from_fixture(g, U, Y, n_seq)              # carrier of the fixture's parameters only
    m = DeepAutoreg([int(w) for w in g['wins']], Y, U=U, U_win=int(g['U_win']), nDims=[int(d) for d in g['nDims']],
                    num_inducing=int(g['M']), init='Y', inference_method='svi' if svi else None,
                    mb_inf_tot_data_size=int(g['n_total']), seed=0)
    m.load_from_oracle(o)
    elbo = float(m.evaluate().item())
    m.handle.sync()
    assert abs(elbo - float(g['elbo'])) <= 1e-11 * abs(float(g['elbo'])), (elbo, float(g['elbo']))
    G = m.grads_as_oracle_dict(None)
    for i in (0, 1, 2):
        close(G[(i, 'Z')], g['g_Z_%d' % i], rtol=1e-9)
        close(np.hstack([np.atleast_1d(G[(i, 'variance')]), np.asarray(G[(i, 'inv_l')]).reshape(-1)]), g['g_kern_%d' % i], rtol=1e-9)
        close(G[(i, 'noise')], g['g_noise_%d' % i], rtol=1e-9)
        if svi:
            close(G[(i, 'qU_m')], g['g_qU_m_%d' % i], rtol=1e-9)
            close(G[(i, 'qU_W')], g['g_qU_W_%d' % i], rtol=1e-9)
            close(G[(i, 'qU_a')], g['g_qU_a_%d' % i], rtol=1e-9)
    for i in (1, 2):
        for s in range(n_seq):
            close(G[(i, 'Xmean', s)], g['g_Xmean_%d_%d' % (i, s)], rtol=1e-9)
            close(G[(i, 'Xvar', s)], g['g_Xvar_%d_%d' % (i, s)], rtol=1e-9)
