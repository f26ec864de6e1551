"""Sequence-encoder back-constraint (autoreg.DeepAutoreg_rnn): the GP part on the device matches the oracle on the
encoder-produced q(X) (latents with one row per raw time step), encoder-weight gradients pass a finite-difference check
(the reference's gate, testing/rnn_rgp_test.py:112), and everything is additive over minibatches (rnn_rgp_test.py:160-168)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_model import make_data


def build(rnn_type='rnn', n_seq=3, T=12, bidirectional=False, n_total=None):
    from rgp_amd.model import DeepAutoreg_rnn
    U, Y = make_data(n_seq, T, seed=4)
    m = DeepAutoreg_rnn([0, 3, 3], Y, U=U, U_win=2, nDims=[3, 2, 3], num_inducing=4, rnn_type=rnn_type,
                        rnn_hidden_dims=[5], rnn_bidirectional=bidirectional, inference_method='svi',
                        mb_inf_tot_data_size=n_total, seed=1)
    rng = np.random.default_rng(2)
    for L in m.layers:
        m.set_param(L, 'inv_l', 0.3 + rng.random(L.Q))
        m.set_param(L, 'noise', 0.2 + 0.1 * rng.random())
        m.set_param(L, 'qU_W', rng.standard_normal((L.M, L.M)) * 0.2)
        m.set_param(L, 'qU_a', 0.3)
    return m, U, Y


@pytest.mark.parametrize('rnn_type', ['rnn', 'lstm', 'gru'])
def test_gp_part_matches_oracle_on_encoder_latents(rnn_type):
    from oracle import model as om
    m, U, Y = build(rnn_type)
    e = float(m.evaluate().item())
    m.handle.sync()
    o = om.DeepAutoreg([0, 3, 3], Y, U=U, U_win=2, nDims=[3, 2, 3], num_inducing=4, init='Y', inference_method='svi',
                       rng=np.random.default_rng(0), fast_psi=False, latent_full=True)
    for L in m.layers:
        p = o.P[L.index]
        for nm in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            getattr(p, nm)[...] = m.view(L, nm).cpu().numpy()
        p.variance = float(m.view(L, 'variance').item()); p.noise = float(m.view(L, 'noise').item())
        p.qU_a = float(m.view(L, 'qU_a').item())
    for i in (1, 2):
        for s in range(3):
            assert o.Xmean[i][s].shape == (12, m.nDims[i])
            o.Xmean[i][s][...] = m.latent(i, s).cpu().numpy()
            o.Xvar[i][s][...] = m.latent(i, s, var=True).cpu().numpy()
    eo, Go = o.elbo_and_grads()
    assert abs(e - eo) <= 1e-11 * abs(eo), (e, eo)
    for i in (1, 2):
        for s in range(3):
            np.testing.assert_allclose(m.latent(i, s, grad=True).cpu().numpy(), Go[(i, 'Xmean', s)], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(m.latent(i, s, var=True, grad=True).cpu().numpy(), Go[(i, 'Xvar', s)], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(m.view(m.layers[0], 'Z', grad=True).cpu().numpy(), Go[(m.layers[0].index, 'Z')], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('rnn_type,bi', [('rnn', False), ('gru', True)])
def test_encoder_weight_gradients_fd(rnn_type, bi):
    import torch
    m, U, Y = build(rnn_type, bidirectional=bi)
    m.evaluate()
    m.handle.sync()
    g = m.encoder_grad_vector().cpu().numpy().copy()
    params = list(m.encoder.parameters())
    rng = np.random.default_rng(7)
    eps = 1e-6
    off = 0
    for p in params:
        n = p.numel()
        for _ in range(2):
            j = int(rng.integers(n))
            def setv(val):
                with torch.no_grad():
                    p.view(-1)[j] = val
            old = float(p.view(-1)[j].item())
            setv(old + eps); ep = float(m.evaluate().item())
            setv(old - eps); em = float(m.evaluate().item())
            setv(old)
            num = (ep - em) / (2 * eps)
            assert abs(num - g[off + j]) <= 2e-5 * max(1.0, abs(num)), (off + j, num, g[off + j])
        off += n


def test_minibatch_additivity_including_encoder():
    m, U, Y = build('lstm', n_seq=4, n_total=4)
    e_all = float(m.evaluate().item()); g_all = m.grad_array[:m.n_global].cpu().numpy().copy()
    ge_all = m.encoder_grad_vector().cpu().numpy().copy()
    e1 = float(m.evaluate(slice(0, 2)).item()); g1 = m.grad_array[:m.n_global].cpu().numpy().copy()
    ge1 = m.encoder_grad_vector().cpu().numpy().copy()
    e2 = float(m.evaluate([2, 3]).item()); g2 = m.grad_array[:m.n_global].cpu().numpy().copy()
    ge2 = m.encoder_grad_vector().cpu().numpy().copy()
    m.handle.sync()
    assert abs(e1 + e2 - e_all) <= 1e-12 * abs(e_all)
    np.testing.assert_allclose(g1 + g2, g_all, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(ge1 + ge2, ge_all, rtol=1e-8, atol=1e-10)


def test_svi_steps_improve_the_bound():
    m, U, Y = build('gru', n_seq=4, n_total=4)
    e0 = float(m.evaluate().item())
    for k in range(30):
        m.svi_step(slice(0, 4), lr=5e-3)
    e1 = float(m.evaluate().item())
    m.handle.sync()
    assert np.isfinite(e1) and e1 > e0, (e0, e1)


@pytest.mark.parametrize('cell,T,B,Din,H,bi', [('gru', 9, 3, 5, 4, False), ('gru', 40, 2, 7, 20, True), ('gru', 25, 1, 3, 33, False),
                                                ('gru', 17, 4, 2, 80, False), ('rnn', 30, 3, 5, 12, True), ('rnn', 11, 2, 3, 140, False),
                                                ('lstm', 30, 3, 5, 12, True), ('lstm', 13, 2, 4, 64, False),
                                                ('gru', 300, 4, 6, 10, False)])          # >= 1024 rows: split-K head gradients
def test_rnn_scan_matches_torch_module(cell, T, B, Din, H, bi):
    """rgp_rnn_scan_forward/_backward (device scan of the recurrence) against torch's own nn.GRU / nn.RNN / nn.LSTM on the same
    module: outputs, and the gradients of every weight, bias and of the input."""
    import torch
    from rgp_amd.capi import Handle
    from rgp_amd.rnn_encoder import MeanVarRNN
    Hd = Handle()
    torch.manual_seed(5)
    lay = MeanVarRNN(Din, 3, H, cell, bi).double().to(Hd.device)
    lay.handle = Hd
    assert lay._scan_cell() is not None
    x = torch.randn(T, B, Din, dtype=torch.float64, device=Hd.device, requires_grad=True)
    wm = torch.randn(T, B, 3, dtype=torch.float64, device=Hd.device)
    wv = torch.randn(T, B, 3, dtype=torch.float64, device=Hd.device)
    res = []
    for scan in (True, False):
        lay.use_scan = scan
        lay.zero_grad()
        x.grad = None
        mean, var = lay(x)
        ((mean * wm).sum() + (var * wv).sum()).backward()
        Hd.sync()
        res.append((mean.detach().clone(), var.detach().clone(), x.grad.clone(), [p.grad.clone() for p in lay.parameters()]))
    (m1, v1, gx1, gp1), (m2, v2, gx2, gp2) = res
    tol = dict(rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), **tol)
    np.testing.assert_allclose(v1.cpu().numpy(), v2.cpu().numpy(), **tol)
    np.testing.assert_allclose(gx1.cpu().numpy(), gx2.cpu().numpy(), **tol)
    for a, b in zip(gp1, gp2):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-9, atol=1e-11)


def test_chain_carry_over_equals_plain_steps():
    """The encoder models start the next step's M x M chains right after the parameter update (DeepAutoreg._restart_chains);
    the parameters after several steps, with evaluate() and a set_param() in between, equal those of the plain sequence."""
    a, U, Y = build('gru', n_seq=4, n_total=4)
    b, _, _ = build('gru', n_seq=4, n_total=4)
    b.pipeline = False
    for k in range(4):
        ea = float(a.svi_step(slice(0, 4), lr=1e-2).item())
        eb = float(b.svi_step(slice(0, 4), lr=1e-2).item())
        assert abs(ea - eb) <= 1e-10 * abs(eb), (k, ea, eb)
    ea = float(a.evaluate().item()); eb = float(b.evaluate().item())
    assert abs(ea - eb) <= 1e-10 * abs(eb)
    L = a.layers[0]
    a.set_param(L, 'noise', 0.21); b.set_param(L, 'noise', 0.21)
    for k in range(2):
        ea = float(a.svi_step(slice(0, 2), lr=1e-2).item())
        eb = float(b.svi_step(slice(0, 2), lr=1e-2).item())
        assert abs(ea - eb) <= 1e-10 * abs(eb), (k, ea, eb)
    a.handle.sync(); b.handle.sync()
    assert float((a.param_array[:a.n_global] - b.param_array[:b.n_global]).abs().max()) <= 1e-10 * float(b.param_array[:b.n_global].abs().max())


def test_config3_geometry_matches_oracle():
    """BASELINE config 3 at its geometry (walk_run mocap: D_y = 59, D_u = 1, wins = [0, 20, 20], U_win = 20, M = 100, GRU
    encoder; examples/walk_run_2_alex.py:364-393): ELBO and every gradient group of the three GP layers against the oracle on
    the encoder-produced q(X) (ELBO 1e-9, gradients 1e-8), plus central differences on encoder weights."""
    import torch
    from oracle import model as om
    from rgp_amd.model import DeepAutoreg_rnn
    n_seq, T, wins, nDims, M, Uw = 4, 100, [0, 20, 20], [59, 1, 1], 100, 20
    U, Y = make_data(n_seq, T, u_dim=1, y_dim=59, seed=6)
    m = DeepAutoreg_rnn(wins, Y, U=U, U_win=Uw, nDims=nDims, num_inducing=M, rnn_type='gru', rnn_hidden_dims=[40],
                        inference_method='svi', mb_inf_tot_data_size=n_seq, seed=1)
    rng = np.random.default_rng(2)
    for L in m.layers:
        m.set_param(L, 'inv_l', (0.3 + rng.random(L.Q)) * 4.0 / L.Q)
        m.set_param(L, 'noise', 0.2 + 0.1 * rng.random())
        m.set_param(L, 'qU_W', rng.standard_normal((L.M, L.M)) * 0.05)
        m.set_param(L, 'qU_a', 0.3)
    e = float(m.evaluate().item())
    m.handle.sync()
    o = om.DeepAutoreg(wins, Y, U=U, U_win=Uw, nDims=nDims, num_inducing=M, init='Y', inference_method='svi',
                       rng=np.random.default_rng(0), fast_psi=True, chunk=64, latent_full=True)
    for L in m.layers:
        p = o.P[L.index]
        for nm in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            getattr(p, nm)[...] = m.view(L, nm).cpu().numpy()
        p.variance = float(m.view(L, 'variance').item()); p.noise = float(m.view(L, 'noise').item())
        p.qU_a = float(m.view(L, 'qU_a').item())
    for i in (1, 2):
        for s in range(n_seq):
            o.Xmean[i][s][...] = m.latent(i, s).cpu().numpy()
            o.Xvar[i][s][...] = m.latent(i, s, var=True).cpu().numpy()
    eo, Go = o.elbo_and_grads()
    assert abs(e - eo) <= 1e-9 * abs(eo), (e, eo)
    Gg = m.grads_as_oracle_dict()
    for k, v in Go.items():
        a, b = np.asarray(Gg[k], dtype=float), np.asarray(v, dtype=float)
        scale = max(np.abs(b).max(), 1e-12)
        assert np.abs(a.reshape(b.shape) - b).max() <= 1e-8 * scale, (k, np.abs(a.reshape(b.shape) - b).max(), scale)
    # encoder weights: analytic gradient (device GP gradients pushed through the encoder) vs central differences of the ELBO
    g = m.encoder_grad_vector().cpu().numpy().copy()
    off = 0
    eps = 1e-6
    for p_ in m.encoder.parameters():
        n = p_.numel()
        j = int(rng.integers(n))
        old = float(p_.view(-1)[j].item())
        with torch.no_grad():
            p_.view(-1)[j] = old + eps
        ep = float(m.evaluate().item())
        with torch.no_grad():
            p_.view(-1)[j] = old - eps
        em = float(m.evaluate().item())
        with torch.no_grad():
            p_.view(-1)[j] = old
        num = (ep - em) / (2 * eps)
        assert abs(num - g[off + j]) <= 1e-4 * max(1.0, abs(num)), (off + j, num, g[off + j])
        off += n


def test_checkpoint_and_lbfgs_cover_the_encoder_weights(tmp_path):
    """save / load round-trip the encoder's tensors with the GP vector (the reference's param_array holds both,
    examples/walk_run_2_alex.py:565-576), and the flat optimiser vector is [GP globals | encoder weights]."""
    import torch
    m, U, Y = build('gru')
    for k in range(3):
        m.svi_step(slice(0, 3), lr=1e-2)
    e0 = float(m.evaluate().item()); m.handle.sync()
    m.save(str(tmp_path / 'ck'))
    torch.manual_seed(11)
    m2, _, _ = build('gru')
    with torch.no_grad():
        for p in m2.encoder.parameters():
            p.add_(torch.randn_like(p))
    m2.load(str(tmp_path / 'ck'))
    e1 = float(m2.evaluate().item()); m2.handle.sync()
    assert e1 == e0
    x0 = m2.get_optimizer_array()
    n_enc = sum(p.numel() for p in m2.encoder.parameters())
    assert x0.size == m2.n_global + n_enc
    f0, g0 = m2.objective_and_grad(x0)
    assert abs(f0 + e0) <= 1e-12 * abs(e0) and np.abs(g0[m2.n_global:]).max() > 0
    f1, _ = m2.optimize(max_iters=5)
    assert f1 < f0
