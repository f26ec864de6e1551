"""MLP back-constraint (autoreg.DeepAutoreg with back_cstr=True, layers.py:398-441,623-756): the recursion reproduces a
plain NumPy restatement of Layer._encoder_freerun, the GP part on the device matches the oracle on the recursion-produced
q(X), gradients of the recursion parameters (network weights, initial rows, variances) pass a finite-difference check, and
SVI steps improve the bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_model import make_data

WINS, NDIMS, UWIN, M = [0, 3, 2], [3, 2, 3], 2, 4


def build(n_seq=3, T=12, n_total=None, seed=1):
    from rgp_amd.model import DeepAutoreg_mlp
    U, Y = make_data(n_seq, T, seed=4)
    m = DeepAutoreg_mlp(WINS, Y, U=U, U_win=UWIN, nDims=NDIMS, num_inducing=M, X_variance=0.05, inference_method='svi',
                        mb_inf_tot_data_size=n_total, seed=seed)
    rng = np.random.default_rng(2)
    for L in m.layers:
        m.set_param(L, 'inv_l', 0.3 + rng.random(L.Q))
        m.set_param(L, 'noise', 0.2 + 0.1 * rng.random())
        m.set_param(L, 'qU_W', rng.standard_normal((L.M, L.M)) * 0.2)
        m.set_param(L, 'qU_a', 0.3)
    return m, U, Y


def numpy_freerun(m, U):
    """layers.py:623-666 restated with NumPy for every hidden layer, top first; returns {layer: [per-sequence means]}."""
    out = {}
    up = [m.controls[s].cpu().numpy() for s in range(m.nSeq)]       # controls after the model's data shift (model.py:657-673)
    for k, sp in enumerate(m.encoder.specs):
        i, Xw, Xd, Uw = sp['index'], sp['X_win'], sp['X_dim'], sp['U_win']
        lins = [mod for mod in m.encoder.nets[k] if hasattr(mod, 'weight')]
        Ws = [(l.weight.detach().cpu().numpy(), l.bias.detach().cpu().numpy()) for l in lins]
        init = m.encoder.init_mean[k].detach().cpu().numpy()
        seqs = []
        for s in range(m.nSeq):
            N = sp['N']
            X = np.zeros((Xw + N, Xd))
            X[:Xw] = init[s]
            u_off = up[s].shape[0] - N - Uw + 1
            for n in range(N):
                x = np.concatenate([X[n:n + Xw].ravel(), up[s][u_off + n:u_off + n + Uw].ravel()])
                for j, (W, b) in enumerate(Ws):
                    x = W @ x + b
                    if j < len(Ws) - 1:
                        x = np.tanh(x)
                X[Xw + n] = x
            seqs.append(X)
        out[i] = seqs
        up = seqs
    return out


def test_recursion_matches_numpy_restatement_and_oracle_gp_part():
    from oracle import model as om
    m, U, Y = build()
    e = float(m.evaluate().item())
    m.handle.sync()
    ref = numpy_freerun(m, U)
    for i in (1, 2):
        for s in range(m.nSeq):
            np.testing.assert_allclose(m.latent(i, s).cpu().numpy(), ref[i][s], rtol=1e-12, atol=1e-13)
    o = om.DeepAutoreg(WINS, Y, U=U, U_win=UWIN, nDims=NDIMS, num_inducing=M, init='Y', inference_method='svi',
                       rng=np.random.default_rng(0), fast_psi=False)
    for L in m.layers:
        p = o.P[L.index]
        for nm in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            getattr(p, nm)[...] = m.view(L, nm).cpu().numpy()
        p.variance = float(m.view(L, 'variance').item()); p.noise = float(m.view(L, 'noise').item())
        p.qU_a = float(m.view(L, 'qU_a').item())
    for i in (1, 2):
        for s in range(m.nSeq):
            o.Xmean[i][s][...] = m.latent(i, s).cpu().numpy()
            o.Xvar[i][s][...] = m.latent(i, s, var=True).cpu().numpy()
    eo, Go = o.elbo_and_grads()
    assert abs(e - eo) <= 1e-11 * abs(eo), (e, eo)
    for i in (1, 2):
        for s in range(m.nSeq):
            np.testing.assert_allclose(m.latent(i, s, grad=True).cpu().numpy(), Go[(i, 'Xmean', s)], rtol=1e-8, atol=1e-10)


def test_recursion_parameter_gradients_fd():
    import torch
    m, U, Y = build()
    m.evaluate()
    m.handle.sync()
    g = m.encoder_grad_vector().cpu().numpy().copy()
    rng = np.random.default_rng(7)
    eps = 1e-6
    off = 0
    for p in m.encoder.parameters():
        n = p.numel()
        for _ in range(2):
            j = int(rng.integers(n))
            old = float(p.view(-1)[j].item())
            def setv(val):
                with torch.no_grad():
                    p.view(-1)[j] = val
            setv(old + eps); ep = float(m.evaluate().item())
            setv(old - eps); em = float(m.evaluate().item())
            setv(old)
            num = (ep - em) / (2 * eps)
            assert abs(num - g[off + j]) <= 2e-5 * max(1.0, abs(num)), (off + j, num, g[off + j])
        off += n


def test_minibatch_additivity_and_svi_progress():
    m, U, Y = build(n_seq=4, n_total=4)
    e_all = float(m.evaluate().item()); g_all = m.grad_array[:m.n_global].cpu().numpy().copy()
    gw_all = np.concatenate([p.grad.reshape(-1).cpu().numpy() for p in m.encoder.nets.parameters()])
    e1 = float(m.evaluate(slice(0, 2)).item()); g1 = m.grad_array[:m.n_global].cpu().numpy().copy()
    gw1 = np.concatenate([p.grad.reshape(-1).cpu().numpy() for p in m.encoder.nets.parameters()])
    e2 = float(m.evaluate([2, 3]).item()); g2 = m.grad_array[:m.n_global].cpu().numpy().copy()
    gw2 = np.concatenate([p.grad.reshape(-1).cpu().numpy() for p in m.encoder.nets.parameters()])
    m.handle.sync()
    assert abs(e1 + e2 - e_all) <= 1e-12 * abs(e_all)
    np.testing.assert_allclose(g1 + g2, g_all, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(gw1 + gw2, gw_all, rtol=1e-8, atol=1e-10)
    e0 = e_all
    for k in range(30):
        m.svi_step(slice(0, 4), lr=5e-3)
    e3 = float(m.evaluate().item())
    m.handle.sync()
    assert np.isfinite(e3) and e3 > e0, (e0, e3)


@pytest.mark.parametrize('dims_mid,Xw,Xd,Uw,Ud,N', [((7, 5), 3, 2, 2, 3, 11), ((16,), 1, 1, 0, 0, 9), ((9, 6, 4), 2, 3, 4, 1, 40),
                                                     ((80, 41), 20, 1, 20, 1, 300)])
def test_scan_kernels_match_torch_graph(dims_mid, Xw, Xd, Uw, Ud, N):
    """rgp_mlp_scan_forward/_backward against the same recursion written as a PyTorch graph (one network call per step) and
    torch.autograd: means, gradients of the first rows, of the signal above and of every weight and bias."""
    import torch
    from rgp_amd.capi import Handle
    from rgp_amd.mlp_encoder import MlpBackConstraint
    H = Handle()
    torch.manual_seed(3)
    B = 3
    up_len = N + Uw - 1 + 2 if Uw > 0 else 0                   # two leading rows outside every window
    sp = dict(index=1, X_win=Xw, X_dim=Xd, U_win=Uw, U_dim=Ud, N=N, up_len=up_len)
    enc = MlpBackConstraint([sp], B, X_variance=0.05, MLP_dims=list(dims_mid)).to(H.device)
    enc.handle = H
    with torch.no_grad():
        enc.init_mean[0].normal_()
        for p in enc.nets.parameters():
            p.add_(0.1 * torch.randn_like(p))
    controls = None
    if Uw > 0:
        controls = torch.randn(B, up_len, Ud, dtype=torch.float64, device=H.device, requires_grad=True)
    w = torch.randn(B, Xw + N, Xd, dtype=torch.float64, device=H.device)
    outs = []
    for fn in (enc.forward, enc.forward_loop):
        enc.zero_grad()
        if controls is not None:
            controls.grad = None
        means, _ = fn(controls)
        (means[0] * w).sum().backward()
        H.sync()
        outs.append((means[0].detach().clone(), [p.grad.clone() for p in enc.nets.parameters()], enc.init_mean[0].grad.clone(),
                     None if controls is None else controls.grad.clone()))
    (m1, gw1, gi1, gu1), (m2, gw2, gi2, gu2) = outs
    tol = dict(rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), **tol)
    np.testing.assert_allclose(gi1.cpu().numpy(), gi2.cpu().numpy(), **tol)
    for a, b in zip(gw1, gw2):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-10, atol=1e-11)
    if gu1 is not None:
        np.testing.assert_allclose(gu1.cpu().numpy(), gu2.cpu().numpy(), **tol)
        assert float(gu1[:, :2].abs().max()) == 0.0            # rows above every window get no gradient


# ----------------------------------------------------------------------------------------------------------------------
# minibatch parameterisations of the first rows / the predicted rows' variances (layers.py:408-439, model.py:113-124,165-171)
# ----------------------------------------------------------------------------------------------------------------------
MODES = [('one', 'one'), ('one', 'all'), ('all', 'one'), ('all', 'all'), ('mlp', 'one'), ('mlp', 'all')]


def build_mb(means, vars_, n_seq=4, T=12, seed=1):
    import torch
    from rgp_amd.model import DeepAutoreg_mlp
    U, Y = make_data(n_seq, T, seed=4)
    m = DeepAutoreg_mlp(WINS, Y, U=U, U_win=UWIN, nDims=NDIMS, num_inducing=M, X_variance=0.05, inference_method='svi',
                        mb_inf_tot_data_size=n_seq, seed=seed, minibatch_inference=True, mb_inf_init_xs_means=means,
                        mb_inf_init_xs_vars=vars_, mb_inf_sample_idxes=list(range(n_seq)), init_MLP_dims=[7])
    rng = np.random.default_rng(2)
    for L in m.layers:
        m.set_param(L, 'inv_l', 0.3 + rng.random(L.Q))
        m.set_param(L, 'noise', 0.2 + 0.1 * rng.random())
        m.set_param(L, 'qU_W', rng.standard_normal((L.M, L.M)) * 0.2)
        m.set_param(L, 'qU_a', 0.3)
    torch.manual_seed(5)
    with torch.no_grad():                                        # distinct values everywhere, so a wrong broadcast shows
        for p in list(m.encoder.init_mean) + list(m.encoder.init_var_raw) + list(m.encoder.var_raw):
            p.add_(0.2 * torch.randn_like(p))
    return m, U, Y


@pytest.mark.parametrize('means,vars_', MODES)
def test_minibatch_modes_layout_and_fd(means, vars_):
    """Shapes of the free parameters per mode, the broadcast they imply on q(X), and a finite-difference check of every
    parameter group's gradient (shared blocks receive the sum over the minibatch, layers.py:725-750)."""
    import torch
    m, U, Y = build_mb(means, vars_)
    enc = m.encoder
    n = m.nSeq
    for k, sp in enumerate(enc.specs):
        assert tuple(enc.var_raw[k].shape) == ((1 if vars_ == 'one' else n), 1, sp['X_dim'])
        if means == 'mlp':
            assert len(enc.init_mean) == 0 and len(enc.init_nets) == len(enc.specs)
        else:
            assert tuple(enc.init_mean[k].shape) == ((1 if means == 'one' else n), sp['X_win'], sp['X_dim'])
    seqs = [0, 2, 3]
    m.evaluate(seqs)
    m.handle.sync()
    for sp in enc.specs:
        i, Xw = sp['index'], sp['X_win']
        V = [m.latent(i, s, var=True).cpu().numpy() for s in seqs]
        Mn = [m.latent(i, s).cpu().numpy() for s in seqs]
        for v in V:
            assert np.all(v[Xw:] == v[Xw]), 'one variance vector per sequence, the same for every predicted row'
        same_v = all(np.array_equal(V[0][Xw:], v[Xw:]) for v in V[1:])
        assert same_v == (vars_ == 'one')
        same_m = all(np.array_equal(Mn[0][:Xw], x[:Xw]) for x in Mn[1:])
        assert same_m == (means == 'one')
        if means == 'mlp':
            assert all(np.all(v[:Xw] == 0.05) for v in V)        # first rows' variances stay at X_variance (layers.py:473)
    g = m.encoder_grad_vector().cpu().numpy().copy()
    rng = np.random.default_rng(7)
    eps = 1e-6
    off = 0
    for p in enc.parameters():
        cnt = p.numel()
        for _ in range(2):
            j = int(rng.integers(cnt))
            old = float(p.view(-1)[j].item())

            def setv(val):
                with torch.no_grad():
                    p.view(-1)[j] = val
            setv(old + eps); ep = float(m.evaluate(seqs).item())
            setv(old - eps); em = float(m.evaluate(seqs).item())
            setv(old)
            num = (ep - em) / (2 * eps)
            assert abs(num - g[off + j]) <= 2e-5 * max(1.0, abs(num)), (off + j, num, g[off + j])
        off += cnt


@pytest.mark.parametrize('means,vars_', MODES)
def test_minibatch_modes_additivity(means, vars_):
    """ELBO and every gradient (GP globals, networks, shared and per-sequence blocks) of the whole set equal the sums over
    two disjoint minibatches (testing/minibatch_tests.py:281-296)."""
    m, U, Y = build_mb(means, vars_)

    def ev(seqs):
        e = float(m.evaluate(seqs).item())
        m.handle.sync()
        return e, m.grad_array[:m.n_global].cpu().numpy().copy(), m.encoder_grad_vector().cpu().numpy().copy()
    e_all, g_all, w_all = ev(None)
    e1, g1, w1 = ev([0, 3])
    e2, g2, w2 = ev([1, 2])
    assert abs(e1 + e2 - e_all) <= 1e-12 * abs(e_all)
    np.testing.assert_allclose(g1 + g2, g_all, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(w1 + w2, w_all, rtol=1e-8, atol=1e-10)


def test_streamer_run_touches_every_sequences_state(tmp_path):
    """A streamer-driven SVI run in the per-sequence mode: after one pass every sequence's stored first rows and variance
    vector have moved, a sequence outside the current minibatch keeps its values and its Adam state during a step, and a
    checkpoint restores GP and recursion parameters alike."""
    import torch
    from rgp_amd.data_streamers import StdMemoryDataStreamer
    from rgp_amd.model import DeepAutoreg_mlp
    m, U, Y = build_mb('all', 'all', n_seq=6)
    ds = StdMemoryDataStreamer(Y, U, 2)
    m.set_DataStreamer(ds)
    start = [p.detach().clone() for p in m.encoder.per_sequence_parameters()]
    seen = set()
    for k in range(3):
        mb = m._next_minibatch()
        before = [p.detach().clone() for p in m.encoder.per_sequence_parameters()]
        m.svi_step(mb, lr=1e-2)
        rest = [s for s in range(m.nSeq) if s not in mb]
        for p, b in zip(m.encoder.per_sequence_parameters(), before):
            assert torch.equal(p[rest], b[rest]) and not torch.equal(p[mb], b[mb])
        seen.update(mb)
    assert seen == set(range(m.nSeq))
    for p, b in zip(m.encoder.per_sequence_parameters(), start):
        assert all(not torch.equal(p[s], b[s]) for s in range(m.nSeq))
    # checkpoint: GP globals and every recursion tensor
    e0 = float(m.evaluate().item()); m.handle.sync()
    m.save(str(tmp_path / 'ck'))
    m2, _, _ = build_mb('all', 'all', n_seq=6, seed=9)
    m2.load(str(tmp_path / 'ck'))
    e1 = float(m2.evaluate().item()); m2.handle.sync()
    assert e1 == e0
    # L-BFGS over [GP globals | recursion parameters] improves the bound
    x0 = m2.get_optimizer_array()
    f0, g0 = m2.objective_and_grad(x0)
    assert g0.shape == x0.shape and abs(f0 + e0) <= 1e-12 * abs(e0)
    f1, _ = m2.optimize(max_iters=5)
    assert f1 < f0
