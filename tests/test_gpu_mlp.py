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
