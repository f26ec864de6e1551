"""MLP back-constraint of the hidden layers: the recognition recursion of autoreg.layers.Layer
with ``back_cstr=True`` (/root/reference/autoreg/layers.py:398-441 _init_encoder, :623-666 _encoder_freerun, :668-756
_encoder_update_gradient) and its network (/root/reference/autoreg/mlp.py:13-175, Theano there).

For every hidden layer the latent MEANS of the predicted rows are no free parameters: row X_win + n is the output of a
tanh MLP applied to the layer's own previous X_win means and the U_win-row window of the signal above (controls for the top
hidden layer, the latent means of the layer above otherwise),

    mean[X_win + n] = MLP([mean[n : n + X_win].flat, up_mean[U_off + n : U_off + n + U_win].flat]),  U_off = up_len - N - U_win + 1

so the free parameters are the first X_win rows (init_Xs: mean and variance), the variances of the other rows (Xs_var) and the
network weights.  The reference back-propagates through the recursion by hand (reverse loop, layers.py:705-712).  Here the
recursion and its reverse loop are persistent per-sequence scans in librgp_hip.so (rgp_mlp_scan_forward / _backward,
rgp_amd/csrc/rgp_mlp.hip) wrapped as one torch.autograd.Function per hidden layer; torch keeps the parameters, the softplus
of the variances and the weight-gradient GEMMs over the stored activations.  ``forward_loop`` is the same recursion as a plain
PyTorch-ROCm fp64 graph (one network call per time step): the reference of the scan in the tests, and the path for networks
that exceed the scan's limits (more than 4 linear layers, a layer wider than 256, or weights beyond 160 KB of LDS)."""
import math

import torch
import torch.nn as nn


class _MlpScanFn(torch.autograd.Function):
    """mean = scan(init_mean, up, W0, b0, W1, b1, ...) through librgp_hip.so."""

    @staticmethod
    def forward(ctx, handle, spec, dims, init_mean, up, *wb):
        from . import capi
        Ws, bs = [w.contiguous() for w in wb[0::2]], [b.contiguous() for b in wb[1::2]]
        init_c = init_mean.contiguous()
        up_c = None if up is None else up.contiguous()
        mean, acts = capi.mlp_scan_forward(handle, spec, dims, init_c, up_c, Ws, bs)
        ctx.handle, ctx.spec, ctx.dims, ctx.has_up = handle, spec, dims, up is not None
        ctx.up_needs_grad = up is not None and up.requires_grad
        ctx.save_for_backward(mean, *(() if up_c is None else (up_c,)), *acts, *Ws, *bs)
        return mean

    @staticmethod
    def backward(ctx, g_mean):
        from . import capi
        sp, dims = ctx.spec, ctx.dims
        nl = len(dims) - 1
        saved = list(ctx.saved_tensors)
        mean = saved.pop(0)
        up = saved.pop(0) if ctx.has_up else None
        acts, Ws, bs = saved[:nl - 1], saved[nl - 1:2 * nl - 1], saved[2 * nl - 1:]
        g_init, g_up, deltas = capi.mlp_scan_backward(ctx.handle, sp, dims, up, Ws, bs, acts, g_mean.contiguous(),
                                                      ctx.up_needs_grad)
        # weight gradients: sum over sequences and steps of delta_k in_k^T (in_0 = the window rows, in_k = activations)
        B, Xw, Xd, Uw, Ud, N = mean.shape[0], sp['X_win'], sp['X_dim'], sp['U_win'], sp['U_dim'], sp['N']
        x_in = mean[:, :Xw + N - 1].unfold(1, Xw, 1).permute(0, 1, 3, 2).reshape(B, N, Xw * Xd)
        if Uw > 0:
            u_off = up.shape[1] - N - Uw + 1
            u_in = up[:, u_off:u_off + N + Uw - 1].unfold(1, Uw, 1).permute(0, 1, 3, 2).reshape(B, N, Uw * Ud)
            x_in = torch.cat((x_in, u_in), dim=2)
        ins = [x_in] + list(acts)
        grads = []
        # (out x B N) (B N x in) products with a tiny result: one GEMM would put the whole 130 000-long sum on a single
        # workgroup (14 ms at config-4 sizes), so the sum is split into ~256-row chunks (batched GEMM) and the partial
        # products are added afterwards
        rows = B * N
        chunk = 256
        nch = rows // chunk
        body = nch * chunk
        for k in range(nl):
            dk = deltas[k].reshape(rows, dims[k + 1])
            ik = ins[k].reshape(rows, dims[k])
            g = dk[body:].t().mm(ik[body:])
            if nch > 0:
                g = g + torch.bmm(dk[:body].reshape(nch, chunk, -1).transpose(1, 2), ik[:body].reshape(nch, chunk, -1)).sum(0)
            grads.append(g)
            grads.append(dk.sum(0))
        return (None, None, None, g_init, g_up) + tuple(grads)


def _mlp(n_units):
    """mlp.MLP(nUnits): tanh hidden layers, linear output; W ~ U(-1,1) sqrt(6/(fan_in+fan_out)), b = 0 (mlp.py:27-31,132)."""
    mods = []
    for k in range(len(n_units) - 1):
        lin = nn.Linear(n_units[k], n_units[k + 1])
        with torch.no_grad():
            lin.weight.uniform_(-1.0, 1.0).mul_(math.sqrt(6.0 / (n_units[k] + n_units[k + 1])))
            lin.bias.zero_()
        mods.append(lin)
        if k < len(n_units) - 2:
            mods.append(nn.Tanh())
    return nn.Sequential(*mods)


def _inv_softplus(x):
    return x + torch.log(-torch.expm1(-x))


class MlpBackConstraint(nn.Module):
    """Recursion networks and free latent parameters of all hidden layers (equal-length sequences, batched over them).

    layers: list of dicts, top hidden layer FIRST, with keys X_win, X_dim, U_win, U_dim, N (predicted rows), up_len."""

    def __init__(self, layers, n_seq, X_variance=0.01, MLP_dims=None):
        nn.Module.__init__(self)
        self.specs = layers
        self.nets = nn.ModuleList()
        self.init_mean = nn.ParameterList()
        self.init_var_raw = nn.ParameterList()
        self.var_raw = nn.ParameterList()
        for sp in layers:
            Q = sp['X_win'] * sp['X_dim'] + sp['U_win'] * sp['U_dim']
            assert sp['X_win'] > 0, 'Neural Network constraints only applies autoregressive structure!'      # layers.py:403
            dims = [Q, 2 * Q, Q + sp['X_dim'] // 2, sp['X_dim']] if MLP_dims is None else [Q] + list(MLP_dims) + [sp['X_dim']]
            self.nets.append(_mlp(dims))
            self.init_mean.append(nn.Parameter(torch.randn(n_seq, sp['X_win'], sp['X_dim']) * 1e-2))            # layers.py:437
            v = torch.full((n_seq, sp['X_win'] + sp['N'], sp['X_dim']), float(X_variance))
            self.init_var_raw.append(nn.Parameter(_inv_softplus(v[:, :sp['X_win']])))
            self.var_raw.append(nn.Parameter(_inv_softplus(v[:, sp['X_win']:])))
        self.double()
        self.softplus = nn.Softplus()

    handle = None           # rgp_amd.capi.Handle: set by the model; without it only forward_loop is available

    def _scan_ok(self, k):
        lins = [mod for mod in self.nets[k] if hasattr(mod, 'weight')]
        dims = [lins[0].in_features] + [l.out_features for l in lins]
        lds = sum(l.out_features * (l.in_features | 1) + l.out_features for l in lins) + 2 * max(dims)
        sp = self.specs[k]
        lds += (sp['X_win'] + 1) * sp['X_dim'] + max(1, sp['U_win'] * sp['U_dim'])
        return len(lins) <= 4 and max(dims) <= 256 and lds * 8 <= 160 * 1024, lins, dims

    def forward(self, controls, idx=slice(None)):
        """controls: (n_seq, u_len, D_u) tensor or None.  Returns (means, variances), lists top hidden layer first, each
        (batch, X_win + N, X_dim).  Device scan per hidden layer (module docstring)."""
        if self.handle is None:
            raise RuntimeError('MlpBackConstraint.forward needs the librgp_hip handle (set .handle); forward_loop is the torch graph')
        means, variances = [], []
        up = None if controls is None else controls[idx]
        for k, sp in enumerate(self.specs):
            ok, lins, dims = self._scan_ok(k)
            if not ok:
                m = self._loop_layer(k, up, idx)
            else:
                wb = []
                for l in lins:
                    wb += [l.weight, l.bias]
                m = _MlpScanFn.apply(self.handle, sp, dims, self.init_mean[k][idx], up if sp['U_win'] > 0 else None, *wb)
            v = torch.cat((self.softplus(self.init_var_raw[k][idx]), self.softplus(self.var_raw[k][idx])), dim=1)
            means.append(m)
            variances.append(v)
            up = m
        return means, variances

    def _loop_layer(self, k, up, idx):
        sp = self.specs[k]
        Xw, N, Uw = sp['X_win'], sp['N'], sp['U_win']
        rows = list(self.init_mean[k][idx].unbind(dim=1))                   # X_win tensors (batch, X_dim)
        net = self.nets[k]
        if Uw > 0:
            u_off = up.shape[1] - N - Uw + 1
        for n in range(N):
            parts = rows[n:n + Xw]
            if Uw > 0:
                parts = parts + [up[:, u_off + n:u_off + n + Uw].reshape(up.shape[0], -1)]
            rows.append(net(torch.cat(parts, dim=1)))
        return torch.stack(rows, dim=1)

    def forward_loop(self, controls, idx=slice(None)):
        """The same recursion as a PyTorch graph, one network call per time step (reference of the scan)."""
        means, variances = [], []
        up = None if controls is None else controls[idx]
        for k, sp in enumerate(self.specs):
            m = self._loop_layer(k, up, idx)
            v = torch.cat((self.softplus(self.init_var_raw[k][idx]), self.softplus(self.var_raw[k][idx])), dim=1)
            means.append(m)
            variances.append(v)
            up = m                                                               # the layer below takes these means as its input
        return means, variances
