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

    layers: list of dicts, top hidden layer FIRST, with keys X_win, X_dim, U_win, U_dim, N (predicted rows), up_len.

    Parameterisations of the per-sequence quantities (layers.py:408-439):
      minibatch_inference=False: every sequence owns its first X_win rows (mean and variance) and one variance per
        predicted row (layers.py:436-439);
      minibatch_inference=True: ``init_xs_means`` (the reference's mb_inf_init_xs_means) selects the first X_win rows --
        'one': a single (mean, variance) block shared by all sequences, gradients summed over the minibatch
        (layers.py:412,725-732); 'all': one block per sequence of the whole data set (layers.py:415-416; all n_seq blocks
        live on the device and a minibatch indexes them, which is what the reference's save / load on every minibatch
        switch amounts to, layers.py:342-362); 'mlp': the means are the output of one more network per layer applied to the
        first rows of the layer below -- the observations for the lowest hidden layer -- bottom-up
        (model.py:113-124,165-171, layers.py:443-472; dims [Q, Q^2, X_win X_dim]); the reference leaves this mode
        unfinished (layers.py:739 raises, :771 and model.py:115 stop in the debugger), so its gradient here is simply the
        chain rule through those networks, and the first rows' variances stay at X_variance as there (layers.py:473).
        ``init_xs_vars`` (mb_inf_init_xs_vars) selects the variances of the predicted rows -- 'one': a single X_dim vector
        for all sequences and rows (layers.py:429,744-746); 'all': one X_dim vector per sequence, the same for every row
        (layers.py:432-433,750)."""

    def __init__(self, layers, n_seq, X_variance=0.01, MLP_dims=None, minibatch_inference=False, init_xs_means='all',
                 init_xs_vars='all', low_init=None, init_MLP_dims=None):
        nn.Module.__init__(self)
        assert init_xs_means in ('one', 'all', 'mlp') and init_xs_vars in ('one', 'all')
        self.specs = layers
        self.n_seq = n_seq
        self.minibatch_inference = bool(minibatch_inference)
        self.means_mode = init_xs_means if minibatch_inference else 'all'
        self.vars_mode = (init_xs_vars if minibatch_inference else 'rows')
        self.nets = nn.ModuleList()
        self.init_nets = nn.ModuleList()
        self.init_mean = nn.ParameterList()
        self.init_var_raw = nn.ParameterList()
        self.var_raw = nn.ParameterList()
        self.X_variance = float(X_variance)
        for sp in layers:
            Q = sp['X_win'] * sp['X_dim'] + sp['U_win'] * sp['U_dim']
            assert sp['X_win'] > 0, 'Neural Network constraints only applies autoregressive structure!'      # layers.py:403
            dims = [Q, 2 * Q, Q + sp['X_dim'] // 2, sp['X_dim']] if MLP_dims is None else [Q] + list(MLP_dims) + [sp['X_dim']]
            self.nets.append(_mlp(dims))
            n_init = {'one': 1, 'all': n_seq, 'mlp': 0}[self.means_mode]
            if n_init:
                if self.minibatch_inference:
                    im = torch.zeros(n_init, sp['X_win'], sp['X_dim'])                                          # layers.py:412-416
                else:
                    im = torch.randn(n_init, sp['X_win'], sp['X_dim']) * 1e-2                                   # layers.py:437
                self.init_mean.append(nn.Parameter(im))
                self.init_var_raw.append(nn.Parameter(_inv_softplus(torch.full((n_init, sp['X_win'], sp['X_dim']), float(X_variance)))))
            shape = {'rows': (n_seq, sp['N'], sp['X_dim']), 'all': (n_seq, 1, sp['X_dim']), 'one': (1, 1, sp['X_dim'])}[self.vars_mode]
            self.var_raw.append(nn.Parameter(_inv_softplus(torch.full(shape, float(X_variance)))))
        if self.means_mode == 'mlp':
            assert low_init is not None, "init_xs_means='mlp' needs the first rows of the observations (low_init)"
            self.register_buffer('low_init', torch.as_tensor(low_init, dtype=torch.float64).reshape(n_seq, -1))
            q_low = self.low_init.shape[1]
            nets = []
            for sp in reversed(layers):                                          # lowest hidden layer first (model.py:121-124)
                d_out = sp['X_win'] * sp['X_dim']
                nets.append(_mlp([q_low] + ([q_low * q_low] if init_MLP_dims is None else list(init_MLP_dims)) + [d_out]))
                q_low = d_out
            self.init_nets.extend(reversed(nets))                                # stored like specs: top hidden layer first
        self.double()
        self.softplus = nn.Softplus()

    handle = None           # rgp_amd.capi.Handle: set by the model; without it only forward_loop is available

    # ------------------------------------------------------------------------------------------------------------
    def per_sequence_parameters(self):
        """Parameters whose leading dimension runs over the sequences of the data set (a minibatch touches some rows only)."""
        out = []
        if self.means_mode == 'all':
            out += list(self.init_mean) + list(self.init_var_raw)
        if self.vars_mode in ('rows', 'all'):
            out += list(self.var_raw)
        return out

    def shared_parameters(self):
        """Parameters every sequence depends on: replicated across data-parallel ranks, gradients summed."""
        per_seq = set(id(p) for p in self.per_sequence_parameters())
        return [p for p in self.parameters() if id(p) not in per_seq]

    def _batch(self, idx):
        if isinstance(idx, slice):
            return len(range(*idx.indices(self.n_seq)))
        return int(idx.numel())

    def _initial_rows(self, idx):
        """(means, variances) of the first X_win rows of every hidden layer for the sequences ``idx``, lists like specs."""
        B = self._batch(idx)
        ms, vs = [], []
        if self.means_mode == 'mlp':
            low = self.low_init[idx]
            ms = [None] * len(self.specs)
            for k in range(len(self.specs) - 1, -1, -1):                        # bottom-up (model.py:167-171)
                sp = self.specs[k]
                out = self.init_nets[k](low)
                ms[k] = out.view(B, sp['X_win'], sp['X_dim'])
                low = out
            vs = [torch.full((B, sp['X_win'], sp['X_dim']), self.X_variance, dtype=torch.float64, device=low.device)
                  for sp in self.specs]
            return ms, vs
        for k, sp in enumerate(self.specs):
            if self.means_mode == 'one':
                ms.append(self.init_mean[k].expand(B, -1, -1))
                vs.append(self.softplus(self.init_var_raw[k]).expand(B, -1, -1))
            else:
                ms.append(self.init_mean[k][idx])
                vs.append(self.softplus(self.init_var_raw[k][idx]))
        return ms, vs

    def _row_variances(self, k, idx):
        sp = self.specs[k]
        if self.vars_mode == 'rows':
            return self.softplus(self.var_raw[k][idx])
        v = self.softplus(self.var_raw[k] if self.vars_mode == 'one' else self.var_raw[k][idx])
        return v.expand(self._batch(idx), sp['N'], sp['X_dim'])                 # layers.py:657: one vector for all rows

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
        init_m, init_v = self._initial_rows(idx)
        for k, sp in enumerate(self.specs):
            ok, lins, dims = self._scan_ok(k)
            if not ok:
                m = self._loop_layer(k, up, init_m[k])
            else:
                wb = []
                for l in lins:
                    wb += [l.weight, l.bias]
                m = _MlpScanFn.apply(self.handle, sp, dims, init_m[k], up if sp['U_win'] > 0 else None, *wb)
            v = torch.cat((init_v[k], self._row_variances(k, idx)), dim=1)
            means.append(m)
            variances.append(v)
            up = m
        return means, variances

    def _loop_layer(self, k, up, init_mean):
        sp = self.specs[k]
        Xw, N, Uw = sp['X_win'], sp['N'], sp['U_win']
        rows = list(init_mean.unbind(dim=1))                                # X_win tensors (batch, X_dim)
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
        init_m, init_v = self._initial_rows(idx)
        for k, sp in enumerate(self.specs):
            m = self._loop_layer(k, up, init_m[k])
            v = torch.cat((init_v[k], self._row_variances(k, idx)), dim=1)
            means.append(m)
            variances.append(v)
            up = m                                                               # the layer below takes these means as its input
        return means, variances


class RowAdam(object):
    """Adam (ascent) for the recursion parameters.  A per-sequence parameter keeps one step count per sequence and only the
    rows of the sequences in the minibatch move, so the moments of an untouched sequence neither decay nor push it."""

    def __init__(self, encoder, b1=0.9, b2=0.999, eps=1e-8):
        self.enc, self.b1, self.b2, self.eps = encoder, b1, b2, eps
        per_seq = set(id(p) for p in encoder.per_sequence_parameters())
        self.state = []
        for p in encoder.parameters():
            rows = id(p) in per_seq
            t = torch.zeros(p.shape[0], dtype=torch.float64, device=p.device) if rows else [0]
            self.state.append((p, rows, torch.zeros_like(p), torch.zeros_like(p), t))

    @torch.no_grad()
    def step(self, idx, lr):
        for p, rows, m, v, t in self.state:
            if p.grad is None:
                continue
            if rows:
                g = p.grad[idx]
                t[idx] += 1
                tt = t[idx].view(-1, *([1] * (p.dim() - 1)))
                mi = m[idx].mul_(self.b1).add_(g, alpha=1 - self.b1)
                vi = v[idx].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                m[idx], v[idx] = mi, vi
                upd = (mi / (1 - self.b1 ** tt)) / ((vi / (1 - self.b2 ** tt)).sqrt() + self.eps)
                p[idx] = p[idx] + lr * upd
            else:
                t[0] += 1
                m.mul_(self.b1).add_(p.grad, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
                tt = t[0]
                p.add_(lr * (m / (1 - self.b1 ** tt)) / ((v / (1 - self.b2 ** tt)).sqrt() + self.eps))
