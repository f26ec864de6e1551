"""MLP back-constraint of the hidden layers: the recognition recursion of autoreg.layers.Layer
with ``back_cstr=True`` (/root/reference/autoreg/layers.py:398-441 _init_encoder, :623-666 _encoder_freerun, :668-756
_encoder_update_gradient) and its network (/root/reference/autoreg/mlp.py:13-175, Theano there).

For every hidden layer the latent MEANS of the predicted rows are no free parameters: row X_win + n is the output of a
tanh MLP applied to the layer's own previous X_win means and the U_win-row window of the signal above (controls for the top
hidden layer, the latent means of the layer above otherwise),

    mean[X_win + n] = MLP([mean[n : n + X_win].flat, up_mean[U_off + n : U_off + n + U_win].flat]),  U_off = up_len - N - U_win + 1

so the free parameters are the first X_win rows (init_Xs: mean and variance), the variances of the other rows (Xs_var) and the
network weights.  The reference back-propagates through the recursion by hand (reverse loop, layers.py:705-712); here the
recursion is a PyTorch-ROCm fp64 graph and torch.autograd does that, driven by dL/dq(X) from librgp_hip.so."""
import math

import torch
import torch.nn as nn


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

    def forward(self, controls, idx=slice(None)):
        """controls: (n_seq, u_len, D_u) tensor or None.  Returns (means, variances), lists top hidden layer first, each
        (batch, X_win + N, X_dim)."""
        means, variances = [], []
        up = None if controls is None else controls[idx]
        for k, sp in enumerate(self.specs):
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
            m = torch.stack(rows, dim=1)
            v = torch.cat((self.softplus(self.init_var_raw[k][idx]), self.softplus(self.var_raw[k][idx])), dim=1)
            means.append(m)
            variances.append(v)
            up = m                                                               # the layer below takes these means as its input
        return means, variances
