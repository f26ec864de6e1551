"""Sequence-encoder back-constraint: the recognition model of autoreg.DeepAutoreg_rnn
(/root/reference/autoreg/rnn_encoder.py:21-150, model.py:403-440,525-551).  It stays in PyTorch-ROCm as the north star
prescribes: an fp64 stack of RNN / LSTM / GRU layers maps the raw observations (and controls) of every sequence to the
mean and softplus-variance of each hidden layer's q(X) for ALL time steps; the GP layers then run on the device through
librgp_hip.so and hand back dL/dq(X), which torch.autograd pushes into the encoder weights.

Layer l of the encoder (l = 0 is the one above the observations) takes [mean | variance] of layer l-1's output (layer 0
takes the raw [Y | U]) -- rnn_encoder.py:98-123.

The time loop of every layer (Elman/tanh RNN, LSTM or GRU cell) runs as a per-sequence persistent scan in librgp_hip.so
(rgp_rnn_scan_forward / _backward, rgp_amd/csrc/rgp_rnn.hip) when the model attached its handle: the input projections of all
time steps, the heads and every weight gradient stay (batched) torch GEMMs, only the strictly sequential recurrence moves --
the per-time-step kernel launches of the fp64 torch GRU were 95 % of a config-3 SVI step.  The parameters remain those of the
nn.RNN / nn.LSTM / nn.GRU module, so the torch path (hidden width beyond the scan's limit, no handle) and the scan are
interchangeable and are compared in tests/test_gpu_rnn.py."""
import torch
import torch.nn as nn


def _splitk_tn(a, b, chunk=256):
    """a^T b for tall a (rows x p), b (rows x q) with a tiny p x q result: one GEMM would put the whole row sum on a single
    workgroup (0.22 ms for 4096 rows), so the sum is split into `chunk`-row pieces (batched GEMM) and added afterwards."""
    rows = a.shape[0]
    nch = rows // chunk
    body = nch * chunk
    out = a[body:].t().mm(b[body:])
    if nch > 0:
        out = out + torch.bmm(a[:body].reshape(nch, chunk, -1).transpose(1, 2), b[:body].reshape(nch, chunk, -1)).sum(0)
    return out


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b for x of shape (rows, in) with many rows and few features; the weight gradient uses _splitk_tn."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return torch.addmm(b, x, W.t())

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        return g.mm(W), _splitk_tn(g, x), g.sum(0)


def tall_linear(x, W, b):
    """Linear map over the last dimension of x (any leading shape)."""
    lead = x.shape[:-1]
    return _TallLinear.apply(x.reshape(-1, x.shape[-1]), W, b).view(*lead, W.shape[0])


class _RnnScanFn(torch.autograd.Function):
    """hseq = cell recurrence(gi; W_hh, b_hh) with zero initial state, gi = W_ih x + b_ih already applied; (T, B, .) layouts."""

    @staticmethod
    def forward(ctx, handle, cell, gi, W_hh, b_hh):
        from . import capi
        gi_c, W_c, b_c = gi.contiguous(), W_hh.contiguous(), b_hh.contiguous()
        hseq, gates = capi.rnn_scan_forward(handle, cell, gi_c, W_c, b_c)
        ctx.handle, ctx.cell, ctx.has_gates = handle, cell, gates is not None
        ctx.save_for_backward(W_c, b_c, hseq, *(() if gates is None else (gates,)))
        return hseq

    @staticmethod
    def backward(ctx, g_out):
        from . import capi
        saved = ctx.saved_tensors
        W_c, b_c, hseq = saved[:3]
        gates = saved[3] if ctx.has_gates else None
        dgi, dgh = capi.rnn_scan_backward(ctx.handle, ctx.cell, W_c, b_c, hseq, gates, g_out.contiguous())
        T, B, H = hseq.shape
        # dW_hh = sum_t dgh_t h_{t-1}^T (h_{-1} = 0): split-K batched GEMM, the result is only G H x H
        gW = _splitk_tn(dgh[1:].reshape(-1, dgh.shape[2]), hseq[:-1].reshape(-1, H))
        return None, None, dgi, gW, dgh.sum((0, 1))


class MeanVarRNN(nn.Module):
    """One recurrent layer with a linear mean head and a softplus variance head (rnn_encoder.py:21-83)."""

    def __init__(self, input_dim, output_dim, hidden_dim, rnn_type='rnn', bidirectional=False):
        nn.Module.__init__(self)
        cls = {'rnn': nn.RNN, 'lstm': nn.LSTM, 'gru': nn.GRU}.get(rnn_type)
        if cls is None:
            raise ValueError('Unknown rnn type')
        self.rnn = cls(input_size=input_dim, hidden_size=hidden_dim, num_layers=1, bidirectional=bidirectional)
        ndir = 2 if bidirectional else 1
        self.linear_mean = nn.Linear(hidden_dim * ndir, output_dim)
        self.linear_var = nn.Linear(hidden_dim * ndir, output_dim)
        self.soft_plus = nn.Softplus()

    handle = None               # rgp_amd.capi.Handle (set by the model): the time loops then run as device scans
    use_scan = True

    def _scan_cell(self):
        """(cell id, gate rows per unit) when this layer's recurrence fits the device scan, else None."""
        from . import _lib
        rnn = self.rnn
        if isinstance(rnn, nn.GRU):
            cell, G = _lib.CELL_GRU, 3
        elif isinstance(rnn, nn.LSTM):
            cell, G = _lib.CELL_LSTM, 4
        elif isinstance(rnn, nn.RNN) and rnn.nonlinearity == 'tanh':
            cell, G = _lib.CELL_RNN_TANH, 1
        else:
            return None
        Hh = rnn.hidden_size                       # G H lanes per step, and W_hh lives in LDS next to the step's vectors
        fits = G * Hh <= 256 and (G * Hh * (Hh | 1) + 2 * G * Hh + 2 * Hh) * 8 <= 160 * 1024
        return cell if fits else None

    def _scan(self, x, cell):
        rnn, outs = self.rnn, []
        for sfx in ([''] if not rnn.bidirectional else ['', '_reverse']):
            W_ih, W_hh = getattr(rnn, 'weight_ih_l0' + sfx), getattr(rnn, 'weight_hh_l0' + sfx)
            b_ih, b_hh = getattr(rnn, 'bias_ih_l0' + sfx), getattr(rnn, 'bias_hh_l0' + sfx)
            xs = x.flip(0) if sfx else x
            gi = tall_linear(xs, W_ih, b_ih)
            hs = _RnnScanFn.apply(self.handle, cell, gi, W_hh, b_hh)
            outs.append(hs.flip(0) if sfx else hs)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)

    def forward(self, x):                       # x: (seq_len, batch, input_dim); zero initial state (h_0_type='zero')
        cell = self._scan_cell() if (self.handle is not None and self.use_scan) else None
        if cell is not None:
            out = self._scan(x, cell)
        else:
            out = self.rnn(x)[0]
        if out.is_cuda and out.shape[0] * out.shape[1] >= 1024:      # many rows, tiny heads: split-K weight gradients
            return (tall_linear(out, self.linear_mean.weight, self.linear_mean.bias),
                    self.soft_plus(tall_linear(out, self.linear_var.weight, self.linear_var.bias)))
        return self.linear_mean(out), self.soft_plus(self.linear_var(out))


class SeqEncoder(nn.Module):
    """Stack of MeanVarRNN layers, lower hidden layer first (rnn_encoder.py:86-150)."""

    def __init__(self, input_dims, output_dims, hidden_dim, rnn_type='rnn', bidirectional=False):
        nn.Module.__init__(self)
        assert len(input_dims) == len(output_dims)
        self.layers = nn.ModuleList()
        for l, (din, dout) in enumerate(zip(input_dims, output_dims)):
            self.layers.append(MeanVarRNN(din if l == 0 else 2 * din, dout, hidden_dim, rnn_type, bidirectional))
        self.double()

    def forward(self, x0):
        means, variances = [], []
        m, v = x0, None
        for l, layer in enumerate(self.layers):
            m, v = layer(m if l == 0 else torch.cat((m, v), dim=2))
            means.append(m)
            variances.append(v)
        return means, variances
