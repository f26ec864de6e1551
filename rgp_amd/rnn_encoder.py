"""Sequence-encoder back-constraint: the recognition model of autoreg.DeepAutoreg_rnn
(/root/reference/autoreg/rnn_encoder.py:21-150, model.py:403-440,525-551).  It stays in PyTorch-ROCm as the north star
prescribes: an fp64 stack of RNN / LSTM / GRU layers maps the raw observations (and controls) of every sequence to the
mean and softplus-variance of each hidden layer's q(X) for ALL time steps; the GP layers then run on the device through
librgp_hip.so and hand back dL/dq(X), which torch.autograd pushes into the encoder weights.

Layer l of the encoder (l = 0 is the one above the observations) takes [mean | variance] of layer l-1's output (layer 0
takes the raw [Y | U]) -- rnn_encoder.py:98-123."""
import torch
import torch.nn as nn


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

    def forward(self, x):                       # x: (seq_len, batch, input_dim); zero initial state (h_0_type='zero')
        out, _ = self.rnn(x)
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
