"""Synthetic workload of BASELINE.json's configs: sequences, and the reference's canonical
initialisation recipe (svi_experiments/rgp_experiments.py:632-645)."""
import numpy as np

CONFIG4 = dict(name='config4: synthetic T=4096 D_y=16 D_u=1, 2 hidden layers H=[1,1], wins=[0,20,20], U_win=20, M=100, '
                    'SVI minibatch 32 of 256 sequences',
               n_total=256, T=4096, D_y=16, D_u=1, wins=[0, 20, 20], nDims=[16, 1, 1], U_win=20, M=100, minibatch=32)


# the other reading of "D=16": the hidden layers are 16-dimensional too, so the lag windows are 340 / 640 / 320 columns wide
CONFIG4W = dict(name='config4-wide: synthetic T=4096 D_y=16 D_u=1, 2 hidden layers H=[16,16], wins=[0,20,20], U_win=20, M=100 '
                     '(windows of 340 / 640 / 320 columns), SVI minibatch 32 of 256 sequences',
                n_total=256, T=4096, D_y=16, D_u=1, wins=[0, 20, 20], nDims=[16, 16, 16], U_win=20, M=100, minibatch=32)


CONFIG5 = dict(name='config5: synthetic T=65536 D_y=64 D_u=1, 3 hidden layers H=[1,1,1], wins=[0,20,20,20], U_win=20, M=500, '
                    'SVI minibatch 8 of 16 sequences',
               n_total=16, T=65536, D_y=64, D_u=1, wins=[0, 20, 20, 20], nDims=[64, 1, 1, 1], U_win=20, M=500, minibatch=8)


def generate_sequences(n_seq, T, D_y, D_u=1, seed=20240, first=0):
    """u_t ~ N(0,1); x_t = 0.7 x_{t-1} - 0.2 x_{t-2} + tanh(u_{t-1}) + 0.05 eps; Y = tanh(x) w + 0.1 eps; z-scored.
    ``first`` skips that many sequences of the same global stream so shards of a data-parallel job are disjoint pieces of
    ONE dataset (statistics for the z-score are those of the first 256 sequences' generator, fixed constants)."""
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((1, D_y))
    Us, Ys = [], []
    for s in range(first, first + n_seq):                 # every sequence has its own generator: skipping costs nothing
        r = np.random.default_rng([seed, s])
        u = r.standard_normal((T, D_u))
        e = r.standard_normal(T)
        if T > 8192:                                      # long sequences: the AR(2) recursion as a linear filter
            from scipy.signal import lfilter
            b = np.zeros(T)
            b[2:] = np.tanh(u[1:T - 1, 0]) + 0.05 * e[2:]
            x = lfilter([1.0], [1.0, -0.7, 0.2], b)
        else:
            x = np.zeros(T)
            for t in range(2, T):
                x[t] = 0.7 * x[t - 1] - 0.2 * x[t - 2] + np.tanh(u[t - 1, 0]) + 0.05 * e[t]
        y = np.tanh(x)[:, None] * w + 0.1 * r.standard_normal((T, D_y))
        Us.append(u)
        Ys.append(y)
    # z-score with per-dataset statistics of what was generated (drivers normalise, rgp_experiments.py:52-63)
    Ucat, Ycat = np.concatenate(Us), np.concatenate(Ys)
    um, us, ym, ys = Ucat.mean(0), Ucat.std(0), Ycat.mean(0), Ycat.std(0)
    return [(u - um) / us for u in Us], [(y - ym) / ys for y in Ys]
