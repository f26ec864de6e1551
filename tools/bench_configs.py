"""Timing of one objective+gradient evaluation (or SVI step) for the geometries of BASELINE.json's configs 1-3 on synthetic
data of the same shape (config 4 is bench.py; config 5 needs M > 128, not in this build).  Writes one JSON object per config:
  python tools/bench_configs.py [--cpu] > profiles/r1_configs.json
--cpu also times the NumPy oracle on the same problem (host cores)."""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from rgp_amd.model import DeepAutoreg, DeepAutoreg_rnn


def data(n_seq, T, du, dy, seed=0):
    rng = np.random.default_rng(seed)
    U = [rng.standard_normal((T, du)) for _ in range(n_seq)]
    Y = [np.tanh(np.cumsum(rng.standard_normal((T, dy)), 0) * 0.1) + 0.05 * rng.standard_normal((T, dy)) for _ in range(n_seq)]
    return U, Y


def timeit(fn, sync, warm=10, reps=30):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def oracle_time(wins, Y, U, U_win, nDims, M, inference):
    from oracle import model as om
    o = om.DeepAutoreg(list(wins), Y, U=U, U_win=U_win, nDims=list(nDims), X_variance=0.05, num_inducing=M, init='Y',
                       inference_method=inference, rng=np.random.default_rng(1), fast_psi=True, chunk=256)
    o.elbo_and_grads()
    t0 = time.perf_counter()
    o.elbo_and_grads()
    return time.perf_counter() - t0


def main():
    cpu = '--cpu' in sys.argv
    out = []
    # config 1: Actuator-like, 1 hidden layer, M = 20, collapsed bound, full batch
    U, Y = data(1, 1024, 1, 1)
    m = DeepAutoreg([0, 10], Y, U=U, U_win=10, nDims=[1, 1], X_variance=0.05, num_inducing=20, init='Y', inference_method=None, seed=1)
    t = timeit(lambda: m.evaluate(), m.handle.sync)
    r = dict(config='1: actuator-like (T=1024, 1-D in/out), 1 hidden layer, M=20, collapsed bound, full batch', ms_per_evaluation=t * 1e3)
    if cpu:
        r['cpu_oracle_ms'] = oracle_time([0, 10], Y, U, 10, [1, 1], 20, None) * 1e3
    out.append(r)
    # config 2: Ballbeam-like, 2 hidden layers, M = 50
    U, Y = data(1, 1000, 1, 1)
    m = DeepAutoreg([0, 10, 10], Y, U=U, U_win=10, nDims=[1, 1, 1], X_variance=0.05, num_inducing=50, init='Y', inference_method=None, seed=1)
    t = timeit(lambda: m.evaluate(), m.handle.sync)
    r = dict(config='2: ballbeam-like (T=1000), 2 hidden layers, M=50, collapsed bound, full batch', ms_per_evaluation=t * 1e3)
    if cpu:
        r['cpu_oracle_ms'] = oracle_time([0, 10, 10], Y, U, 10, [1, 1, 1], 50, None) * 1e3
    out.append(r)
    # config 3: walk_run-like mocap (59-D outputs), 2 layers + sequence-encoder back-constraint, M = 100, SVI
    U, Y = data(8, 512, 1, 59)
    m = DeepAutoreg_rnn([0, 10], Y, U=U, U_win=10, nDims=[59, 3], num_inducing=100, rnn_type='gru', rnn_hidden_dims=[20],
                        inference_method='svi', mb_inf_tot_data_size=8, seed=1)
    t = timeit(lambda: m.svi_step(slice(0, 8), lr=1e-3), m.handle.sync, warm=3, reps=10)
    out.append(dict(config='3: mocap-like (8 x T=512, 59-D outputs), GP layers + GRU encoder back-constraint, M=100, SVI',
                    ms_per_svi_step=t * 1e3))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
