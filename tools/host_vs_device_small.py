"""Host enqueue time vs total time of one collapsed-bound evaluation at the sizes of BASELINE configs 1 and 2."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from bench_configs import data
from rgp_amd.model import DeepAutoreg
for name, wins, nD, M, T in (('config 1', [0, 10], [1, 1], 20, 1024), ('config 2', [0, 10, 10], [1, 1, 1], 50, 1000)):
    U, Y = data(1, T, 1, 1)
    m = DeepAutoreg(wins, Y, U=U, U_win=10, nDims=nD, X_variance=0.05, num_inducing=M, init='Y', inference_method=None, seed=1)
    for _ in range(5): m.evaluate()
    m.handle.sync()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): m.evaluate()
    t1 = time.perf_counter()
    m.handle.sync()
    t2 = time.perf_counter()
    print('%s: host enqueue %.3f ms, total %.3f ms per evaluation' % (name, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
