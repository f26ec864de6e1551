"""A/B of the psi2 term for M > 128: blocked sweep kernels (mode 0) against the blocked pair form (mode 2), timed through the unit
entry point rgp_psi_backward with dL/dpsi2 only (prep of Z and rows, psi2 forward + backward, finalisation), HIP events via torch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from rgp_amd.capi import Handle

H = Handle()
rng = np.random.default_rng(0)
def dev(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=H.device)
print('%8s %5s %4s %12s %12s %8s' % ('rows', 'M', 'Q', 'sweeps ms', 'pair ms', 'ratio'))
for M, Q in ((500, 40), (500, 20), (200, 40), (1000, 40)):
    for N in (512, 1024, 2048, 4096, 16384, 65536):
        if M == 1000 and N > 16384: continue
        mu = dev(rng.standard_normal((N, Q)) + 0.5); S = dev(0.02 + 0.3 * rng.random((N, Q)))
        Z = dev(rng.standard_normal((M, Q)) * 1.2 + 0.5); var = dev([0.8]); il = dev((0.3 + rng.random(Q)) * (8.0 / Q))
        w2 = dev(rng.standard_normal((M, M)))
        ms = {}
        for mode in (0, 2):
            H.set_psi2_pair(mode)
            for rep in range(2):
                H.psi_backward(mu, S, Z, var, il, 0.0, None, w2); H.sync()
            t0 = time.perf_counter()
            reps = 3
            for rep in range(reps):
                H.psi_backward(mu, S, Z, var, il, 0.0, None, w2)
            H.sync()
            ms[mode] = (time.perf_counter() - t0) / reps * 1e3
        print('%8d %5d %4d %12.2f %12.2f %8.2f' % (N, M, Q, ms[0], ms[2], ms[0] / ms[2]), flush=True)
