"""L-BFGS function + gradient evaluations per second on the two system-identification notebooks' models, the only throughput
figures the reference publishes (examples/IdentificationTraining_Actuator.ipynb cell 16: 1.9 evals/s; ..._Ballbeam.ipynb cell 16:
4.5 evals/s; examples/Svi/*: IE5 and Drive; authors' laptop CPU, hardware unstated).  Same geometry and data
(tests/golden/sysid_data.npz), the notebook's bound, full batch; an evaluation is DeepAutoreg.objective_and_grad: parameters host -> device, all layer steps, gradients device -> host.
The NumPy oracle is timed next to it on this host's cores.

    PYTHONPATH=. python tools/sysid_evals.py > profiles/rX_sysid_evals.txt        (on the GPU box)
"""
import time

import numpy as np

from oracle import model as om
from rgp_amd.model import DeepAutoreg

z = np.load('tests/golden/sysid_data.npz')
# (title, series, geometry, bound, published evals/s, parameter count printed by the notebook)
CASES = [('Actuator', 'actuator', dict(wins=[0, 10], U_win=10, M=100), None, 1.9, 4058),
         ('Ballbeam', 'ballbeam', dict(wins=[0, 20], U_win=20, M=50), None, 4.5, 4064),
         ('IE5 (SVI bound)', 'ie5', dict(wins=[0, 5], U_win=5, M=50), 'svi', 12.4, 6471),     # examples/Svi/IdentificationTraining_IE5-SVI.ipynb cell 16
         ('IE5', 'ie5', dict(wins=[0, 5], U_win=5, M=50), None, 12.5, 1369),                  # examples/Svi/IdentificationTraining_IE5.ipynb cell 16
         ('Drive', 'drive', dict(wins=[0, 10], U_win=10, M=50), None, 11.2, 2034),            # examples/Svi/IdentificationTraining_Driver-SVI.ipynb cell 16
         ('Drive, 2 hidden layers', 'drive', dict(wins=[0, 10, 10], U_win=10, M=50), None, 6.2, None)]   # ..._Driver.ipynb cell 16
for title, key, cfg, bound, published, n_ref in CASES:
    k = int(z[key + '_split'])
    u, y = z[key + '_u'], z[key + '_y']
    u = (u - u[:k].mean()) / u[:k].std()
    y = (y - y[:k].mean()) / y[:k].std()
    kw = dict(U=[u[:k]], U_win=cfg['U_win'], nDims=[1] * len(cfg['wins']), X_variance=0.05, num_inducing=cfg['M'], init='Y')
    g = DeepAutoreg(cfg['wins'], [y[:k]], inference_method=bound, seed=1, **kw)
    x = g.get_optimizer_array()
    for _ in range(3):
        g.objective_and_grad(x)
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        f, gr = g.objective_and_grad(x + 1e-6 * i)
    dt = time.perf_counter() - t0
    o = om.DeepAutoreg(cfg['wins'], [y[:k]], inference_method=bound, n_total=None, rng=np.random.default_rng(2), fast_psi=True,
                       chunk=128, **kw)
    xo = o.get_optimizer_array()
    o.objective_and_grad(xo)
    t1 = time.perf_counter()
    for i in range(3):
        o.objective_and_grad(xo + 1e-6 * i)
    do = (time.perf_counter() - t1) / 3
    print('%-22s wins %s U_win %d M %d, %d parameters (notebook: %s): device %.0f evals/s (%.2f ms each); NumPy oracle on this host '
          '%.1f evals/s; published by the reference (other hardware) %.1f evals/s'
          % (title, cfg['wins'], cfg['U_win'], cfg['M'], x.size, n_ref, n / dt, 1e3 * dt / n, 1.0 / do, published))
