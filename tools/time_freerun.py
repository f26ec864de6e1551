"""Times the moment-matching free-run cascade (rgp_freerun) on the config-4 geometry: python tools/time_freerun.py [n_seq] [T]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rgp_amd.synthetic import CONFIG4 as cfg, generate_sequences
from rgp_amd.model import DeepAutoreg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Us, Ys = generate_sequences(n, T, cfg['D_y'], cfg['D_u'])
m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                init='Y', inference_method='svi', mb_inf_tot_data_size=n, seed=3)
for L in m.layers:
    m.set_param(L, 'inv_l', np.full(L.Q, 0.25)); m.set_param(L, 'noise', 0.05)
m.evaluate(); m.handle.sync()
m.freerun(Us); torch.cuda.synchronize()
t0 = time.perf_counter()
out = m.freerun(Us)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('free-run of %d sequences x %d steps through %d layers: %.1f ms (%.1f us per time step per layer)' % (n, T, len(m.layers), dt * 1e3, dt * 1e6 / T / len(m.layers)))
