"""Host time to ENQUEUE one pipelined SVI step (no device wait) vs the device time per step: python tools/host_overhead.py [mb]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rgp_amd.synthetic import CONFIG4 as cfg, generate_sequences
from rgp_amd.model import DeepAutoreg
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Us, Ys = generate_sequences(mb, 4096, cfg['D_y'], cfg['D_u'])
m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                init='Y', inference_method='svi', mb_inf_tot_data_size=256, seed=3)
for L in m.layers:
    m.set_param(L, 'inv_l', np.full(L.Q, 0.25)); m.set_param(L, 'noise', 0.05)
for _ in range(3): m.svi_step(slice(0, mb), lr=1e-3)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): m.svi_step(slice(0, mb), lr=1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('minibatch %d sequences: host enqueue %.2f ms/step, total %.2f ms/step' % (mb, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
