"""Times pipelined SVI steps of the config-4 geometry with a different number of inducing points:  python tools/time_m.py M
(RGP_PSI2_NO_PACKED=1 selects the rotating four-team psi2 kernel for comparison)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rgp_amd.synthetic import CONFIG4 as cfg, generate_sequences
from rgp_amd.model import DeepAutoreg
M = int(sys.argv[1])
Us, Ys = generate_sequences(8, 4096, cfg['D_y'], cfg['D_u'])
m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=M,
                init='Y', inference_method='svi', mb_inf_tot_data_size=8, seed=3)
for L in m.layers:
    m.set_param(L, 'inv_l', np.full(L.Q, 0.25)); m.set_param(L, 'noise', 0.05)
for _ in range(2): m.svi_step(slice(0, 8), lr=1e-3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m.svi_step(slice(0, 8), lr=1e-3)
torch.cuda.synchronize(); print('M', M, 'ms/step', (time.perf_counter() - t0) / 5 * 1e3)
