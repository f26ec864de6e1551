"""MLP back-constraint at config-4 sizes (32 sequences x T = 4096, windows 20, networks [40, 80, 40, 1]): time of the device
scans and of one full SVI step of DeepAutoreg_mlp; the torch-graph recursion is timed at T = 276 for comparison.
python tools/time_mlp.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rgp_amd.synthetic import CONFIG4 as cfg, generate_sequences
from rgp_amd.model import DeepAutoreg_mlp


def build(n, T):
    Us, Ys = generate_sequences(n, T, cfg['D_y'], cfg['D_u'])
    m = DeepAutoreg_mlp(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05, num_inducing=cfg['M'],
                        inference_method='svi', mb_inf_tot_data_size=256, seed=3)
    for L in m.layers:
        m.set_param(L, 'inv_l', np.full(L.Q, 0.25)); m.set_param(L, 'noise', 0.05)
    return m


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


m = build(32, 4096)
enc = m.encoder
print('networks:', [[l.in_features] + [l.out_features] for net in enc.nets for l in net if hasattr(l, 'weight')])
def fwd_bwd():
    enc.zero_grad()
    means, variances = enc(m.controls)
    (sum((x * x).sum() for x in means)).backward()
print('device scans, forward only  : %.2f ms' % timeit(lambda: enc(m.controls), 5))
print('device scans, forward + back: %.2f ms' % timeit(fwd_bwd, 5))
print('full SVI step (32 x 4096)   : %.2f ms' % timeit(lambda: m.svi_step(slice(0, 32), lr=1e-3), 5))
m.handle.sync()
m2 = build(32, 276)
enc2 = m2.encoder
def loop_fwd_bwd():
    enc2.zero_grad()
    means, variances = enc2.forward_loop(m2.controls)
    (sum((x * x).sum() for x in means)).backward()
def scan_fwd_bwd():
    enc2.zero_grad()
    means, variances = enc2(m2.controls)
    (sum((x * x).sum() for x in means)).backward()
print('T = 276: torch graph forward + back %.1f ms, device scans %.2f ms' % (timeit(loop_fwd_bwd, 2), timeit(scan_fwd_bwd, 5)))
