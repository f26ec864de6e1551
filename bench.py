#!/usr/bin/env python3
"""bench.py -- SVI steps/s of the recurrent-GP hot path on BASELINE.json's config 4 (synthetic T=4096, D_y=16, 2 hidden
layers, M=100, SVI minibatch of 32 sub-sequences), 1..8 MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one SVI step: minibatch (32 sequences, sharded over the ranks) -> ELBO + every gradient of all three GP
layers -> all-reduce of each layer's global gradients as soon as they exist (N>1) -> Adam on the global parameters and on the touched sequences'
latent parameters.  Inputs are resident in HBM before the timed region.  Scaling is STRONG: the minibatch is fixed at
32 sequences and split over the ranks (32/N each), as BASELINE.json's config 4 names it.
Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (the fused psi2 kernel, timed with HIP
events inside librgp_hip.so on the launch stream) and `cpu_baseline` (the NumPy oracle timed on this host, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X public fp64 peak (vector = matrix); the local guide lists none. Measured by
                             # tools/f64_peak.hip / f64_lat.hip on this pool (profiles/r1_f64_peak.txt, r1_f64_lat.txt): v_fma_f64 63.5 TF at 2 waves/SIMD,
                             # 68.2 TF at 8; v_mfma_f64_16x16x4 45.6 / 49.4 TF.


def measured_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs; FETCH_SIZE doubled as the gfx950 guide prescribes for wide coalesced reads).  None if absent."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r1_psi2_rowsN_traffic_raw.json')))
        return (2.0 * d['FETCH_SIZE']['avg_per_launch_kb'] + d['WRITE_SIZE']['avg_per_launch_kb']) * 1024.0
    except Exception:
        return None


def init_like_reference(m, Ys):
    """The canonical initialisation recipe (svi_experiments/rgp_experiments.py:632-645)."""
    import torch
    means = m.means.cpu().numpy()
    yvar = float(np.concatenate(Ys).var())
    for L in m.layers:
        Xm = m._host_X_mean(L.index, m._Ys_host, m._Us_host, means)
        il = np.mean(1.0 / ((Xm.max(0) - Xm.min(0)) / np.sqrt(2.0)))
        m.set_param(L, 'inv_l', np.full(L.Q, il))
        m.set_param(L, 'noise', 0.01 * yvar)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-update', action='store_true', help='debug: skip the Adam update (kernel ablation runs); INVALID as a benchmark')
    ap.add_argument('--no-prefetch', action='store_true', help='debug: do not prepare the next minibatch during the running step')
    ap.add_argument('--n-total', type=int, default=256, help='sequences in the dataset (config 4: 256)')
    ap.add_argument('--T', type=int, default=4096)
    ap.add_argument('--minibatch', type=int, default=32)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rgp_amd.synthetic import CONFIG4, generate_sequences
    from rgp_amd.model import DeepAutoreg

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, 'launch one process per GPU (torch.distributed.run --nproc-per-node N) with --gpus N'
    # RGP_BENCH_ONE_GPU=1 (debugging aid, never set by the driver): all ranks share GPU 0 and talk over gloo, which lets the
    # N > 1 code path of this file be exercised on a single-GPU box; the numbers of such a run mean nothing
    one_gpu = os.environ.get('RGP_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    cfg = dict(CONFIG4)
    n_total, T, mb = a.n_total, a.T, a.minibatch
    assert n_total % world == 0 and mb % world == 0
    n_loc, mb_loc = n_total // world, mb // world

    # ---- synthetic data: this rank's shard of the dataset (sequences are independent: no data-path exchange) ----
    Us, Ys = generate_sequences(n_loc, T, cfg['D_y'], cfg['D_u'], seed=20240, first=rank * n_loc)
    m = DeepAutoreg(cfg['wins'], Ys, U=Us, U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05,
                    num_inducing=cfg['M'], init='Y', inducing_init='rand', inference_method='svi',
                    mb_inf_tot_data_size=n_total, seed=20241)
    m._Ys_host = [y[cfg['U_win']:] for y in Ys]
    m._Us_host = [u[:-1] for u in Us]
    init_like_reference(m, Ys)
    if world > 1:                      # identical global parameters everywhere (q(U) init depends on local data)
        dist.broadcast(m.param_array[:m.n_global], src=0)
    allreduce = (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)) if world > 1 else None
    m.handle.set_profiling(False)

    n_mb = n_loc // mb_loc
    def step(k):
        j = k % n_mb
        j2 = (k + 1) % n_mb            # the data streamer's next minibatch: its row terms are prepared while this step runs
        nxt = slice(j2 * mb_loc, (j2 + 1) * mb_loc) if (n_mb > 1 and not a.no_update and not a.no_prefetch) else None
        # lazy_elbo: the ELBO scalar is produced on the library's side stream; it is read only after the final synchronisation
        return m.svi_step(slice(j * mb_loc, (j + 1) * mb_loc), lr=a.lr, allreduce=allreduce,
                          qU_ratio=float(mb) / n_total / world if world > 1 else None, update=not a.no_update,
                          lazy_elbo=not a.no_update, next_seqs=nxt)

    for k in range(a.warmup):
        step(k)
    if not a.no_update:
        m.handle.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    m.handle.set_profiling(True)
    m.handle.get_profile(reset=True)
    t0 = time.perf_counter()
    for k in range(a.steps):
        e = step(a.warmup + k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    psi2_ms, psi2_n, psi2_nmmq = m.handle.get_profile(reset=True)
    if not a.no_update:
        m.handle.sync()
    elbo = float(e.item())
    tmax = torch.tensor([dt], dtype=torch.float64, device=m.device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        # ---- roofline of the dominant kernel (fused psi2 pass), DESIGN.md section 5 ----
        # algorithmic flops per (row, m, m') of the factorised form: Gram 2Q + y-update 2Q, + 2 for the accumulations;
        # the N*M*M exp evaluations are listed separately (no fp64 transcendental unit: ~20 FMA each on the VALU).
        flops = 4.0 * psi2_nmmq                      # 4 Q flop per (n, m, m')
        achieved = flops / (psi2_ms * 1e-3) / 1e12 if psi2_ms > 0 else 0.0
        out = {
            'metric': 'SVI steps/sec, 2-layer RGP M=100 T=4096 (ELBO rel-err vs CPU oracle: elbo_rel_err_vs_oracle)',
            'value': a.steps / dt, 'unit': 'SVI steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['name'], 'n_total': n_total, 'T': T, 'minibatch_sequences': mb,
                       'rows_per_layer_per_step': mb * (T - cfg['U_win']), 'parallelism': 'dp%d (sequences sharded; %d doubles of global gradients + ELBO all-reduced per step, one collective per layer)' % (world, m.n_global + 1)},
            'elbo_last_step': elbo,
            'roofline': {'bound': 'mfma', 'bound_note': 'fp64 compute (VALU); the fp64 matrix pipe is slower than the vector pipe on MI355X, see DESIGN.md', 'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': measured_traffic_bytes(),
                         'traffic_note': 'bytes per Q=40 launch from a separate rocprofv3 --pmc pass (profiles/r1_psi2_rowsN_traffic_raw.json); algorithmic ~0.25e9',
                         'kernel': 'psi2_rowsN_kernel (fused psi2 forward+backward, packed teams, fp64 VALU, 3 launches per step: Q=40,40,20)',
                         'avg_launch_ms': psi2_ms / max(psi2_n, 1), 'launches': psi2_n,
                         'algorithmic_flop_per_launch': flops / max(psi2_n, 1),
                         'note': 'flop = 4*N*M*M*Q per launch (Gram + W.z products of the factorised form); the N*M*M fp64 exp '
                                 'evaluations are extra VALU work not counted; SURVEY 8d counts 10*N*M*M*Q for the GPy form'},
        }
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, Us, Ys, m, mb, n_total)
            if 'elbo_rel_err' in out['cpu_baseline']:          # device vs oracle ELBO of one full-length sequence, same parameters
                out['elbo_rel_err_vs_oracle'] = out['cpu_baseline']['elbo_rel_err']
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(cfg, Us, Ys, m, mb, n_total):
    """The NumPy oracle (vectorised restatement, BLAS threads = all host cores) on a bounded sample of the same
    workload: ONE sequence of the minibatch through all three layers (ELBO + all gradients), scaled to the 32-sequence
    step by the additivity of the SVI bound over sequences (testing/minibatch_tests.py:288-296)."""
    from oracle import model as om
    T_s = min(len(Ys[0]), 4096)
    o = om.DeepAutoreg(cfg['wins'], [Ys[0][:T_s]], U=[Us[0][:T_s]], U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05,
                       num_inducing=cfg['M'], init='Y', inference_method='svi', n_total=n_total,
                       rng=np.random.default_rng(20241), fast_psi=True, chunk=256)
    # same parameters and the same sequence-0 latents as the device model (after the timed steps), so that the oracle's ELBO
    # can be compared with the device's on identical inputs at the full T
    for L in m.layers:
        p = o.P[L.index]
        for nm in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            getattr(p, nm)[...] = m.view(L, nm).cpu().numpy()
        p.variance = float(m.view(L, 'variance').item())
        p.noise = float(m.view(L, 'noise').item())
        p.qU_a = float(m.view(L, 'qU_a').item())
    same_T = T_s == len(Ys[0])
    if same_T:
        for i in range(1, m.nLayers):
            o.Xmean[i][0][...] = m.latent(i, 0).cpu().numpy()
            o.Xvar[i][0][...] = m.latent(i, 0, var=True).cpu().numpy()
    t0 = time.perf_counter()
    reps = 0
    while True:
        e_o, _ = o.elbo_and_grads()
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 3:
            break
    per_seq = (time.perf_counter() - t0) / reps
    cores = os.cpu_count()
    parity = {}
    if same_T:
        e_d = float(m.evaluate([0]).item())
        m.handle.sync()
        parity = {'elbo_oracle': float(e_o), 'elbo_device': e_d, 'elbo_rel_err': abs(e_d - float(e_o)) / abs(float(e_o))}
    return {**parity, 'value': 1.0 / (per_seq * mb), 'unit': 'SVI steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d evaluation(s) of ELBO+gradients of ONE T=%d sequence through all 3 GP layers (%.2f s each), '
                      'scaled x%d sequences per step; NumPy oracle, chunked GEMM psi-statistics, BLAS threads = %d'
                      % (reps, T_s, per_seq, mb, cores)}


if __name__ == '__main__':
    main()
