#!/usr/bin/env python3
"""bench.py -- SVI steps/s of the recurrent-GP hot path on BASELINE.json's synthetic configurations, 1..8 MI355X, one process
per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3                 # config 4 (the configuration the metric is quoted on)
    python bench.py --config 5 --steps 3 --warmup 1                # config 5: T=65536, D_y=64, 3 hidden layers, M=500
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one SVI step: minibatch of sequences (sharded over the ranks) -> ELBO + every gradient of all GP layers ->
all-reduce of the global gradients (N > 1) -> Adam on the global parameters and on the touched sequences' latent parameters.
Inputs are resident in HBM before the timed region.  Scaling is STRONG: the minibatch is fixed and split over the ranks.
Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (the dominant psi2 kernel, timed with HIP
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


def measured_traffic_bytes(name):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs; FETCH_SIZE doubled as the gfx950 guide prescribes for wide coalesced reads).  None if absent."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', name)))
        return (2.0 * d['FETCH_SIZE']['avg_per_launch_kb'] + d['WRITE_SIZE']['avg_per_launch_kb']) * 1024.0
    except Exception:
        return None


def init_like_reference(m, Ys):
    """The canonical initialisation recipe (svi_experiments/rgp_experiments.py:632-645)."""
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
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', type=int, default=4, choices=[4, 5], help='BASELINE.json config (4: the metric; 5: M=500, T=65536)')
    ap.add_argument('--wide', action='store_true',
                    help='config 4 read with 16-dimensional hidden layers: lag windows of 340 / 640 / 320 columns (GEMM-shaped psi2)')
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-update', action='store_true', help='debug: skip the Adam update (kernel ablation runs); INVALID as a benchmark')
    ap.add_argument('--no-prefetch', action='store_true', help='debug: do not prepare the next minibatch during the running step')
    ap.add_argument('--n-total', type=int, default=None, help='sequences in the dataset (config 4: 256)')
    ap.add_argument('--T', type=int, default=None)
    ap.add_argument('--minibatch', type=int, default=None)
    ap.add_argument('--bucket', choices=['layer', 'single'], default='layer',
                    help='N > 1: all-reduce each layer\'s global gradients as soon as they exist (layer) or ONE bucket per step (single)')
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rgp_amd.synthetic import CONFIG4, CONFIG4W, CONFIG5, generate_sequences
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
        import faulthandler            # a stuck collective in this debugging mode ends the process with every thread's traceback
        faulthandler.dump_traceback_later(int(os.environ.get('RGP_BENCH_WATCHDOG_S', '900')), exit=True)
    torch.cuda.set_device(local)
    # RGP_BENCH_FORCE_DIST=1 (debugging aid): a single rank still creates the RCCL process group and passes its all-reduce to
    # the step, so the collective's stream handling can be exercised on a one-GPU box
    force_dist = world == 1 and os.environ.get('RGP_BENCH_FORCE_DIST') == '1'
    if force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local))
    if world > 1:
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    assert not (a.wide and a.config != 4)
    cfg = dict((CONFIG4W if a.wide else CONFIG4) if a.config == 4 else CONFIG5)
    n_total = a.n_total or cfg['n_total']
    T = a.T or cfg['T']
    mb = a.minibatch or cfg['minibatch']
    steps = a.steps if a.steps is not None else (20 if (a.config == 4 and not a.wide) else 3)
    warmup = a.warmup if a.warmup is not None else (3 if (a.config == 4 and not a.wide) else 1)
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
        if one_gpu:                    # gloo: CPU tensors only
            c = m.param_array[:m.n_global].cpu()
            dist.broadcast(c, src=0)
            m.param_array[:m.n_global].copy_(c)
        else:
            dist.broadcast(m.param_array[:m.n_global], src=0)
    allreduce = None
    if force_dist:
        allreduce = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if world > 1:
        if one_gpu:                    # gloo: CPU tensors only
            def allreduce(t):
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
                t.copy_(c)
        else:
            allreduce = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)
    m.single_bucket = (a.bucket == 'single')
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

    for k in range(warmup):
        step(k)
    if not a.no_update:
        m.handle.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    m.handle.set_profiling(True)
    m.handle.get_profile(reset=True)
    # per-step marks on the handle's main stream (HIP events; the only host work they add is one record per step)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    t0 = time.perf_counter()
    for k in range(steps):
        e = step(warmup + k)
        marks[k + 1].record()
    t_enq = time.perf_counter() - t0       # host time to enqueue all steps (the device is still running unless the host is the limit)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    psi2_ms, psi2_n, psi2_nmmq = m.handle.get_profile(reset=True)
    if not a.no_update:
        m.handle.sync()
    elbo = float(e.item())
    step_ms = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(steps)])
    tmax = torch.tensor([dt], dtype=torch.float64, device=m.device)
    if world > 1:
        if one_gpu:
            c = tmax.cpu(); dist.all_reduce(c, op=dist.ReduceOp.MAX); tmax.copy_(c)
        else:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        # ---- roofline of the dominant kernel ----
        # algorithmic flops per (row, m, m') of the factorised form: Gram 2Q + y-update 2Q; the N*M*M exp evaluations are
        # listed separately (no fp64 transcendental unit: 14 VALU instructions each).
        pair = False
        flops = 4.0 * psi2_nmmq                      # 4 Q flop per (n, m, m')
        if a.wide:                                   # executed count: three products over the M (M + 1) / 2 pairs, 2 N P Q flop each
            flops = 3.0 * psi2_nmmq * (cfg['M'] + 1.0) / cfg['M']
        achieved = flops / (psi2_ms * 1e-3) / 1e12 if psi2_ms > 0 else 0.0
        nl = len(m.layers)
        if a.wide:
            metric = ('SVI steps/sec, 2-layer RGP M=100 T=4096 with 16-dimensional hidden layers (windows of 340/640/320 columns; '
                      'ELBO rel-err vs CPU oracle: elbo_rel_err_vs_oracle)')
            kernel = ('gemm_f64_large_kernel (v_mfma_f64_16x16x4, 128 x 128 tiles) inside the GEMM-shaped psi2 term: Gram product, T and C '
                      'products over the N x P weights in HBM; the time is that of the whole psi2 term of a layer (three products + '
                      'exp pass + reductions), 3 terms per step')
            try:
                tw = json.load(open(os.path.join(ROOT, 'profiles', 'r2_wide_traffic.json')))
                traffic = (tw['total_fetch_raw'] + tw['total_write']) * 1e9
            except Exception:
                traffic = None
            tnote = ('bytes of ALL kernels of one layer step (the psi2 term is 3/4 of them), FETCH_SIZE as counted + WRITE_SIZE from separate '
                     'rocprofv3 --pmc passes (profiles/r2_wide_traffic.json; 8-byte-per-lane reads: FETCH_SIZE uncalibrated, at most 2x); '
                     'algorithmic: the 5.27 GB of N x P weights written once, rewritten once and read three times')
        elif a.config == 4:
            metric = 'SVI steps/sec, 2-layer RGP M=100 T=4096 (ELBO rel-err vs CPU oracle: elbo_rel_err_vs_oracle)'
            rows = mb_loc * (T - cfg['U_win'])
            pair = m.handle.psi2_pair_would_run(rows, cfg['M'], 2 * cfg['U_win'])
            if pair:
                kernel = ('fused psi2 pass in the pair form: psi2_pair_prep_kernel + psi2_pair_kernel (v_mfma_f64_16x16x4 over the '
                          'M(M+1)/2 unordered pairs of inducing rows: Gram, C and T products, exponentials between them) + '
                          'psi2_pair_rows_kernel + reduce + dz3; the time is that of the whole pass, 3 passes per step: Q=40,40,20')
                traffic = measured_traffic_bytes('r3_psi2_pair_traffic_raw.json')
                tnote = ('bytes per Q=40 pass (record, main and row kernels) from separate rocprofv3 --pmc passes '
                         '(profiles/r3_psi2_pair_traffic_raw.json); algorithmic ~0.27e9; the pass moves more: 1.46e9 of fp64 atomic adds '
                         'of the 10 pair groups into the N x 140 sums (resolved in L2/MALL, counted as writes), 0.17e9 of records, '
                         '0.15e9 of sums read back')
            else:
                kernel = 'psi2_rowsP_kernel (fused psi2 forward+backward, packed teams, the two waves of a SIMD half a step apart, fp64 VALU, 3 launches per step: Q=40,40,20)'
                traffic = measured_traffic_bytes('r2_psi2_rowsP_traffic_raw.json')
                tnote = 'bytes per Q=40 launch from separate rocprofv3 --pmc passes (profiles/r2_psi2_rowsP_traffic_raw.json); algorithmic ~0.27e9'
        else:
            metric = 'SVI steps/sec, 3-layer RGP M=500 T=65536 D=64 (BASELINE config 5; ELBO rel-err vs CPU oracle: elbo_rel_err_vs_oracle)'
            rows = mb_loc * (T - cfg['U_win'])
            pair = m.handle.psi2_pair_would_run(rows, cfg['M'], 2 * cfg['U_win'])
            if pair:
                kernel = ('psi2 term in the pair form, blocked: psi2_pairblk_prep_kernel + psi2_pairblk_kernel (v_mfma_f64_16x16x4 over the '
                          'unordered pairs of inducing rows in groups over 16 x 16 blocks; forward sums and backward in one pass) + '
                          'the products that finish the row gradients and dZ on the GEMM kernel; the time is that of the whole term, '
                          '4 per step: Q=40,40,40,20')
                traffic = measured_traffic_bytes('r3_cfg5_pairblk_traffic_raw.json')
                tnote = ('bytes per Q=40 term (record, main kernel and the kernels behind it) from separate rocprofv3 --pmc passes '
                         '(profiles/r3_cfg5_pairblk_traffic_raw.json); most of it is the fp64 atomic adds of the 257 pair groups into V2 and T '
                         'and their reads of the h blocks and d2 records')
            else:
                kernel = ('big_psi2_bwd_kernel (psi2 backward sweep over inducing blocks, fp64 VALU, 4 launches per step: Q=40,40,40,20); the '
                          'forward sums run in big_psi2_fwd_kernel (another 2*N*M*M*Q flop on 56 % of the pairs, not counted here)')
                traffic = measured_traffic_bytes('r2_cfg5_psi2_bwd_traffic_raw.json')
                tnote = 'bytes per Q=40 launch from a separate rocprofv3 --pmc pass (profiles/r2_cfg5_psi2_bwd_traffic_raw.json)'
        out = {
            'metric': metric,
            'value': steps / dt, 'unit': 'SVI steps/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['name'], 'n_total': n_total, 'T': T, 'minibatch_sequences': mb,
                       'rows_per_layer_per_step': mb * (T - cfg['U_win']),
                       'parallelism': 'dp%d (sequences sharded; %d doubles of global gradients + ELBO all-reduced per step, %s)'
                                      % (world, m.n_global + 1, 'ONE bucket' if (a.bucket == 'single' or nl > 3) else 'one collective per layer')},
            'elbo_last_step': elbo,
            'host_enqueue_ms_per_step': 1e3 * t_enq / steps,
            'step_ms_main_stream': {'median': float(np.median(step_ms)), 'p10': float(np.percentile(step_ms, 10)),
                                    'p90': float(np.percentile(step_ms, 90)),
                                    'note': 'HIP events on the main stream between consecutive steps (rank 0); value uses the wall clock over all steps'},
            'roofline': {'bound': 'mfma', 'bound_note': 'fp64 MFMA (v_mfma_f64_16x16x4): with windows in the hundreds of columns the contraction '
                                                        'is three large matrix products' if a.wide else
                                                        'fp64 MFMA (v_mfma_f64_16x16x4) for the three products of the pair form; the exponentials and '
                                                        'operand products between them are fp64 VALU on the same pipe (costs add, '
                                                        'profiles/r3_mfma_probes.txt)' if pair else
                                                        'fp64 VALU: the contract offers hbm|mfma; on MI355X the fp64 matrix pipe shares the vector '
                                                        'pipe\'s datapath and is slower (profiles/r1_f64_peak.txt), so the contraction runs as v_fma_f64',
                         'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': traffic, 'traffic_note': tnote,
                         'kernel': kernel,
                         'avg_launch_ms': psi2_ms / max(psi2_n, 1), 'launches': psi2_n,
                         'algorithmic_flop_per_launch': flops / max(psi2_n, 1),
                         'note': ('flop = 3*N*M*(M+1)*Q per psi2 term as executed: three N x P x Q products over the P = M(M+1)/2 pairs of '
                                  'inducing rows (the dense factorised form counts 4*N*M*M*Q); each product alone runs at about 50 TFLOP/s '
                                  '(profiles/r2_wide_kernel_stats.csv)') if a.wide else
                                 ('flop = 4*N*M*M*Q per pass, the count of the dense factorised form, kept so that the fraction compares with '
                                  'earlier rounds; the pair form executes 3*N*M*(M+1)*Q (1.60e11 per Q=40 pass, '
                                  'SQ_INSTS_VALU_MFMA_MOPS_F64 in profiles/r3_psi2_pair_counters.json) on the matrix pipe, which is busy 48 % '
                                  'of the main kernel\'s time') if (a.config == 4 and pair) else
                                 ('flop = 4*N*M*M*Q per psi2 term, the count of the dense factorised backward form, kept so that the fraction '
                                  'compares with earlier rounds; the pair form executes 3*N*M*(M+1)*Q plus 5 % of padded pairs on the matrix '
                                  'pipe and yields the forward sums as well (the sweep kernels spent another 2*N*M*M*Q on those)') if pair else
                                 ('flop = 4*N*M*M*Q per launch (Gram + W.z products of the factorised form); the N*M*M fp64 exp '
                                  'evaluations are extra VALU work not counted; the unfactorised GPy form would count 10*N*M*M*Q')},
        }
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, Us, Ys, m, mb, n_total, T)
            if 'elbo_rel_err' in out['cpu_baseline']:          # device vs oracle ELBO of one sequence, same parameters
                out['elbo_rel_err_vs_oracle'] = out['cpu_baseline']['elbo_rel_err']
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _oracle_for(cfg, U, Y, m, n_total, chunk):
    from oracle import model as om
    o = om.DeepAutoreg(cfg['wins'], [Y], U=[U], U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05,
                       num_inducing=cfg['M'], init='Y', inference_method='svi', n_total=n_total,
                       rng=np.random.default_rng(20241), fast_psi=True, chunk=chunk)
    for L in m.layers:
        p = o.P[L.index]
        for nm in ('Z', 'inv_l', 'qU_m', 'qU_W'):
            getattr(p, nm)[...] = m.view(L, nm).cpu().numpy()
        p.variance = float(m.view(L, 'variance').item())
        p.noise = float(m.view(L, 'noise').item())
        p.qU_a = float(m.view(L, 'qU_a').item())
    return o


def _time_oracle(o, budget_s, max_reps):
    t0 = time.perf_counter()
    reps = 0
    while True:
        e_o, _ = o.elbo_and_grads()
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= max_reps:
            break
    return (time.perf_counter() - t0) / reps, reps, e_o


def cpu_baseline(cfg, Us, Ys, m, mb, n_total, T):
    """The NumPy oracle on a bounded sample of the same workload, on this host's cores.
    (ii) vectorised restatement (chunked GEMM psi statistics): ELBO + all gradients of ONE sequence (the first T_s steps of
         sequence 0) through all GP layers, scaled to the step by the additivity of the SVI bound over rows
         (testing/minibatch_tests.py:288-296).  BLAS threads are swept (1, 8, 32, all) and the best is reported.
    (i)  reference-faithful structure: psi2n materialised (N x M x M) and the per-row Python scatter loop of
         autoreg/layers.py:1433-1450, on a shorter piece of the same sequence (`reference_structure`)."""
    from oracle import model as om
    import torch
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    big = cfg['M'] > 128 or max(cfg['nDims'][1:]) > 4        # the oracle's psi2 work per row grows with M * M * Q: a shorter sample
    T_s = min(len(Ys[0]), 4096 if not big else 276)
    rows_per_seq_full = len(Ys[0]) - cfg['U_win']
    rows_s = T_s - cfg['U_win']
    scale = float(mb) * rows_per_seq_full / rows_s           # sample evaluations per SVI step
    o = _oracle_for(cfg, Us[0][:T_s], Ys[0][:T_s], m, n_total, 256 if not big else 16)
    same_T = T_s == len(Ys[0])
    if same_T:
        for i in range(1, m.nLayers):
            o.Xmean[i][0][...] = m.latent(i, 0).cpu().numpy()
            o.Xvar[i][0][...] = m.latent(i, 0, var=True).cpu().numpy()
    cores = os.cpu_count()
    sweep = {}
    best = None
    counts = [c for c in (1, 8, 32) if c < cores] + [cores]
    for c in counts:
        if threadpool_limits is None and c != cores:
            continue
        if threadpool_limits is not None:
            with threadpool_limits(limits=c):
                per, reps, e_o = _time_oracle(o, 4.0, 2)
        else:
            per, reps, e_o = _time_oracle(o, 4.0, 2)
        sweep[str(c)] = per
        if best is None or per < best[0]:
            best = (per, c, reps)
    per_seq, best_threads, reps = best
    parity = {}
    if same_T:
        e_d = float(m.evaluate([0]).item())
        m.handle.sync()
        parity = {'elbo_oracle': float(e_o), 'elbo_device': e_d, 'elbo_rel_err': abs(e_d - float(e_o)) / abs(float(e_o))}
    else:
        # parity on the sample itself: a device model of the same truncated sequence with the same parameters
        from rgp_amd.model import DeepAutoreg
        g = DeepAutoreg(cfg['wins'], [Ys[0][:T_s]], U=[Us[0][:T_s]], U_win=cfg['U_win'], nDims=cfg['nDims'], X_variance=0.05,
                        num_inducing=cfg['M'], init='Y', inference_method='svi', mb_inf_tot_data_size=n_total, seed=20241)
        g.load_from_oracle(o)
        e_d = float(g.evaluate().item())
        g.handle.sync()
        parity = {'elbo_oracle': float(e_o), 'elbo_device': e_d, 'elbo_rel_err': abs(e_d - float(e_o)) / abs(float(e_o)),
                  'parity_sample': 'first %d steps of sequence 0' % T_s}
        del g
    # (i) reference-faithful structure on a short piece
    T_r = min(T_s, 148 if not big else 36)
    o_r = _oracle_for(cfg, Us[0][:T_r], Ys[0][:T_r], m, n_total, 10 ** 9)
    o_r.fast_psi = False                                                 # psi2n materialised as one (N, M, M) array per layer
    o_r.reference_scatter = True                                         # per-row Python loop (layers.py:1433-1450)
    per_r, reps_r, _ = _time_oracle(o_r, 6.0, 2)
    scale_r = float(mb) * rows_per_seq_full / (T_r - cfg['U_win'])
    return {**parity, 'value': 1.0 / (per_seq * scale), 'unit': 'SVI steps/s', 'cores': best_threads, 'kind': 'port',
            'host_cores': cores, 'blas_thread_sweep_s_per_sample': sweep,
            'sample': '%d evaluation(s) of ELBO+gradients of ONE T=%d piece of sequence 0 through all %d GP layers (%.2f s each at the best '
                      'of the BLAS thread counts tried), scaled x%.1f by rows per step; NumPy oracle, chunked GEMM psi-statistics'
                      % (reps, T_s, len(m.layers), per_seq, scale),
            'reference_structure': {'value': 1.0 / (per_r * scale_r), 'unit': 'SVI steps/s',
                                    'sample': 'psi2n materialised (N x M x M) + per-row Python scatter loop on a T=%d piece (%.2f s), scaled x%.1f'
                                              % (T_r, per_r, scale_r)}}


if __name__ == '__main__':
    main()
