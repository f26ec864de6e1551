"""Data-parallel SVI on the HIP path with world_size 2 (both ranks share the one GPU of the test box, gloo moves the device
buckets): sequence shards + local qU_ratio + per-layer all-reduce inside the pipelined svi_step reproduce the global
parameters of a single-process run on the union minibatch, step after step (bench.py's multi-GPU path with RCCL does the same
calls)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from test_gpu_model import make_data

WINS, NDIMS, UWIN, M, NTOT, T, STEPS = [0, 3, 3], [3, 2, 3], 2, 4, 4, 14, 3


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _model(Y, U):
    from rgp_amd.model import DeepAutoreg
    return DeepAutoreg(WINS, Y, U=U, U_win=UWIN, nDims=NDIMS, X_variance=0.05, num_inducing=M, init='Y',
                       inference_method='svi', mb_inf_tot_data_size=NTOT, seed=3)


def _globals0():
    """Identical starting globals for every process (q(U) initialisation depends on the local data otherwise)."""
    U, Y = make_data(NTOT, T, seed=0)
    return _model(Y, U).param_array[:_model(Y, U).n_global].clone()


def _worker(rank, world, port, out_path, lazy, single_bucket=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rgp_amd import parallel
    U, Y = make_data(NTOT, T, seed=0)
    full = _model(Y, U)
    a, b = parallel.shard_range(NTOT, world, rank)
    m = _model(Y[a:b], U[a:b])
    m.param_array[:m.n_global].copy_(full.param_array[:full.n_global])
    m.params_changed()
    def allreduce(t):              # gloo moves host tensors only: stage the device bucket through the host
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        t.copy_(c)
    m.single_bucket = single_bucket
    elbos = []
    for k in range(STEPS):
        e = m.svi_step(slice(0, b - a), lr=1e-2, allreduce=allreduce, qU_ratio=float(NTOT) / NTOT / world, lazy_elbo=lazy)
        elbos.append(e if lazy else float(e.item()))
    m.handle.sync()
    if lazy:                       # as in bench.py: the ELBO scalars are read only after the final synchronisation
        torch.cuda.synchronize()
        elbos = [float(e.item()) for e in elbos]
    if rank == 0:
        np.savez(out_path, g=m.param_array[:m.n_global].cpu().numpy(), e=np.array(elbos))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('lazy,single_bucket', [(False, False), (True, False), (False, True)])
def test_two_ranks_track_the_single_process_run(tmp_path, lazy, single_bucket):
    """Per-layer collectives inside the pipelined step (eager and lazy ELBO) and ONE bucket per step give the same
    trajectory as the single-process run on the union minibatch."""
    out = str(tmp_path / 'dp.npz')
    mp.spawn(_worker, args=(2, _free_port(), out, lazy, single_bucket), nprocs=2, join=True)
    got = np.load(out)
    U, Y = make_data(NTOT, T, seed=0)
    m = _model(Y, U)
    elbos = []
    for k in range(STEPS):
        elbos.append(float(m.svi_step(slice(0, NTOT), lr=1e-2).item()))
    m.handle.sync()
    np.testing.assert_allclose(got['e'], np.array(elbos), rtol=1e-10)
    np.testing.assert_allclose(got['g'], m.param_array[:m.n_global].cpu().numpy(), rtol=1e-8, atol=1e-10)


def test_side_streams_keep_their_own_hardware_queues_after_other_streams_exist():
    """A process that created many streams before the handle (an RCCL communicator makes six) must still get side streams that
    run next to the main stream: rgp_create picks them by test, not by creation order."""
    import torch
    from rgp_amd.capi import Handle
    hold = [torch.cuda.Stream() for _ in range(6)]
    for s in hold:
        with torch.cuda.stream(s):
            torch.zeros(8, device='cuda')
    torch.cuda.synchronize()
    h = Handle()
    assert h.side_streams_independent() == 3
