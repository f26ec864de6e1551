"""world_size-2 gloo test of the data-parallel path: shards + local qU_ratio + one SUM all-reduce of the global bucket
reproduce the single-process ELBO and global gradients (the identity of testing/minibatch_tests.py:288-296).  The
numerical engine in this CPU test is the oracle; on the GPU box the same helpers wrap the HIP path (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_oracle_model import make_data


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _build(Y, U, n_total):
    from oracle import model as om
    m = om.DeepAutoreg([0, 3, 3], Y, U=U, U_win=2, nDims=[3, 2, 3], X_variance=0.05, num_inducing=4, init='Y',
                       inference_method='svi', n_total=n_total, rng=np.random.default_rng(1), fast_psi=False)
    return m


def _globals(m, G):
    keys = [k for k in m.param_list() if len(k) == 2]
    return keys, np.concatenate([np.asarray(G[k], dtype=float).ravel() for k in keys])


def _worker(rank, world, port, ref_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rgp_amd import parallel
    n_total = 4
    U, Y = make_data(n_total, 12, seed=0)
    full = _build(Y, U, n_total)                               # identical globals on every rank (same rng)
    a, b = parallel.shard_range(n_total, world, rank)
    loc = _build(Y[a:b], U[a:b], n_total)                      # this rank's shard; ratio = n_local / n_total inside
    for i in loc.order:                                        # broadcast_globals equivalent for the oracle objects
        for name in ('Z', 'qU_m', 'qU_W'):
            getattr(loc.P[i], name)[...] = getattr(full.P[i], name)
    e, G = loc.elbo_and_grads()
    keys, g = _globals(loc, G)
    bucket = torch.tensor(np.concatenate([g, [e]]))
    parallel.allreduce_bucket(bucket)
    if rank == 0:
        np.save(ref_path, bucket.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_reproduces_full_batch(tmp_path):
    ref_path = str(tmp_path / 'bucket.npy')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ref_path), nprocs=2, join=True)
    got = np.load(ref_path)
    U, Y = make_data(4, 12, seed=0)
    full = _build(Y, U, 4)
    e, G = full.elbo_and_grads()
    keys, g = _globals(full, G)
    want = np.concatenate([g, [e]])
    assert np.isclose(got[-1], want[-1], rtol=1e-13, atol=0)
    np.testing.assert_allclose(got[:-1], want[:-1], rtol=1e-9, atol=1e-11)


def test_shard_helpers():
    from rgp_amd import parallel
    covered = []
    for r in range(3):
        a, b = parallel.shard_range(10, 3, r)
        covered += list(range(a, b))
    assert covered == list(range(10))
    assert parallel.local_minibatch(5, 8, 4) == slice(4, 8)
    assert parallel.local_qU_ratio(4, 256) == 4 / 256
