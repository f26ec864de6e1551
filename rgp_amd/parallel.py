"""Data-parallel SVI over sequences: one process per GPU, sequences sharded, ONE all-reduce of
the global-gradient bucket per step.  Backend-agnostic helpers (torch.distributed: "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests); nothing here touches the numerical path.

Identity used (reference: autoreg/layers.py:76-79,328-329; asserted by testing/minibatch_tests.py:288-296): with
qU_ratio_local = n_local / n_total on every rank,
    sum_ranks [ELBO_local, d ELBO_local / d theta_global]  ==  the single-process value for the union minibatch.
Per-sequence (latent) parameters never leave the rank that owns the sequence."""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous shard [start, stop) of n sequences for this rank (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def local_qU_ratio(n_local_batch, n_total):
    return float(n_local_batch) / float(n_total)


def local_minibatch(step, n_local, mb_local):
    """The k-th contiguous minibatch of this rank's shard (StdMemoryDataStreamer order, data_streamers.py:237-267)."""
    n_mb = max(n_local // mb_local, 1)
    j = step % n_mb
    return slice(j * mb_local, min((j + 1) * mb_local, n_local))


def allreduce_bucket(bucket):
    """SUM all-reduce of the flat [global gradients | ELBO] bucket, in place.  No-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    return bucket


def broadcast_globals(param_globals, src=0):
    """Make the global parameters identical on every rank (q(U) initialisation depends on the local data)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(param_globals, src=src)
    return param_globals
