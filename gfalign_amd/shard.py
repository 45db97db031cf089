"""Alignment sharding for the multi-GPU path.

evaluatePath's counters are sums over alignments of independent per-pair
decisions (reference src/eval.cpp:80-106), so the alignment set is cut into
contiguous ranges with equal step counts, one per rank; every rank scores the
whole candidate batch against its range and the per-path counters are summed
with ONE integer all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU
tests).  Integer sums are order-independent, so the result is bit-exact for any
number of ranks.
"""
import numpy as np


def shard_bounds(aln_off, world_size):
    """Cut points (len world_size+1) balancing the number of steps per rank."""
    aln_off = np.asarray(aln_off, dtype=np.int64)
    n_aln = len(aln_off) - 1
    total = int(aln_off[-1])
    targets = (np.arange(1, world_size, dtype=np.int64) * total) // world_size
    cuts = np.searchsorted(aln_off, targets, side="left")
    bounds = np.concatenate([[0], np.minimum(cuts, n_aln), [n_aln]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def take_shard(aln_off, aln_steps, rank, world_size):
    """CSR arrays of this rank's alignments (offsets re-based to 0)."""
    b = shard_bounds(aln_off, world_size)
    lo, hi = int(b[rank]), int(b[rank + 1])
    aln_off = np.asarray(aln_off)
    off = (aln_off[lo:hi + 1] - aln_off[lo]).astype(np.int32)
    steps = np.asarray(aln_steps)[aln_off[lo]:aln_off[hi]].astype(np.int32)
    return off, steps


def all_reduce_counts(counts, group=None):
    """Sum the [3P] per-path counters over the ranks (in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts
