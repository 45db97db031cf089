"""Alignment sharding for the multi-GPU path.

evaluatePath's counters are sums over alignments of independent per-pair
decisions (reference src/eval.cpp:80-106): every rank scores the whole candidate
batch against its shard of the alignments and the per-path counters are summed
with ONE integer all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU
tests).  Integer sums are order-independent, so the result is bit-exact for any
number of ranks and any partition.

The partition the PRODUCT uses (bench.py, the CLI) is made inside the scorer:
every rank hands gfal_scorer_create_sharded the whole set, the library sorts it
the way the kernels want it (length buckets, content order) and keeps its share
of the groups of 64: the alignment lengths are laid on a line in item order (a
fixed stretch for holding a length at all, then its groups weighted by what one
costs), rank k owns the k-th n-th of the line, and group i of a length falls on
its stretch at frac(i * golden ratio) -- so a rank holds FEW lengths (the scan
pays a fixed cost per length it holds) and its groups are spread over the whole
content order.  Policy "product" here asks the library
for exactly that assignment (gfal_shard_owner, host code, no GPU needed).
"range" (contiguous ranges balanced by step count) and "content" (copies of one
alignment kept together) are earlier policies kept for comparison: correct, but
slower on the GPU.
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


def content_owner(aln_off, aln_steps, world_size):
    """Rank that owns each alignment under the "content" policy: a hash of the
    alignment's steps, so all copies of one alignment live on one rank.

    The scan kernel works on 64 content-sorted alignments at a time and is
    fastest when they are (near-)identical; cutting the input into contiguous
    ranges spreads the copies of every alignment over all ranks (8x fewer
    copies per rank at 8 GPUs: +35 % scan work per alignment, measured), a
    content hash keeps them together and still balances the ranks.
    """
    aln_off = np.asarray(aln_off, dtype=np.int64)
    steps = np.asarray(aln_steps).astype(np.uint64)
    m = np.diff(aln_off)
    pos = np.arange(len(steps), dtype=np.uint64) - np.repeat(aln_off[:-1], m).astype(np.uint64)
    with np.errstate(over="ignore"):
        mixed = (steps + np.uint64(1)) * (np.uint64(0x9E3779B97F4A7C15) + pos * np.uint64(0xBF58476D1CE4E5B9))
        mixed ^= mixed >> np.uint64(29)
        h = np.zeros(len(m), np.uint64)
        nz = m > 0
        h[nz] = np.add.reduceat(mixed, aln_off[:-1][nz])
        h = (h ^ (h >> np.uint64(31))) * np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(32)
    # groups of identical alignments, heaviest first, dealt to the ranks in
    # serpentine order: step counts end up within a fraction of a percent
    _, inverse, = np.unique(h, return_inverse=True)[:2]
    weight = np.bincount(inverse, weights=m.astype(np.float64))
    by_weight = np.argsort(-weight, kind="stable")
    turn = np.arange(len(by_weight)) % (2 * world_size)
    rank_of_turn = np.where(turn < world_size, turn, 2 * world_size - 1 - turn)
    group_owner = np.empty(len(by_weight), np.int64)
    group_owner[by_weight] = rank_of_turn
    return group_owner[inverse]


def product_owner(aln_off, aln_steps, n_nodes, world_size):
    """Rank of every alignment under the product's own partition (the library's
    gfal_shard_owner: what gfal_scorer_create_sharded keeps on each rank)."""
    from . import scorer
    return scorer.shard_owner(aln_off, aln_steps, n_nodes, world_size)


def take_shard(aln_off, aln_steps, rank, world_size, policy="content", n_nodes=None):
    """CSR arrays of this rank's alignments (offsets re-based to 0).

    policy "product": what gfal_scorer_create_sharded keeps on this rank (needs
    n_nodes); "content": alignments are dealt to ranks by a hash of their steps
    (see content_owner); "range": contiguous ranges balanced by step count
    (shard_bounds).  Any partition gives the same summed counters.
    """
    aln_off = np.asarray(aln_off)
    aln_steps = np.asarray(aln_steps)
    if world_size == 1:
        return aln_off.astype(np.int32), aln_steps.astype(np.int32)
    if policy == "product":
        if n_nodes is None:
            n_nodes = int(aln_steps.max() >> 1) + 1 if len(aln_steps) else 1
        owner = product_owner(aln_off, aln_steps, n_nodes, world_size)
    elif policy == "range":
        b = shard_bounds(aln_off, world_size)
        lo, hi = int(b[rank]), int(b[rank + 1])
        off = (aln_off[lo:hi + 1] - aln_off[lo]).astype(np.int32)
        steps = aln_steps[aln_off[lo]:aln_off[hi]].astype(np.int32)
        return off, steps
    elif policy == "content":
        owner = content_owner(aln_off, aln_steps, world_size)
    else:
        raise ValueError("unknown sharding policy %r" % policy)
    mine = np.flatnonzero(owner == rank)
    m = (aln_off[mine + 1] - aln_off[mine]).astype(np.int64)
    off = np.zeros(len(mine) + 1, np.int64)
    np.cumsum(m, out=off[1:])
    idx = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], m) + np.repeat(aln_off[mine].astype(np.int64), m)
    return off.astype(np.int32), aln_steps[idx].astype(np.int32)


def all_reduce_counts(counts, group=None):
    """Sum the [3P] per-path counters over the ranks (in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts
