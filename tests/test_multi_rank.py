"""CPU, world_size 2 over gloo: the multi-GPU contract of SURVEY.md 8(e).

Each rank takes its shard of the alignments -- the partition the PRODUCT makes
(gfal_scorer_create_sharded keeps its share of the groups of 64 of its own sorted
order: few alignment lengths per rank, the groups of a length spread over the
whole content order; gfal_shard_owner reports that assignment from host code, so
it can be checked here) -- produces the per-path counters for the whole candidate batch,
and the [3P] integer counters are summed with one all-reduce.  On the GPU box
the per-rank scorer is the HIP path and the backend is RCCL
(tests/test_gpu_multi_rank.py runs bench.py's N-rank path there); here the
oracle stands in for the scorer so that the partition and the collective can be
checked without a GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from gfalign_amd import shard, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = synth.make("smoke")
    off, st = shard.take_shard(t.aln_off, t.aln_steps, rank, world, policy="product", n_nodes=t.V)
    bad, good, una = oracle.evaluate_paths(off, st, t.path_off, t.path_steps, True)
    counts = torch.from_numpy(np.concatenate([bad, good, una]).astype(np.int32))
    shard.all_reduce_counts(counts)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), counts.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_counters_equal_unsharded(tmp_path, world):
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    t = synth.make("smoke")
    exp = np.concatenate(oracle.evaluate_paths(t.aln_off, t.aln_steps, t.path_off,
                                               t.path_steps, True)).astype(np.int32)
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert np.array_equal(got, exp), r


def test_shard_bounds_cover_everything_once():
    t = synth.make("config2")
    for world in (1, 2, 3, 8):
        b = shard.shard_bounds(t.aln_off, world)
        assert b[0] == 0 and b[-1] == t.N and np.all(np.diff(b) >= 0)
        sizes = [int(t.aln_off[b[r + 1]] - t.aln_off[b[r]]) for r in range(world)]
        assert sum(sizes) == t.S
        assert max(sizes) - min(sizes) <= 64          # balanced by step count
        for policy in ("range", "content"):
            total, steps = 0, 0
            for r in range(world):
                off, st = shard.take_shard(t.aln_off, t.aln_steps, r, world, policy)
                assert off[0] == 0 and off[-1] == len(st)
                total += len(off) - 1
                steps += len(st)
                if world > 1:
                    assert abs(len(st) - t.S / world) < 0.05 * t.S / world   # balanced
            assert total == t.N and steps == t.S


def test_product_partition_is_what_the_library_reports():
    """gfal_shard_owner (host half of gfal_scorer_create_sharded): every
    alignment has exactly one owner, the shards balance, zero-step alignments go
    to shard 0, and the same input gives the same partition every time (the
    ranks never communicate about it)."""
    t = synth.make("config2")
    for world in (2, 3, 8):
        owner = shard.product_owner(t.aln_off, t.aln_steps, t.V, world)
        assert owner.min() >= 0 and owner.max() == world - 1
        steps = np.bincount(owner, weights=np.diff(t.aln_off).astype(np.float64), minlength=world)
        # A shard holds few alignment lengths (each costs it a workgroup per tile whatever it
        # holds of it) and the groups of a length are spread over its shards: what balances
        # is groups weighted by what one costs + a small charge per length, not the step
        # count -- within a third at 2 and 3 shards of this small set.
        assert steps.sum() == t.S
        if world <= 3:
            assert steps.max() - steps.min() < 0.33 * t.S / world
        assert np.array_equal(owner, shard.product_owner(t.aln_off, t.aln_steps, t.V, world))
    off = np.array([0, 0, 2, 2, 5], np.int32)         # two zero-step alignments
    st = np.array([0, 2, 4, 2, 0], np.int32)
    owner = shard.product_owner(off, st, 4, 3)
    assert owner[0] == 0 and owner[2] == 0 and set(owner.tolist()) <= {0, 1, 2}


def test_content_policy_keeps_copies_together():
    t = synth.make("config2")
    owner = shard.content_owner(t.aln_off, t.aln_steps, 8)
    seen = {}
    for k in range(0, t.N, 7):
        key = t.aln_steps[t.aln_off[k]:t.aln_off[k + 1]].tobytes()
        assert seen.setdefault(key, owner[k]) == owner[k]


def test_more_ranks_than_alignments():
    off = np.array([0, 2, 5], np.int32)
    st = np.arange(5, dtype=np.int32)
    for policy in ("range", "content", "product"):
        seen = 0
        for r in range(4):
            o, s = shard.take_shard(off, st, r, 4, policy)
            seen += len(o) - 1
        assert seen == 2
