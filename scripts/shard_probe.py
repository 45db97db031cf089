"""Emulates one rank of an N-GPU run on a single GPU: scores the full config3
batch against 1/N of the alignments and prints the per-kernel times, for a
few scan grid sizes (GFAL_SCAN_GROUPS)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import shard, synth
from gfalign_amd.scorer import Scorer

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t = synth.make("config3")
if os.environ.get("RELABEL"):
    # node ids renumbered in truth-walk order: what a locality-preserving
    # node ordering at create time could achieve at best
    first = np.full(t.V, -1, np.int64)
    nodes = t.T >> 1
    seen = 0
    for v in nodes:
        if first[v] < 0:
            first[v] = seen; seen += 1
    rest = np.flatnonzero(first < 0)
    first[rest] = seen + np.arange(len(rest))
    remap = lambda a: ((first[a >> 1] << 1) | (a & 1)).astype(np.int32)
    t.aln_steps = remap(t.aln_steps); t.path_steps = remap(t.path_steps)
dev = torch.device("cuda", 0)
policy = os.environ.get('POLICY', 'items')
if policy == 'items':      # what bench.py / the CLI do: the scorer cuts its share after its own sort
    sc = Scorer(t.aln_off, t.aln_steps, t.V, shard=(0, world))
else:                      # the partitions of gfalign_amd/shard.py, for comparison
    off, st = shard.take_shard(t.aln_off, t.aln_steps, 0, world, policy)
    sc = Scorer(off, st, t.V)
P = t.P
if os.environ.get("SORT_PATHS"):
    lens = np.diff(t.path_off)
    order = np.argsort(-lens, kind="stable")
    paths = [t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in order]
    t.path_off = np.zeros(P + 1, np.int32); t.path_off[1:] = np.cumsum([len(x) for x in paths])
    t.path_steps = np.concatenate(paths).astype(np.int32)
d_off = torch.from_numpy(t.path_off).to(dev); d_st = torch.from_numpy(t.path_steps).to(dev)
d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
mx = int(np.diff(t.path_off).max())
for groups in sys.argv[2:] or ["8192"]:
    os.environ["GFAL_SCAN_GROUPS"] = groups
    for i in range(2):
        sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(); sc.set_profiling(True)
    for i in range(10):
        sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(); info = sc.info(); sc.set_profiling(False)
    print("1/%d shard, groups %s: wg %d  scan %.3f ms  sort+dp %.3f ms  call %.3f ms (ideal %.3f)" % (
        world, groups, info["n_workgroups"], info["scan_ms"], info["dp_ms"], info["total_ms"], 11.7 / world))
