"""Every shard of an N-way split of a workload's alignments on one GPU, one after the
other: step, scan and sort+DP time of the 10 000-path batch (the slowest is the N-GPU step).
usage: shard_each.py <config> <N>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

t = synth.make(sys.argv[1] if len(sys.argv) > 1 else "config3")
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
P = t.P
d_off = torch.from_numpy(t.path_off).to(dev); d_st = torch.from_numpy(t.path_steps).to(dev)
d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
mx = int(np.diff(t.path_off).max())
for k in range(world):
    with Scorer(t.aln_off, t.aln_steps, t.V, shard=(k, world)) as sc:
        for _ in range(3):
            sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        sc.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(10):
            sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        i = sc.info()
        print("shard %d/%d: %7d alignments %8d steps  step %.3f ms  scan %.3f  sort+dp %.3f  wg %d  dp pairs %d" % (
            k, world, i["n_aln"], i["n_steps"], dt, i["scan_ms"], i["dp_ms"], i["n_workgroups"], i["dp_pairs"]), flush=True)
