"""One scorer on the config-N batch for a kernel timeline (run under rocprofv3 --kernel-trace,
then scripts/ktimeline.py):  timeline_probe.py <config> <world> <dedup 0|1>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

t = synth.make(sys.argv[1] if len(sys.argv) > 1 else "config3")
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dedup = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda", 0)
P = t.P
d_off = torch.from_numpy(t.path_off).to(dev); d_st = torch.from_numpy(t.path_steps).to(dev)
d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
mx = int(np.diff(t.path_off).max())
with Scorer(t.aln_off, t.aln_steps, t.V, shard=(0, world), dedup=dedup) as sc:
    run = lambda: sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
    for _ in range(3): run()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    print("%s 1/%d dedup=%d: %.3f ms per step, dp_pairs %d, lanes %d" % (sys.argv[1], world, dedup, (time.perf_counter() - t0) / 20 * 1e3,
          sc.info()["dp_pairs"], sc.info().get("n_lanes", -1)), flush=True)
