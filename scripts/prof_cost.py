"""What gfal_scorer_set_profiling costs a step (HIP events between the phases of a call).
usage: prof_cost.py [workload]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer
t = synth.make(sys.argv[1] if len(sys.argv) > 1 else "config3")
dev = torch.device("cuda", 0); P = t.P
d_off = torch.from_numpy(t.path_off.astype(np.int32)).to(dev); d_st = torch.from_numpy(t.path_steps.astype(np.int32)).to(dev)
d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev); stream = torch.cuda.current_stream(dev)
total = int(t.path_off[-1]); mx = int(np.diff(t.path_off).max())
with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
    for prof in (False, True, False, True):
        sc.set_profiling(prof)
        for _ in range(5):
            sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, total, mx, True, d_cnt.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, total, mx, True, d_cnt.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        print("profiling %-5s  %.3f ms per step" % (prof, (time.perf_counter() - t0) / 200 * 1e3), flush=True)
