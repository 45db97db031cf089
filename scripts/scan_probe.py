"""Time one score_device call of a workload under several settings of the
scorer's debug / tuning environment variables (same process, fresh scorer per
setting).  usage: scan_probe.py <workload> "<VAR=val,VAR=val>" ["..."]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

t = synth.make(sys.argv[1])
dev = torch.device("cuda", 0)
P = t.P
d_off = torch.from_numpy(t.path_off.astype(np.int32)).to(dev)
d_steps = torch.from_numpy(t.path_steps.astype(np.int32)).to(dev)
d_counts = torch.zeros(3 * P, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
total = int(t.path_off[-1]); max_len = int(np.diff(t.path_off).max())
for setting in sys.argv[2:]:
    added = []
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        added.append(k)
    with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
        for _ in range(3):
            sc.score_device(d_off.data_ptr(), d_steps.data_ptr(), P, total, max_len, True, d_counts.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        sc.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(10):
            sc.score_device(d_off.data_ptr(), d_steps.data_ptr(), P, total, max_len, True, d_counts.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        info = sc.info()
        c = d_counts.cpu().numpy().view(np.uint32).astype(np.uint64)
        chk = int(c[:P].sum() * 3 + c[P:2 * P].sum() * 5 + c[2 * P:].sum() * 7)
        print("%-50s %8.3f ms/step  scan %7.3f  dp %6.3f  tile %d wg %d lds %d  checksum %d" % (
            setting or "(default)", dt * 1e3, info["scan_ms"], info["dp_ms"], info["tile_paths"],
            info["n_workgroups"], info["lds_bytes"], chk), flush=True)
    for k in added:
        del os.environ[k]
