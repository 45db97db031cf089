"""Timing probes for k_scan: same alignments (config3), different candidate
batches.  Prints scan/dp ms per variant."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

t = synth.make("config3")
dev = torch.device("cuda", 0)
sc = Scorer(t.aln_off, t.aln_steps, t.V)
P = t.P

def run(name, paths):
    off = np.zeros(len(paths) + 1, np.int32)
    off[1:] = np.cumsum([len(p) for p in paths])
    st = np.concatenate(paths).astype(np.int32)
    d_off = torch.from_numpy(off).to(dev); d_st = torch.from_numpy(st).to(dev)
    d_cnt = torch.zeros(3 * len(paths), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    mx = int(np.diff(off).max())
    for i in range(2):
        sc.score_device(d_off.data_ptr(), d_st.data_ptr(), len(paths), int(off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    sc.set_profiling(True)
    for i in range(5):
        sc.score_device(d_off.data_ptr(), d_st.data_ptr(), len(paths), int(off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    info = sc.info(); sc.set_profiling(False)
    c = d_cnt.cpu().numpy()
    print("%-28s scan %7.2f ms  dp %6.2f ms  tile %d  good/path %.0f bad/path %.0f dp_pairs %d" % (
        name, info["scan_ms"], info["dp_ms"], info["tile_paths"], c[len(paths):2*len(paths)].mean(), c[:len(paths)].mean(), info["dp_pairs"]))

T = t.T
run("bench batch", [t.path_steps[t.path_off[k]:t.path_off[k+1]] for k in range(P)])
run("full walk x P", [T] * P)
run("half walk x P", [T[:len(T)//2]] * P)
run("len-2 prefix x P", [T[:2]] * P)
run("len-100 prefix x P", [T[:100]] * P)
rc = (T[::-1] ^ 1).astype(np.int32)
run("rc full walk x P", [rc] * P)
