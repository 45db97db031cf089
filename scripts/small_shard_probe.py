"""The 128-path blocking call on every shard of an N-way split (the search-sized batch of
scripts/shard_curve.py): wall time, phases, launch shape.  usage: small_shard_probe.py <config> <N>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

t = synth.make(sys.argv[1] if len(sys.argv) > 1 else "config3")
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
P = t.P
rng = np.random.default_rng(5)
pick = np.sort(rng.choice(P, 128, replace=False))
soff = [0]; sst = []
for p in pick:
    sst.append(t.path_steps[t.path_off[p]:t.path_off[p + 1]]); soff.append(soff[-1] + len(sst[-1]))
soff = np.asarray(soff, np.int32); sst = np.concatenate(sst).astype(np.int32)
if os.environ.get("NOGC"):       # (is the one-off 40 ms stall inside the timed loop the interpreter's collector?)
    import gc
    gc.disable()
BIG = bool(os.environ.get("BIG"))
if BIG:       # (torch before the scorer library touches HIP)
    import torch
    dev = torch.device("cuda", 0)
    d_off = torch.from_numpy(t.path_off).to(dev); d_st = torch.from_numpy(t.path_steps).to(dev)
    d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    mx = int(np.diff(t.path_off).max())
for k in range(world):
    if only >= 0 and k != only:
        continue
    with Scorer(t.aln_off, t.aln_steps, t.V, shard=(k, world)) as sc:
        if BIG:       # as scripts/shard_curve.py: the 10 000-path device-resident calls first
            for _ in range(13):
                sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            print("  after the big calls: wl_capacity %d dp_pairs %d" % (sc.info()["wl_capacity"], sc.info()["dp_pairs"]))
        for _ in range(5):
            sc.evaluate_paths(soff, sst, True)
        sc.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(50):
            sc.evaluate_paths(soff, sst, True)
        small = (time.perf_counter() - t0) / 50 * 1e3
        i = sc.info()
        print("shard %d/%d: %.3f ms per call; scan %.3f dp %.3f total %.3f ms; wg %d tile %d lds %d; lanes %d dp_pairs %d reruns %d passes %d" % (
            k, world, small, i["scan_ms"], i["dp_ms"], i["total_ms"], i["n_workgroups"], i.get("tile_paths", -1), i.get("lds_bytes", -1),
            i.get("n_lanes", -1), i["dp_pairs"], i["n_overflow_reruns"], i["n_device_passes"]), flush=True)
