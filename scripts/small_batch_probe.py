"""Latency of small candidate batches (what the search driver submits) at a
BASELINE config: blocking gfal_scorer_score wall time and the device phases."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
sizes = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "8,32,128,512,2048".split(","))]
t = synth.make(cfg)
rng = np.random.default_rng(5)
with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
    prof = not os.environ.get("NOPROF")     # device phases need events; graphs are off then
    sc.set_profiling(prof)
    P = len(t.path_off) - 1
    for n in sizes:
        pick = np.sort(rng.choice(P, n, replace=False))
        off = [0]; steps = []
        for p in pick:
            steps.append(t.path_steps[t.path_off[p]:t.path_off[p + 1]])
            off.append(off[-1] + len(steps[-1]))
        off = np.asarray(off, np.int32); steps = np.concatenate(steps).astype(np.int32)
        for _ in range(3):
            sc.evaluate_paths(off, steps, True)
        sc.set_profiling(False); sc.set_profiling(prof)
        t0 = time.perf_counter()
        reps = 200 if not prof else 20
        for _ in range(reps):
            sc.evaluate_paths(off, steps, True)
        wall = (time.perf_counter() - t0) / reps * 1e3
        i = sc.info()
        print("P %5d  wall %.3f ms  device %.3f ms (scan %.3f, sort+dp %.3f, prep %.3f)  wg %d tile %d  ideal %.3f ms" % (
            n, wall, i["total_ms"], i["scan_ms"], i["dp_ms"], i["total_ms"] - i["scan_ms"] - i["dp_ms"],
            i["n_workgroups"], i["tile_paths"], 12.7 * n / 10000), flush=True)
