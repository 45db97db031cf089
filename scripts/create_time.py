import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer
for cfg in sys.argv[1:]:
    t0 = time.time(); t = synth.make(cfg); t1 = time.time()
    sc = Scorer(t.aln_off, t.aln_steps, t.V); t2 = time.time()
    b, g, u = sc.evaluate_paths(t.path_off[:101], t.path_steps[:t.path_off[100]], True); t3 = time.time()
    print("%s: synth %.1f s, create %.2f s (N=%d, S=%d), first 100-path batch %.3f s, info %s" % (
        cfg, t1 - t0, t2 - t1, t.N, t.S, t3 - t2, {k: v for k, v in sc.info().items() if k in ("n_local_nodes", "max_aln_len", "tile_paths", "n_workgroups")}))
    sc.close()
