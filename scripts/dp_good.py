import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer
t = synth.make("config3")
with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
    b, g, u = sc.evaluate_paths(t.path_off, t.path_steps, True)
    print("good total", int(g.astype(np.int64).sum()), "bad total", int(b.astype(np.int64).sum()), "dp_pairs", sc.info()["dp_pairs"])
