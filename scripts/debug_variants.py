import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "tests")
import oracle
from helpers import csr
from gfalign_amd.scorer import Scorer
p = [0, 2, 5]
for sub in ([[1,3,0]], [[1,2,4]], [[1,3,0],[1,2,4]], [[1,3,0],[1,3,0]], [[1,3,0],[4,4,4]], [[1,3,0],[0,2,5]],
            [[1,3,0],[1,3,1]], [[1,3,0],[1,3,2]], [[1,2,4],[1,2,4]]):
    o, s = csr(sub)
    with Scorer(o, s, 8) as sc:
        b, g, _ = sc.evaluate_paths([0, len(p)], p, False)
        inf = sc.info()
        fw, rc = sc.pair_scores(p)
    eb, eg, _ = oracle.evaluate_paths(o, s, [0, len(p)], p, False)
    efw, erc = oracle.pair_scores(o, s, p)
    print(sub, "gpu", (int(b[0]), int(g[0])), "oracle", (int(eb[0]), int(eg[0])), "dp_pairs", inf["dp_pairs"],
          "pairs fw/rc", fw.tolist(), rc.tolist(), "exp", efw.tolist(), erc.tolist())
