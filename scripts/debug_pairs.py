"""Debug aid: find the (path, alignment) pairs where the HIP path and the
oracle disagree on a committed kernel case.  Run on a GPU box."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from gfalign_amd.scorer import Scorer

cases = json.load(open("tests/golden/kernel_cases.json"))
name = sys.argv[1] if len(sys.argv) > 1 else "tiny-alphabet"
c = [x for x in cases if x["name"] == name][0]
aoff, ast = np.array(c["aln_off"]), np.array(c["aln_steps"], np.int32)
poff, pst = np.array(c["path_off"]), np.array(c["path_steps"], np.int32)
shown = 0
for flt in (True, False):
    with Scorer(aoff, ast, c["n_nodes"]) as sc:
        bad, good, una = sc.evaluate_paths(poff, pst, flt)
        print("info", sc.info())
    eb, eg, eu = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
    for k in range(len(poff) - 1):
        if (bad[k], good[k]) == (eb[k], eg[k]):
            continue
        p = pst[poff[k]:poff[k + 1]]
        print("filter", flt, "path", k, p.tolist(), "gpu", (bad[k], good[k]), "oracle", (eb[k], eg[k]))
        for a in range(len(aoff) - 1):
            b = ast[aoff[a]:aoff[a + 1]]
            with Scorer([0, len(b)], b, c["n_nodes"]) as s1:
                gb, gg, _ = s1.evaluate_paths([0, len(p)], p, flt)
            ob, og, _ = oracle.evaluate_paths([0, len(b)], b, [0, len(p)], p, flt)
            if (gb[0], gg[0]) != (ob[0], og[0]):
                print("   aln", a, b.tolist(), "gpu(b,g)", (gb[0], gg[0]), "oracle", (ob[0], og[0]))
                shown += 1
        if shown > 12:
            sys.exit(0)
