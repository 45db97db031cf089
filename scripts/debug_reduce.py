"""Debug aid: shrink a failing kernel case to a minimal alignment subset."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "tests")
import oracle
from helpers import csr
from gfalign_amd.scorer import Scorer

cases = json.load(open("tests/golden/kernel_cases.json"))
c = [x for x in cases if x["name"] == "tiny-alphabet"][0]
aoff, ast = np.array(c["aln_off"]), np.array(c["aln_steps"], np.int32)
poff, pst = np.array(c["path_off"]), np.array(c["path_steps"], np.int32)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = pst[poff[k]:poff[k + 1]].tolist()
alns = [ast[aoff[a]:aoff[a + 1]].tolist() for a in range(len(aoff) - 1)]

def differs(sub):
    o, s = csr(sub)
    with Scorer(o, s, c["n_nodes"]) as sc:
        b, g, _ = sc.evaluate_paths([0, len(p)], p, False)
    eb, eg, _ = oracle.evaluate_paths(o, s, [0, len(p)], p, False)
    return (b[0], g[0]) != (eb[0], eg[0]), (b[0], g[0]), (eb[0], eg[0])

cur = list(alns)
assert differs(cur)[0]
i = 0
while i < len(cur):
    trial = cur[:i] + cur[i + 1:]
    if trial and differs(trial)[0]:
        cur = trial
    else:
        i += 1
print("path", p)
print("minimal alignments", cur, differs(cur))
