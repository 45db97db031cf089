import os, random, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from gfalign_amd.scorer import Scorer
from helpers import csr, random_case
max_m, max_n = 48, 64
rnd = random.Random(100 + max_m)
alns, paths = random_case(rnd, rnd.randint(2, 3), 1200, 60, max_m, max_n, min_m=max(1, max_m // 2 - 2), min_n=max_m // 2)
aoff, ast = csr(alns); poff, pst = csr(paths)
exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
for lim in ("0", "8192", "4000000000"):
    os.environ["GFAL_DP_SYS_LIMIT"] = lim
    with Scorer(aoff, ast, 4) as sc:
        got = sc.evaluate_paths(poff, pst, True)
        print("limit", lim, "dp_pairs", sc.info()["dp_pairs"], [bool(np.array_equal(g, e)) for g, e in zip(got, exp)],
              "sum bad+good gpu", int(got[0].sum() + got[1].sum()), "oracle", int(exp[0].sum() + exp[1].sum()))
ms = np.diff(aoff)
print("m range", ms.min(), ms.max(), "classes", np.bincount(np.digitize(ms, [5, 9, 17, 33])))
os.environ["GFAL_DP_SYS_LIMIT"] = "0"
for lo_m, hi_m, hi_n in ((2, 4, 30), (5, 8, 30), (9, 16, 40), (17, 32, 60), (33, 48, 64), (65, 90, 120)):
    rnd = random.Random(7 + hi_m)
    alns, paths = random_case(rnd, 2, 1200, 60, hi_m, hi_n, min_m=lo_m, min_n=hi_m)
    aoff, ast = csr(alns); poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    with Scorer(aoff, ast, 4) as sc:
        got = sc.evaluate_paths(poff, pst, True)
        print("m", lo_m, hi_m, "dp_pairs", sc.info()["dp_pairs"], [bool(np.array_equal(g, e)) for g, e in zip(got, exp)],
              "bad gpu/oracle", int(got[0].sum()), int(exp[0].sum()), "good", int(got[1].sum()), int(exp[1].sum()))
