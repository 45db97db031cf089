"""Randomised parity run on the GPU box: many small cases of varied shape
against the CPU oracle, with the DP kernel family forced either way and the
alignment set sharded at random.  usage: fuzz_gpu.py [n_cases] [seed0]"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from gfalign_amd.scorer import GFAL_STEP_OTHER, Scorer
from helpers import csr, random_case, walk_case

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
pairs = 0
for case in range(n_cases):
    rnd = random.Random(seed0 + case)
    kind = rnd.choice(["tiny", "walk", "walk", "long"])
    if kind == "tiny":
        V = rnd.randint(1, 4)
        alns, paths = random_case(rnd, V, rnd.randint(1, 1500), rnd.randint(1, 70),
                                  rnd.randint(1, 40), rnd.randint(1, 90))
        n_nodes = V + rnd.randint(0, 3)
    elif kind == "walk":
        V = rnd.randint(5, 400)
        alns, paths = walk_case(rnd, V, rnd.randint(20, 1000), rnd.randint(1, 3000),
                                rnd.randint(1, 200), rnd.randint(1, 24))
        n_nodes = V
    else:
        V = rnd.randint(3, 60)
        alns, paths = walk_case(rnd, V, rnd.randint(100, 1000), rnd.randint(1, 400),
                                rnd.randint(1, 30), rnd.randint(30, 200))
        n_nodes = V
    if rnd.random() < 0.2:
        alns[rnd.randrange(len(alns))] = []
    if rnd.random() < 0.3:                      # path steps whose orientation equals nothing
        for p in paths[:max(1, len(paths) // 4)]:
            p[rnd.randrange(len(p))] |= GFAL_STEP_OTHER
    universe = None
    if rnd.random() < 0.3:                      # nodes the paths may visit: theirs plus a few
        on_paths = {(s & ~GFAL_STEP_OTHER) >> 1 for p in paths for s in p}
        universe = sorted(on_paths | {rnd.randrange(n_nodes) for _ in range(3)})
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    flt = rnd.random() < 0.7
    os.environ["GFAL_DP_SYS_LIMIT"] = rnd.choice(["0", "8192", "4000000000"])
    n_shards = rnd.choice([1, 1, 2, 3, 5])
    dedup = rnd.random() < 0.4
    if dedup and rnd.random() < 0.7:            # make duplicates worth collapsing
        alns = [list(b) for b in alns for _ in range(rnd.choice([1, 1, 3, 9]))]
        rnd.shuffle(alns)
        aoff, ast = csr(alns)
    acc = [np.zeros(len(paths), np.uint64) for _ in range(3)]
    for k in range(n_shards):
        with Scorer(aoff, ast, n_nodes, universe=universe, shard=(k, n_shards), dedup=dedup) as sc:
            for a, part in zip(acc, sc.evaluate_paths(poff, pst, flt)):
                a += part
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
    for name, a, e in zip(("bad", "good", "unaligned"), acc, exp):
        if not np.array_equal(a, e.astype(np.uint64)):
            k = int(np.flatnonzero(a != e.astype(np.uint64))[0])
            print("MISMATCH case %d (%s, seed %d) %s path %d: gpu %d oracle %d; shards %d limit %s filter %s" % (
                case, kind, seed0 + case, name, k, a[k], e[k], n_shards, os.environ["GFAL_DP_SYS_LIMIT"], flt))
            sys.exit(1)
    pairs += len(alns) * len(paths)
    if case % 25 == 24:
        print("%d cases, %.1f M pairs, %.0f s" % (case + 1, pairs / 1e6, time.time() - t0), flush=True)
print("OK: %d cases, %.1f M (alignment, path) pairs bit-exact" % (n_cases, pairs / 1e6))
