"""Randomised parity run of search-mode scoring on the GPU box: random tangles, a
search tree grown for several generations through gfal_group_score_children (stored
parents and in-batch chains, slots reused, shards, weighted lanes, steps that revisit
the start of the path), every batch compared with the CPU oracle on the full paths.
usage: fuzz_children.py [n_cases] [seed0]"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from gfalign_amd.scorer import GFAL_STEP_OTHER, Group, Scorer
from helpers import csr

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
t0 = time.time()
scored = 0
for case in range(n_cases):
    rnd = random.Random(seed0 + case)
    V = rnd.choice([2, 3, 4, 8, 20, 60, 300])
    walk_len = rnd.randint(40, 400)
    max_m = rnd.choice([2, 3, 5, 8, 12, 20, 40])
    walk = [(rnd.randrange(V) << 1) | rnd.randrange(2) for _ in range(walk_len)]
    if rnd.random() < 0.5 and walk_len > 120:          # the walk comes back to where it started
        walk[walk_len // 2: walk_len // 2 + 20] = walk[0:20]
    alns = []
    for _ in range(rnd.randint(50, 4000)):
        m = 0 if rnd.random() < 0.01 else rnd.randint(1, max_m)
        s = rnd.randrange(0, walk_len - m + 1) if rnd.random() < 0.6 else rnd.randrange(0, min(30, walk_len - m + 1))
        b = list(walk[s:s + m])
        if b and rnd.random() < 0.2:
            b[rnd.randrange(len(b))] = (rnd.randrange(V) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        for _ in range(rnd.choice([1, 1, 1, 4])):
            alns.append(b)
    rnd.shuffle(alns)
    aoff, ast = csr(alns)
    longest = max(len(b) for b in alns)
    n_shards = rnd.choice([1, 1, 2, 3])
    dedup = rnd.random() < 0.3
    os.environ["GFAL_DP_SYS_LIMIT"] = rnd.choice(["0", "8192", "4000000000"])
    scs = [Scorer(aoff, ast, V + 1, shard=(k, n_shards), dedup=dedup) for k in range(n_shards)]
    try:
        with Group(scs) as g:
            cap = 256
            g.store_reserve(cap)
            free = list(range(cap - 1, -1, -1))
            start = rnd.randrange(0, 5)
            root = walk[start:start + longest + rnd.randrange(3)]
            if len(root) < max(longest, 1):
                continue
            if rnd.random() < 0.3:
                root[0] = GFAL_STEP_OTHER | (root[0] & ~1)
            slots = [free.pop()]
            off, st = csr([root])
            got = g.score_store(off, st, slots)
            exp = oracle.evaluate_paths(aoff, ast, off, st, True)
            stored = {slots[0]: root}
            ok = all(np.array_equal(a, b) for a, b in zip(got, exp))
            for gen in range(rnd.randint(2, 6)):
                if not ok:
                    break
                parent, step, slot, full = [], [], [], []
                parents = rnd.sample(sorted(stored), min(len(stored), rnd.randint(1, 6)))
                for ps in parents:
                    base = stored[ps]
                    for _ in range(rnd.randint(1, 4)):
                        r = rnd.random()
                        if r < 0.4 and len(base) + start < walk_len:
                            s_ = walk[(start + len(base)) % walk_len]
                        elif r < 0.6:
                            s_ = walk[rnd.randrange(0, min(12, walk_len))] ^ rnd.randrange(2)   # back to the start
                        else:
                            s_ = (rnd.randrange(V + 1) << 1) | rnd.randrange(2)
                        parent.append(ps); step.append(s_); full.append(base + [s_])
                        slot.append(free.pop() if len(free) > 8 and rnd.random() < 0.8 else -1)
                for _ in range(rnd.randint(0, 6)):         # chains inside the batch
                    j = rnd.randrange(len(full))
                    s_ = walk[rnd.randrange(walk_len)] if rnd.random() < 0.7 else (rnd.randrange(V + 1) << 1)
                    parent.append(~j); step.append(s_); full.append(full[j] + [s_])
                    slot.append(free.pop() if len(free) > 8 and rnd.random() < 0.5 else -1)
                if max(len(p) for p in full) > 990:
                    break
                got = g.score_children(parent, step, slot, max(len(p) for p in full))
                off, st = csr(full)
                exp = oracle.evaluate_paths(aoff, ast, off, st, True)
                ok = all(np.array_equal(a, b) for a, b in zip(got, exp))
                scored += len(full)
                for s_, p in zip(slot, full):
                    if s_ >= 0:
                        stored[s_] = p
                for ps in parents:                          # parents that will not be named again: slots reused
                    if rnd.random() < 0.5 and len(stored) > 2:
                        del stored[ps]
                        free.append(ps)
            if not ok:
                k = [int(np.flatnonzero(np.asarray(a) != np.asarray(b))[0]) for a, b in zip(got, exp) if not np.array_equal(a, b)]
                print("MISMATCH case %d (seed %d): V %d max_m %d shards %d dedup %s generation %d path %s" % (
                    case, seed0 + case, V, max_m, n_shards, dedup, gen, k))
                sys.exit(1)
    finally:
        for s in scs:
            s.close()
    if case % 20 == 19:
        print("%d cases, %d candidates, %.0f s" % (case + 1, scored, time.time() - t0), flush=True)
print("OK: %d cases, %d candidates scored from their parents, bit-exact against the oracle" % (n_cases, scored))
