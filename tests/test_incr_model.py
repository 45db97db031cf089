"""The parent -> child identities of tests/incr_model.py against the direct rule
(CPU only; the HIP side of it is tests/test_gpu_children.py)."""
import random

import incr_model as im
import kernel_model as km


def _case(rnd, n_nodes, max_m, n_aln, walk_len):
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(walk_len)]
    alns = []
    for _ in range(n_aln):
        m = rnd.randint(0 if rnd.random() < 0.02 else 1, max_m)
        s = rnd.randrange(0, walk_len - m + 1)
        b = list(walk[s:s + m])
        if b and rnd.random() < 0.2:
            b[rnd.randrange(len(b))] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = km.rc(b)
        alns.append(b)
    return walk, alns


def test_child_counters_follow_from_the_parents():
    rnd = random.Random(11)
    checked = 0
    for it in range(60):
        n_nodes = rnd.choice([2, 3, 5, 12])
        max_m = rnd.choice([2, 4, 6])
        walk, alns = _case(rnd, n_nodes, max_m, 120, 30)
        idx = im.Index(alns)
        start = rnd.randrange(0, 8)
        A = walk[start:start + idx.max_m]
        Pass, G1, _, _ = im.direct(A, alns)
        for s in walk[start + idx.max_m:] + [(rnd.randrange(n_nodes + 1) << 1) | 1, A[-1] ^ 1, A[-1]]:
            Pass, G1 = im.child(A, s, Pass, G1, idx)
            A = A + [s]
            dP, dG1, dG2, cand = im.direct(A, alns)
            assert (Pass, G1) == (dP, dG1), (it, A, s)
            assert sorted(im.dp_candidates(A, idx)) == cand
            # and the split is the reference's counters (kernel_model.evaluate is
            # fuzzed against the oracle in test_kernel_model.py)
            bad, good, _ = km.evaluate([(A, [a >> 1 for a in A])], alns, True)[0]
            assert good == dG1 + dG2 and bad == dP - good
            checked += 1
    assert checked > 500


def test_longest_old_window():
    # path 1+ 2+ 3+ 1+ 2+ ; appending 3+ repeats the window (1+ 2+ 3+)
    A = [2, 4, 6, 2, 4]
    assert im.longest_old_window(A, 6) == 3
    # appending 1- : rc window (1+) occurs -> 1; (2+ 1-)' = (1+ 2-) does not
    assert im.longest_old_window(A, 3) == 1
    # appending 2- after ...2+ : rc(2+ 2-) = (2+ 2-) occurs only in the child itself
    assert im.longest_old_window(A, 5) == 1
    assert im.longest_old_window(A, 8) == 0


def test_dp_status_is_inherited_when_the_tail_avoids_the_alignment():
    rnd = random.Random(5)
    inherited = changed = 0
    for it in range(40):
        n_nodes = rnd.choice([3, 5, 8, 14])
        max_m = rnd.choice([2, 3, 5])
        walk, alns = _case(rnd, n_nodes, max_m, 80, 40)
        idx = im.Index(alns)
        A = walk[:idx.max_m + rnd.randrange(4)]
        for depth_run in range(6):
            R = list(A)
            for d in range(1, 4):          # chains of up to three steps above a stored ancestor
                A = A + [(rnd.randrange(n_nodes + 2) << 1) | rnd.randrange(2)]
                for B in alns:
                    if len(B) == 0:
                        continue
                    if im.inherits(A, d, B):
                        assert im.dp_status(A, B) == im.dp_status(R, B), (R, A, B)
                        inherited += 1
                    elif im.dp_status(A, B) != im.dp_status(R, B):
                        changed += 1
    assert inherited > 2000 and changed > 0      # (the test is not vacuous either way)
