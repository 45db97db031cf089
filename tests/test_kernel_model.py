"""CPU: the decision rule the kernels implement, modelled in Python
(tests/kernel_model.py), fuzzed against the oracle."""
import random

import numpy as np

import kernel_model as km
import oracle
from helpers import csr, random_case, walk_case


def _compare(alns, paths, nodes_of=None):
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    for flt in (True, False):
        bad, good, una = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
        model = km.evaluate([(p, [s >> 1 for s in p]) for p in paths], alns, flt)
        for k, (mb, mg, mu) in enumerate(model):
            assert (mb, mg) == (bad[k], good[k]), (flt, paths[k])
            if flt:
                assert mu == una[k]


def test_model_tiny_alphabets():
    rnd = random.Random(3)
    for _ in range(40):
        alns, paths = random_case(rnd, rnd.randint(1, 4), 40, 16, 7, 9)
        _compare(alns, paths)


def test_model_walks():
    rnd = random.Random(4)
    for _ in range(10):
        alns, paths = walk_case(rnd, rnd.randint(3, 15), 30, 50, 12, 8)
        _compare(alns, paths)


def test_exit_propagation_equals_full_traceback():
    rnd = random.Random(5)
    for _ in range(30):
        alns, paths = random_case(rnd, rnd.randint(1, 4), 25, 4, 12, 12)
        aoff, ast = csr(alns)
        for p in paths:
            fw, rc = oracle.pair_scores(aoff, ast, p)
            for k, b in enumerate(alns):
                assert km.traceback_score(p, b) == fw[k]
                assert km.traceback_score(p, km.rc(b)) == rc[k]


def test_row_skipping_equals_the_full_fill():
    """The exact-DP kernels skip table rows whose path step has no node in the
    alignment, outside a window after each row that has one (scorer.hip
    traceback_score_skip).  Same score as the full fill, for every window size
    from m up, on dense (2-node) and sparse (100-node) alphabets; on the sparse
    ones most rows are skipped."""
    rnd = random.Random(77)
    total = computed = 0
    for it in range(6000):
        V = rnd.choice([2, 3, 5, 20, 100])
        n = rnd.randint(1, 70)
        m = rnd.randint(1, min(n, 12))
        A = [(rnd.randrange(V) << 1) | rnd.randrange(2) for _ in range(n)]
        if rnd.random() < 0.6:          # a piece of the path, possibly damaged / with an overhang
            s = rnd.randrange(n)
            B = A[s:s + m]
            if B and rnd.random() < 0.5:
                B[rnd.randrange(len(B))] ^= rnd.choice([1, 2, 3])
            if rnd.random() < 0.4:
                B = [(rnd.randrange(V) << 1) | rnd.randrange(2) for _ in range(rnd.randint(1, 3))] + B
            if rnd.random() < 0.3:
                B = km.rc(B)
        else:
            B = [(rnd.randrange(V) << 1) | rnd.randrange(2) for _ in range(m)]
        if not B or len(B) > n:
            continue
        if rnd.random() < 0.1:
            A[rnd.randrange(n)] = None   # a path step whose orientation equals nothing
        full = km.traceback_score(A, B)
        for window in (len(B), len(B) + 1, 16):
            if window < len(B):
                continue
            got, rows = km.traceback_score_skip(A, B, window)
            assert got == full, (A, B, window, got, full)
        got, rows = km.traceback_score_skip(A, B)
        if V == 100:
            total += n
            computed += rows
    assert computed < 0.7 * total
