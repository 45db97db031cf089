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
