"""CPU: oracle/gfalign_fast.c (the kernels' decision rule as a multi-threaded CPU
program; bench.py's cpu_fast line) against oracle/gfalign_oracle.c (the
reference's algorithm).  A third implementation next to the oracle and the HIP
kernels: all three must agree bit for bit."""
import random

import numpy as np
import pytest

import oracle
from gfalign_amd.scorer import GFAL_STEP_OTHER
from helpers import csr, random_case, walk_case


def both(alns, paths, flt, threads):
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
    got = oracle.fast_evaluate_paths(aoff, ast, poff, pst, flt, threads=threads)
    for name, g, e in zip(("bad", "good", "unaligned"), got, exp):
        assert np.array_equal(g, e), (name, flt)


@pytest.mark.parametrize("seed", range(6))
def test_tiny_alphabets(seed):
    rnd = random.Random(900 + seed)
    alns, paths = random_case(rnd, rnd.randint(1, 4), 400, 25, 12, 30)
    alns[3] = []
    for flt in (True, False):
        both(alns, paths, flt, threads=1 + seed % 3)


@pytest.mark.parametrize("seed", range(4))
def test_walks_and_other_orientation(seed):
    rnd = random.Random(950 + seed)
    alns, paths = walk_case(rnd, 30, 200, 600, 20, 25)
    for p in paths[:6]:                       # steps whose orientation equals nothing
        p[rnd.randrange(len(p))] |= GFAL_STEP_OTHER
    both(alns, paths, True, threads=2)
    both(alns, paths, False, threads=0)


def test_long_alignments_and_m_greater_n():
    rnd = random.Random(77)
    alns, paths = walk_case(rnd, 8, 300, 200, 12, 120)
    paths += [paths[0][:2], paths[1][:1]]
    both(alns, paths, True, threads=0)


def test_rejects_what_the_oracle_rejects():
    with pytest.raises(ValueError):
        oracle.fast_evaluate_paths([0, 1], [2], [0, 0], [], True)     # zero-step path
