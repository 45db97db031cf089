"""CPU: the oracle against every known-answer vector held for this path."""
import json
import os

import numpy as np

import oracle
from helpers import (GOLDEN, csr, load_appendix_c, parse_path_string,
                     random3_alignments)


def test_random3_evaluate_path_with_filter():
    """bad/good of the four search extensions."""
    ids, _, alns = random3_alignments()
    aoff, ast = csr(alns)
    gold = load_appendix_c()["evaluate_path_filter"]
    paths = [parse_path_string(g["path"], ids) for g in gold]
    poff, pst = csr(paths)
    bad, good, _ = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    assert bad.tolist() == [g["bad"] for g in gold]
    assert good.tolist() == [g["good"] for g in gold]


def test_random3_eval_path_scores():
    """per-alignment best scores and the summary row."""
    ids, _, alns = random3_alignments()
    aoff, ast = csr(alns)
    gold = load_appendix_c()["eval_path"]
    path = parse_path_string(gold["path"], ids)
    fw, rc = oracle.pair_scores(aoff, ast, path)
    assert np.maximum(fw, rc).tolist() == gold["best_scores"]
    bad, good, una = oracle.evaluate_paths(aoff, ast, [0, len(path)], path, False)
    assert (int(bad[0]), int(good[0]), int(una[0])) == (gold["bad"], gold["good"], 0)
    alt = int(bad[0]) - int(good[0]) - len(set(s >> 1 for s in path))
    assert gold["stdout"][-1] == "%d\t%d\t%d\t%d\t%d" % (bad[0], good[0], alt, len(path), 4)


def test_overhang_examples():
    """a start-overhang can be free, or not."""
    a1, a2, a3, x = 0, 2, 4, 6
    fw, _ = oracle.pair_scores([0, 3], [x, a1, a2], [a1, a2, a3])
    assert fw[0] == 0
    fw, _ = oracle.pair_scores([0, 3], [x, a1, a1], [a1, a1, a1])
    assert fw[0] < 0


def test_test6_counters_are_zero_without_alignments():
    """validateFiles/test.6.tst runs without -g: every row has bad = good = 0."""
    with open(os.path.join(GOLDEN, "reference_testfiles", "test.6.tst")) as f:
        rows = [l.rstrip("\n").split("\t") for l in f.read().splitlines()[2:]]
    ids = {str(k + 1): k for k in range(5)}
    for r in rows:
        path = parse_path_string(r[7], ids)
        bad, good, una = oracle.evaluate_paths([0], [], [0, len(path)], path, True)
        assert (int(bad[0]), int(good[0])) == (int(r[1]), int(r[2])) == (0, 0)
        assert int(r[3]) == -len(set(s >> 1 for s in path))
        assert int(r[4]) == len(path)


def test_quirks_of_the_reference_table():
    # longer than the path -> always good (row 0 is initialised up to n only)
    fw, rc = oracle.pair_scores([0, 3], [10, 12, 14], [0, 2])
    assert fw[0] == 0 and rc[0] == 0
    # overhang at the END of the path is never free
    fw, _ = oracle.pair_scores([0, 3], [2, 4, 6], [0, 2, 4])
    assert fw[0] < 0
    # zero-step alignment: free traceback
    bad, good, _ = oracle.evaluate_paths([0, 0], [], [0, 2], [0, 2], True)
    assert (bad[0], good[0]) == (0, 1)
    # orientation that is neither + nor - equals nothing but still counts as a node
    OTHER = 0x40000000
    bad, good, una = oracle.evaluate_paths([0, 1], [0], [0, 2], [OTHER | 0, 2], True)
    assert (bad[0], good[0], una[0]) == (1, 0, 0)


def test_committed_kernel_cases_reproduce():
    """The oracle built here agrees with the one that wrote kernel_cases.json."""
    with open(os.path.join(GOLDEN, "kernel_cases.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 4
    for c in cases:
        for key, flt in (("filter", True), ("nofilter", False)):
            bad, good, una = oracle.evaluate_paths(c["aln_off"], c["aln_steps"],
                                                   c["path_off"], c["path_steps"], flt)
            assert bad.tolist() == c[key]["bad"], c["name"]
            assert good.tolist() == c[key]["good"], c["name"]
            assert una.tolist() == c[key]["unaligned"], c["name"]
        p0 = c["path_steps"][c["path_off"][0]:c["path_off"][1]]
        fw, rc = oracle.pair_scores(c["aln_off"], c["aln_steps"], p0)
        assert fw.tolist() == c["pair_scores_path0"]["fw"]
        assert rc.tolist() == c["pair_scores_path0"]["rc"]
