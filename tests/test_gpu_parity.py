"""GPU: the HIP path, called through the C ABI, against the CPU oracle
(bit-exact integer counters) and the committed golden vectors."""
import json
import os
import random

import numpy as np
import pytest

import oracle
from gfalign_amd import scorer, synth
from gfalign_amd.scorer import GFAL_STEP_OTHER, Scorer, ScorerError
from helpers import (GOLDEN, csr, load_appendix_c, parse_path_string,
                     random3_alignments, random_case, walk_case)

pytestmark = pytest.mark.gpu


def check(alns, paths, n_nodes, filters=(True, False)):
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    with Scorer(aoff, ast, n_nodes) as sc:
        for flt in filters:
            bad, good, una = sc.evaluate_paths(poff, pst, flt)
            ebad, egood, euna = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
            assert np.array_equal(bad, ebad), ("bad", flt)
            assert np.array_equal(good, egood), ("good", flt)
            assert np.array_equal(una, euna), ("unaligned", flt)


def test_library_is_the_in_tree_hip_build(gpu):
    assert os.path.samefile(os.path.dirname(scorer.library_path()),
                            os.path.join(os.path.dirname(scorer.__file__), "csrc"))
    assert scorer.device_count() >= 1
    # the binary that is loaded was compiled from the sources next to it
    from gfalign_amd import build as gbuild
    assert scorer.load_library().gfal_build_id().decode() == gbuild.scorer_build_id()


def test_random3_appendix_c(gpu):
    ids, _, alns = random3_alignments()
    aoff, ast = csr(alns)
    gold = load_appendix_c()
    paths = [parse_path_string(g["path"], ids) for g in gold["evaluate_path_filter"]]
    poff, pst = csr(paths)
    with Scorer(aoff, ast, 5) as sc:
        bad, good, _ = sc.evaluate_paths(poff, pst, True)
        assert bad.tolist() == [g["bad"] for g in gold["evaluate_path_filter"]]
        assert good.tolist() == [g["good"] for g in gold["evaluate_path_filter"]]
        ep = gold["eval_path"]
        path = parse_path_string(ep["path"], ids)
        fw, rc = sc.pair_scores(path)
        assert np.maximum(fw, rc).tolist() == ep["best_scores"]
        bad, good, una = sc.evaluate_paths([0, len(path)], path, False)
        assert (bad[0], good[0], una[0]) == (ep["bad"], ep["good"], 0)


def test_committed_kernel_cases(gpu):
    with open(os.path.join(GOLDEN, "kernel_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        with Scorer(c["aln_off"], c["aln_steps"], c["n_nodes"]) as sc:
            for key, flt in (("filter", True), ("nofilter", False)):
                bad, good, una = sc.evaluate_paths(c["path_off"], c["path_steps"], flt)
                assert bad.tolist() == c[key]["bad"], (c["name"], key)
                assert good.tolist() == c[key]["good"], (c["name"], key)
                assert una.tolist() == c[key]["unaligned"], (c["name"], key)
            p0 = c["path_steps"][c["path_off"][0]:c["path_off"][1]]
            fw, rc = sc.pair_scores(p0)
            assert fw.tolist() == c["pair_scores_path0"]["fw"], c["name"]
            assert rc.tolist() == c["pair_scores_path0"]["rc"], c["name"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_tiny_alphabets(gpu, seed):
    """Few nodes -> many repeats, start-overhangs and traceback ties: the
    exact-DP kernel decides a large share of the pairs."""
    rnd = random.Random(seed)
    alns, paths = random_case(rnd, rnd.randint(1, 4), 3000, 150, 8, 12)
    check(alns, paths, 8)


@pytest.mark.parametrize("seed", [4, 5])
def test_fuzz_walks(gpu, seed):
    rnd = random.Random(seed)
    alns, paths = walk_case(rnd, 25, 120, 4000, 120, 14)
    check(alns, paths, 32)


@pytest.mark.parametrize("max_m,max_n", [(16, 24), (32, 40), (48, 64)])
def test_fuzz_dp_length_classes(gpu, max_m, max_n):
    """Two or three nodes and long alignments: start-overhangs in every
    length class of the exact-DP kernels (rows in 8/16/32 registers, LDS)."""
    rnd = random.Random(100 + max_m)
    alns, paths = random_case(rnd, rnd.randint(2, 3), 1200, 60, max_m, max_n,
                              min_m=max(1, max_m // 2 - 2), min_n=max_m // 2)
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    with Scorer(aoff, ast, 4) as sc:
        got = sc.evaluate_paths(poff, pst, True)
        assert sc.info()["dp_pairs"] > 500
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


@pytest.mark.parametrize("limit", ["0", "4000000000"])
@pytest.mark.parametrize("max_m,max_n", [(7, 40), (30, 70), (64, 80), (90, 120)])
def test_both_dp_kernel_families(gpu, monkeypatch, limit, max_m, max_n):
    """The exact DP has a throughput family (one pair per lane, k_dp_regs /
    k_dp_long) and a latency family (one column per lane, k_dp_sys) chosen by
    the worklist length; force each on the same overhang-heavy inputs."""
    rnd = random.Random(300 + max_m)
    alns, paths = random_case(rnd, rnd.randint(2, 3), 900, 48, max_m, max_n,
                              min_m=1, min_n=max(2, max_m // 2))
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    monkeypatch.setenv("GFAL_DP_SYS_LIMIT", limit)
    with Scorer(aoff, ast, 4) as sc:
        got = sc.evaluate_paths(poff, pst, True)
        assert sc.info()["dp_pairs"] > 300
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


def test_long_alignments_take_the_generic_path(gpu):
    """Alignments longer than the 16 register-resident steps."""
    rnd = random.Random(6)
    alns, paths = walk_case(rnd, 40, 300, 1500, 40, 60)
    assert max(len(a) for a in alns) > 16
    check(alns, paths, 64)


def test_maximum_sizes(gpu):
    """n = m = 1000 (reference MAX_N - 1), repeats included."""
    rnd = random.Random(7)
    walk = [(rnd.randrange(300) << 1) | rnd.randrange(2) for _ in range(1000)]
    alns = [walk, [s ^ 1 for s in reversed(walk)], walk[1:], walk[:999],
            walk[:500] + [walk[500] ^ 1] + walk[501:], [walk[-1]] + walk[:-1]]
    alns += [walk[s:s + rnd.randint(1, 30)] for s in
             (rnd.randrange(0, 970) for _ in range(300))]
    paths = [walk, walk[:999], walk[1:], walk[:2]]
    check(alns, paths, 300)


def test_edge_cases(gpu):
    # zero-step alignments, single steps, a node nobody aligns to, m > n
    alns = [[], [0], [1], [2, 4], [], [4, 2], [6, 6, 6], [0, 2, 4, 6, 8, 10]]
    paths = [[0], [1], [0, 2, 4], [2, 4], [20, 22], [6, 6], [4, 2, 0]]
    check(alns, paths, 16)
    # empty shard
    with Scorer([0], [], 4) as sc:
        bad, good, una = sc.evaluate_paths([0, 2], [0, 2], True)
        assert (bad[0], good[0], una[0]) == (0, 0, 0)
    # only zero-step alignments
    check([[], [], []], [[0, 2]], 4)


def test_other_orientation_steps(gpu):
    """A path step whose orientation is neither + nor - (evalPath can produce
    one): equals no alignment step, still counts for the filter."""
    O = GFAL_STEP_OTHER
    alns = [[0], [1], [0, 2], [2, 0], [3, 1], [4], [0, 2, 4]]
    paths = [[O | 0, 2], [0, O | 2, 4], [O | 0], [O | 4, 2, 0], [0, 2, O | 4]]
    check(alns, paths, 8)


def test_argument_errors(gpu):
    with Scorer([0, 2], [0, 2], 4) as sc:
        with pytest.raises(scorer.ScorerError) as e:
            sc.evaluate_paths([0, 1001], [0] * 1001, True)
        assert e.value.code == -2
        with pytest.raises(scorer.ScorerError) as e:
            sc.evaluate_paths([0, 0], [], True)          # empty path
        assert e.value.code == -2
        with pytest.raises(scorer.ScorerError) as e:
            sc.evaluate_paths([0, 1], [99 << 1], True)   # node id out of range
        assert e.value.code == -2
    with pytest.raises(scorer.ScorerError) as e:
        Scorer([0, 1001], [0] * 1001, 4)
    assert e.value.code == -2
    with pytest.raises(scorer.ScorerError) as e:
        Scorer([0, 1], [8 << 1], 4)
    assert e.value.code == -2


def test_smoke_config_all_paths(gpu):
    t = synth.make("smoke")
    aoff, ast, poff, pst = t.aln_off, t.aln_steps, t.path_off, t.path_steps
    with Scorer(aoff, ast, t.V) as sc:
        for flt in (True, False):
            got = sc.evaluate_paths(poff, pst, flt)
            exp = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
            for g, e in zip(got, exp):
                assert np.array_equal(g, e)


def test_config2_sample_against_oracle(gpu):
    """BASELINE config 2 (500 nodes, 100 k alignments): 24 of the 1000
    candidates, spread over the length range, checked pair-exactly."""
    t = synth.make("config2")
    order = np.argsort(np.diff(t.path_off), kind="stable")
    pick = order[np.linspace(0, t.P - 1, 24).astype(int)]
    paths = [t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick]
    poff, pst = csr(paths)
    with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
        bad, good, una = sc.evaluate_paths(t.path_off, t.path_steps, True)
        ebad, egood, euna = oracle.evaluate_paths(t.aln_off, t.aln_steps, poff, pst, True)
        assert np.array_equal(bad[pick], ebad)
        assert np.array_equal(good[pick], egood)
        assert np.array_equal(una[pick], euna)
        # batch composition must not matter
        b2, g2, u2 = sc.evaluate_paths(poff, pst, True)
        assert np.array_equal(b2, ebad) and np.array_equal(g2, egood) and np.array_equal(u2, euna)


def test_universe_folds_outside_nodes(gpu):
    """A GAF over a huge node space (more distinct nodes than the device tables
    hold) scored against a small tangle: with the universe given, results are
    identical to the oracle on the unreduced input."""
    rnd = random.Random(31)
    big = 200_000                              # node-id space
    tangle = [rnd.randrange(big) for _ in range(40)]
    walk = [(rnd.choice(tangle) << 1) | rnd.randrange(2) for _ in range(120)]
    alns = []
    for _ in range(3000):
        m = rnd.randint(1, 9)
        s0 = rnd.randrange(0, len(walk) - m)
        b = list(walk[s0:s0 + m])
        if rnd.random() < 0.5:                 # touches the rest of the genome
            b[rnd.randrange(m)] = (rnd.randrange(big) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    alns += [[(rnd.randrange(big) << 1)] for _ in range(40_000)]   # > 32766 distinct nodes
    paths = [walk[:k] for k in (2, 17, 60, 120)] + [walk[5:50]]
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    with pytest.raises(scorer.ScorerError) as e:
        Scorer(aoff, ast, big)                 # no universe: too many nodes
    assert e.value.code == -2
    with Scorer(aoff, ast, big, universe=sorted(set(tangle))) as sc:
        assert sc.info()["n_local_nodes"] <= len(set(tangle)) + 1
        for flt in (True, False):
            got = sc.evaluate_paths(poff, pst, flt)
            exp = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
            for g, x in zip(got, exp):
                assert np.array_equal(g, x), flt
        outside = next(v for v in range(big) if v not in set(tangle))
        with pytest.raises(scorer.ScorerError) as e:
            sc.evaluate_paths([0, 2], [walk[0], outside << 1], True)
        assert e.value.code == -2


def test_sharded_scorers_fill_pair_scores_between_them(gpu):
    """evalPath over several devices: every shard writes the fw / rc scores of
    its own alignments into the same arrays (zero-step alignments included)."""
    rnd = random.Random(52)
    alns, paths = walk_case(rnd, 30, 200, 700, 3, 20)
    alns[5] = []
    alns[77] = []
    aoff, ast = csr(alns)
    path = paths[0]
    with Scorer(aoff, ast, 64) as sc:
        exp_fw, exp_rc = sc.pair_scores(path)
    fw = np.full(len(alns), -12345, np.int32)
    rc = np.full(len(alns), -12345, np.int32)
    owned_total = 0
    for k in range(4):
        with Scorer(aoff, ast, 64, shard=(k, 4)) as sc:
            owned_total += sc.info()["n_aln"]
            sc.pair_scores(path, out=(fw, rc))
    assert owned_total == len(alns)
    assert np.array_equal(fw, exp_fw) and np.array_equal(rc, exp_rc)
    ofw, orc = oracle.pair_scores(aoff, ast, path)
    assert np.array_equal(fw, ofw) and np.array_equal(rc, orc)


def test_more_than_32k_paths_in_one_batch(gpu):
    """Batches above 32 768 paths sort by length with the three-kernel counting
    sort instead of the one-workgroup one; results are per path either way."""
    rnd = random.Random(61)
    alns, base = walk_case(rnd, 12, 60, 300, 40, 8)
    paths = [base[k % len(base)][:rnd.randint(1, 30)] for k in range(33000)]
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    with Scorer(aoff, ast, 16) as sc:
        bad, good, una = sc.evaluate_paths(poff, pst, True)
        # the same paths in small batches (one-workgroup sort)
        for lo in (0, 15000, 32990):
            hi = min(lo + 10, len(paths))
            o2, s2 = csr(paths[lo:hi])
            b2, g2, u2 = sc.evaluate_paths(o2, s2, True)
            assert np.array_equal(b2, bad[lo:hi]) and np.array_equal(g2, good[lo:hi])
            assert np.array_equal(u2, una[lo:hi])
    sample = list(range(0, len(paths), 1500))
    o3, s3 = csr([paths[k] for k in sample])
    eb, eg, eu = oracle.evaluate_paths(aoff, ast, o3, s3, True)
    assert np.array_equal(bad[sample], eb) and np.array_equal(good[sample], eg)
    assert np.array_equal(una[sample], eu)


def test_graph_replay_gives_the_same_counters(gpu, monkeypatch):
    """GFAL_GRAPHS=1: the blocking call captures its launches into a HIP graph and
    updates the executable from call to call (batch shapes change)."""
    rnd = random.Random(71)
    alns, paths = walk_case(rnd, 25, 150, 900, 60, 14)
    aoff, ast = csr(alns)
    monkeypatch.setenv("GFAL_GRAPHS", "1")
    with Scorer(aoff, ast, 32) as sc:
        for lo, hi in ((0, 7), (7, 40), (3, 4), (0, 60), (10, 11)):
            poff, pst = csr(paths[lo:hi])
            got = sc.evaluate_paths(poff, pst, True)
            exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
            for g, e in zip(got, exp):
                assert np.array_equal(g, e)


def test_more_shards_than_items(gpu):
    """Eight shards of three alignments: most shards own nothing (and one owns the
    zero-step alignment); the counters still add up."""
    alns = [[2, 4, 6], [], [7, 5], [2, 4]]
    paths = [[2, 4, 6, 8], [6, 4, 2], [9], [2, 4]]
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    acc = [np.zeros(len(paths), np.uint64) for _ in range(3)]
    owned = 0
    for k in range(8):
        with Scorer(aoff, ast, 8, shard=(k, 8)) as sc:
            owned += sc.info()["n_aln"]
            for a, part in zip(acc, sc.evaluate_paths(poff, pst, True)):
                a += part
    assert owned == len(alns)
    for a, e in zip(acc, exp):
        assert np.array_equal(a, e.astype(np.uint64))


@pytest.mark.parametrize("seed", [81, 82, 83])
def test_dedup_scorer_matches_the_oracle(gpu, seed):
    """gfal_scorer_create_dedup: identical alignments collapsed into weighted
    lanes.  Counters (scan and both DP kernel families), pair scores and shards
    are those of the uncollapsed set."""
    rnd = random.Random(seed)
    if seed == 81:
        base, paths = random_case(rnd, 3, 120, 30, 10, 40)       # overhang-heavy
    else:
        base, paths = walk_case(rnd, 25, 200, 150, 30, 20 if seed == 82 else 70)
    alns = []
    for b in base:                                                 # 1 .. 40 copies each
        alns += [list(b) for _ in range(rnd.choice([1, 1, 2, 5, 40]))]
    rnd.shuffle(alns)
    alns[7] = []
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    for flt in (True, False):
        exp = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
        with Scorer(aoff, ast, 32, dedup=True) as sc:
            info = sc.info()
            assert info["n_aln"] == len(alns) and info["n_lanes"] < 0.7 * len(alns)
            got = sc.evaluate_paths(poff, pst, flt)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e), flt
    # shards of a dedup scorer, with the wavefront kernels forced
    os.environ["GFAL_DP_SYS_LIMIT"] = "4000000000"
    try:
        acc = [np.zeros(len(paths), np.uint64) for _ in range(3)]
        fw = np.full(len(alns), -777, np.int32)
        rc = np.full(len(alns), -777, np.int32)
        for k in range(3):
            with Scorer(aoff, ast, 32, shard=(k, 3), dedup=True) as sc:
                for a, part in zip(acc, sc.evaluate_paths(poff, pst, True)):
                    a += part
                sc.pair_scores(paths[0], out=(fw, rc))
    finally:
        del os.environ["GFAL_DP_SYS_LIMIT"]
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    for a, e in zip(acc, exp):
        assert np.array_equal(a, e.astype(np.uint64))
    ofw, orc = oracle.pair_scores(aoff, ast, paths[0])
    assert np.array_equal(fw, ofw) and np.array_equal(rc, orc)


def test_worklist_overflow_grows_the_list_once(gpu, monkeypatch):
    """More exact-DP pairs than the worklist holds: the blocking API grows the
    list to the need the pass itself counted and runs the batch once more; the
    next call on the same scorer fits without a re-run."""
    rnd = random.Random(41)
    alns, paths = random_case(rnd, 2, 2500, 64, 7, 10, min_m=2, min_n=4)
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    monkeypatch.setenv("GFAL_DEBUG_WL_CAPACITY", "1")      # capacity = max(n_aln, 1)
    with Scorer(aoff, ast, 4) as sc:
        assert sc.info()["wl_capacity"] == len(alns)
        got = sc.evaluate_paths(poff, pst, True)
        info = sc.info()
        assert info["dp_pairs"] > len(alns)                  # it did not fit the first list
        assert info["n_overflow_reruns"] == 1 and info["n_device_passes"] == 2
        assert info["wl_capacity"] >= info["dp_pairs"]
        again = sc.evaluate_paths(poff, pst, True)
        info = sc.info()
        assert info["n_overflow_reruns"] == 1 and info["n_device_passes"] == 3
        assert info["n_score_calls"] == 2
    for g, a, x in zip(got, again, exp):
        assert np.array_equal(g, x) and np.array_equal(a, x)


def test_worklist_overflow_splits_the_batch(gpu, monkeypatch):
    """A list that may not grow (GFAL_DEBUG_WL_NO_GROW; in production: a batch
    that needs more than the 2 x 8 GiB limit): the blocking API halves the batch
    until every piece fits (a single path always does)."""
    rnd = random.Random(41)
    alns, paths = random_case(rnd, 2, 2500, 64, 7, 10, min_m=2, min_n=4)
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    monkeypatch.setenv("GFAL_DEBUG_WL_CAPACITY", "1")
    monkeypatch.setenv("GFAL_DEBUG_WL_NO_GROW", "1")
    with Scorer(aoff, ast, 4) as sc:
        got = sc.evaluate_paths(poff, pst, True)
        info = sc.info()
        assert info["dp_pairs"] <= len(alns)                 # the last piece fitted
        assert info["wl_capacity"] == len(alns) and info["n_overflow_reruns"] >= 1
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    for g, x in zip(got, exp):
        assert np.array_equal(g, x)


class _Hip:
    """Device buffers for the device-resident API without torch (raw HIP runtime
    through ctypes: the test process has usually initialised HIP through the scorer
    library long before torch would)."""

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.lib = ctypes.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.lib.hipFree.argtypes = [ctypes.c_void_p]
        self.bufs = []

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.c.c_void_p()
        assert self.lib.hipMalloc(self.c.byref(p), max(arr.nbytes, 4)) == 0
        assert self.lib.hipMemcpy(p, arr.ctypes.data_as(self.c.c_void_p), arr.nbytes, 1) == 0   # H2D
        self.bufs.append(p)
        return p.value

    def download(self, ptr, n, dtype):
        out = np.zeros(n, dtype)
        assert self.lib.hipDeviceSynchronize() == 0
        assert self.lib.hipMemcpy(out.ctypes.data_as(self.c.c_void_p), self.c.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.bufs:
            self.lib.hipFree(p)


def test_device_api_reports_overflow_and_grows(gpu, monkeypatch):
    """score_device cannot re-run by itself: sync_status reports the overflow,
    grows the list, and the caller's second attempt fits."""
    rnd = random.Random(43)
    alns, paths = random_case(rnd, 2, 2000, 48, 7, 10, min_m=2, min_n=4)
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    monkeypatch.setenv("GFAL_DEBUG_WL_CAPACITY", "1")
    P = len(paths)
    with Scorer(aoff, ast, 4) as sc:
        hip = _Hip()
        try:
            d_off = hip.upload(np.asarray(poff, np.int32))
            d_st = hip.upload(np.asarray(pst, np.int32))
            d_cnt = hip.upload(np.zeros(3 * P, np.uint32))
            args = (d_off, d_st, P, len(pst), max(len(p) for p in paths), True, d_cnt, 0)
            sc.score_device(*args)
            with pytest.raises(ScorerError) as e:
                sc.sync_status()
            assert e.value.code == -5 and "grown" in str(e.value)
            sc.score_device(*args)
            sc.sync_status()
            counts = hip.download(d_cnt, 3 * P, np.uint32)
        finally:
            hip.free()
    for g, x in zip((counts[:P], counts[P:2 * P], counts[2 * P:]), exp):
        assert np.array_equal(g, x)


@pytest.mark.parametrize("n", [33, 64, 998, 999, 1000])
def test_windows_at_the_ends_of_the_path(gpu, n):
    """Alignments of every register-resident length (and just beyond) cut from
    the very start and the very end of the path, both strands, exact and with
    one step changed: exercises the dword window reads next to the array ends."""
    rnd = random.Random(50 + n)
    path = [(rnd.randrange(120) << 1) | rnd.randrange(2) for _ in range(n)]
    alns = []
    for m in list(range(1, 36)) + [40]:
        if m > n:
            continue
        for piece in (path[:m], path[n - m:], path[1:1 + m], path[n - m - 1:n - 1]):
            if len(piece) != m:
                continue
            alns.append(list(piece))
            alns.append([x ^ 1 for x in reversed(piece)])
            bad = list(piece)
            bad[-1] ^= 1
            alns.append(bad)
            bad = list(piece)
            bad[0] = (121 << 1)
            alns.append(bad)
    check(alns, [path, path[:n - 1], path[1:]], 128)
