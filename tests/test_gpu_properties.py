"""GPU: size-independent properties at BASELINE's full sizes (config 3:
2 k nodes, 1 M alignments, 10 k candidates), where the oracle is too slow to
run over everything."""
import numpy as np
import pytest

import oracle
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer
from helpers import csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tangle():
    return synth.make("config3")


@pytest.fixture(scope="module")
def full(gpu, tangle):
    t = tangle
    with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
        yield sc, sc.evaluate_paths(t.path_off, t.path_steps, True)


def test_counts_are_bounded_and_reproducible(full, tangle):
    sc, (bad, good, una) = full
    t = tangle
    assert np.all(bad.astype(np.int64) + good <= t.N)
    again = sc.evaluate_paths(t.path_off, t.path_steps, True)
    for a, b in zip((bad, good, una), again):
        assert np.array_equal(a, b)


def test_unaligned_matches_a_histogram(full, tangle):
    """unaligned(path) = alignment steps whose node is not on the path."""
    _, (_, _, una) = full
    t = tangle
    hist = np.bincount(t.aln_steps >> 1, minlength=t.V)
    for k in (0, 17, 4242, t.P - 1):
        nodes = np.unique(t.path_steps[t.path_off[k]:t.path_off[k + 1]] >> 1)
        assert una[k] == t.S - hist[nodes].sum()


def test_shards_add_up(full, tangle):
    """The multi-GPU contract: counters of disjoint alignment shards sum to the
    counters of the whole set (src/eval.cpp:80-106 is a sum over alignments)."""
    _, (bad, good, una) = full
    t = tangle
    cut = t.N // 3
    acc = [np.zeros(t.P, np.uint64) for _ in range(3)]
    for lo, hi in ((0, cut), (cut, t.N)):
        off = (t.aln_off[lo:hi + 1] - t.aln_off[lo]).astype(np.int32)
        st = t.aln_steps[t.aln_off[lo]:t.aln_off[hi]]
        with Scorer(off, st, t.V) as sc:
            for a, part in zip(acc, sc.evaluate_paths(t.path_off, t.path_steps, True)):
                a += part
    assert np.array_equal(acc[0], bad) and np.array_equal(acc[1], good)
    assert np.array_equal(acc[2], una)


@pytest.mark.parametrize("n_shards", [3, 8])
def test_sharded_create_partitions_the_set(full, tangle, n_shards):
    """gfal_scorer_create_sharded (what bench.py --gpus N and `--devices N` use):
    the shards of one alignment set own every alignment exactly once, none is
    empty or holds the bulk of the set, and their counters add up to the unsharded
    ones at the full BASELINE size."""
    _, (bad, good, una) = full
    t = tangle
    acc = [np.zeros(t.P, np.uint64) for _ in range(3)]
    n_owned, steps_owned = [], []
    for k in range(n_shards):
        with Scorer(t.aln_off, t.aln_steps, t.V, shard=(k, n_shards)) as sc:
            info = sc.info()
            n_owned.append(info["n_aln"])
            steps_owned.append(info["n_steps"])
            for a, part in zip(acc, sc.evaluate_paths(t.path_off, t.path_steps, True)):
                a += part
    assert sum(n_owned) == t.N and sum(steps_owned) == int(t.aln_off[-1])
    # (a shard holds few alignment lengths -- each costs it a prologue per tile -- so what
    # balances is groups plus a charge per length, not the step count: DESIGN.md section 6)
    assert max(steps_owned) < 2.2 * sum(steps_owned) / n_shards and min(n_owned) > 0
    assert np.array_equal(acc[0], bad) and np.array_equal(acc[1], good)
    assert np.array_equal(acc[2], una)


def test_truth_walk_explains_every_clean_alignment(full, tangle):
    """Scoring the whole truth walk: every alignment that is an exact sub-walk
    (either strand) is good; counted here independently with numpy."""
    sc, _ = full
    t = tangle
    T = t.T
    bad, good, _ = sc.evaluate_paths([0, len(T)], T, True)
    # independent count: windows of T (and of rc(T)) as byte strings
    Trc = (T[::-1] ^ 1).astype(np.int32)
    m = np.diff(t.aln_off)
    exp_good = 0
    for L in np.unique(m):
        win = {T[s:s + L].tobytes() for s in range(len(T) - L + 1)}
        win |= {Trc[s:s + L].tobytes() for s in range(len(T) - L + 1)}
        idx = np.flatnonzero(m == L)
        rows = t.aln_steps[(t.aln_off[idx][:, None] + np.arange(L)[None, :])]
        exp_good += sum(r.tobytes() in win for r in rows)
    # start-overhangs can add a few more goods, never fewer
    assert good[0] >= exp_good
    assert good[0] - exp_good <= 0.01 * t.N
    assert bad[0] + good[0] <= t.N


def test_sample_of_full_batch_is_pair_exact(full, tangle):
    """12 of the 10 k candidates against a 60 k-alignment slice, oracle-exact."""
    t = tangle
    pick = np.linspace(0, t.P - 1, 12).astype(int)
    paths = [t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick]
    poff, pst = csr(paths)
    hi = 60_000
    off, st = t.aln_off[:hi + 1], t.aln_steps[:t.aln_off[hi]]
    with Scorer(off, st, t.V) as sc:
        got = sc.evaluate_paths(poff, pst, True)
    exp = oracle.evaluate_paths(off, st, poff, pst, True)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


def test_config5_size_shards_and_oracle_sample(gpu):
    """BASELINE config 5's size (5 k nodes, 10 M alignments): the eight shards of
    an 8-GPU run add up to the unsharded counters on a sample of the candidate
    batch, `unaligned` matches the histogram, and four of the paths agree with
    the oracle on the first 60 k alignments."""
    t = synth.make("config5")
    pick = np.linspace(0, t.P - 1, 96).astype(int)
    poff, pst = csr([t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick])
    with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
        bad, good, una = sc.evaluate_paths(poff, pst, True)
    assert np.all(bad.astype(np.int64) + good <= t.N)
    hist = np.bincount(t.aln_steps >> 1, minlength=t.V)
    for k in (0, 50, 95):
        nodes = np.unique(pst[poff[k]:poff[k + 1]] >> 1)
        assert una[k] == t.S - hist[nodes].sum()
    acc = [np.zeros(len(pick), np.uint64) for _ in range(3)]
    for r in range(8):
        with Scorer(t.aln_off, t.aln_steps, t.V, shard=(r, 8)) as sc:
            for a, part in zip(acc, sc.evaluate_paths(poff, pst, True)):
                a += part
    assert np.array_equal(acc[0], bad) and np.array_equal(acc[1], good) and np.array_equal(acc[2], una)
    n_sub = 60000
    aoff = t.aln_off[:n_sub + 1]
    ast = t.aln_steps[:aoff[-1]]
    sub = [5, 40, 70, 95]
    soff, sst = csr([pst[poff[k]:poff[k + 1]] for k in sub])
    with Scorer(aoff, ast, t.V) as sc:
        got = sc.evaluate_paths(soff, sst, True)
    exp = oracle.evaluate_paths(aoff, ast, soff, sst, True)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


def test_full_size_counters_equal_the_fast_cpu_checker(full, tangle):
    """Every fourth path of the config-3 batch against ALL 1 M alignments, on the
    CPU with oracle/gfalign_fast.c (pinned to the oracle by
    tests/test_oracle_fast.py): the HIP counters at the full BASELINE size, bit
    for bit -- not a property, the numbers themselves."""
    _, (bad, good, una) = full
    t = tangle
    pick = np.arange(0, t.P, 4)
    poff, pst = csr([t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick])
    eb, eg, eu = oracle.fast_evaluate_paths(t.aln_off, t.aln_steps, poff, pst, True, threads=16)
    assert np.array_equal(bad[pick], eb)
    assert np.array_equal(good[pick], eg)
    assert np.array_equal(una[pick], eu)


def test_dedup_scorer_gives_the_same_counters_at_full_size(full, tangle):
    """gfal_scorer_create_dedup on the config-3 set: the same counters for the
    whole batch from far fewer resident alignments."""
    _, (bad, good, una) = full
    t = tangle
    with Scorer(t.aln_off, t.aln_steps, t.V, dedup=True) as sc:
        info = sc.info()
        assert info["n_aln"] == t.N and info["n_lanes"] < t.N // 2
        b, g, u = sc.evaluate_paths(t.path_off, t.path_steps, True)
    assert np.array_equal(b, bad) and np.array_equal(g, good) and np.array_equal(u, una)
