"""GPU parity of the scan kernels.

k_scan   walks occurrence chains per (item, tile path); it keeps the rare
         alignment lengths and the batches of a few dozen paths.
k_scan2  (round 2) answers the subpath test for all 8 paths of a tile at once from
         a hashed window table per (tile, alignment length).
k_scan3  (round 3, the default) does it for 31 paths by content identity: k_tile
         looks the tile's windows up in the scorer's content table, k_scan3 probes
         a table of content indices (k_tile_masks, k_overhang beside them).

`GFAL_SCAN=1` / `=2` / `=3` force one kernel for everything, `GFAL_HASH_MIN_ITEMS=1`
makes every length a k_scan3 segment in the automatic mode, `GFAL_SCAN3_NMG=1`
keeps k_scan3's node masks in HBM (what a tangle of many nodes does by itself),
`GFAL_SCAN3_SLOTS=4096` halves its table (more passes over unrelated paths); every
case below is compared bit for bit with the oracle (reference src/eval.cpp:67-108
restated) under each of them.  The reference has no fixture for this path beyond
tests/golden (parity unpinned): the oracle is the checker.
"""
import random

import numpy as np
import pytest

import oracle
from gfalign_amd.scorer import GFAL_STEP_OTHER, Scorer
from helpers import csr, random_case, walk_case

pytestmark = pytest.mark.gpu

MODES = [{"GFAL_SCAN": "1"}, {"GFAL_SCAN": "2"}, {"GFAL_SCAN": "3"}, {"GFAL_HASH_MIN_ITEMS": "1"}, {},
         {"GFAL_SCAN": "3", "GFAL_SCAN3_NMG": "1"}, {"GFAL_SCAN": "3", "GFAL_SCAN3_SLOTS": "4096"}]


@pytest.fixture(scope="module")
def gpu():
    from gfalign_amd import scorer
    if scorer.device_count() < 1:
        pytest.fail("these tests need an MI355X: the product path has no CPU fallback")


def check_modes(monkeypatch, alns, paths, n_nodes, filters=(True, False), **kw):
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = {f: oracle.evaluate_paths(aoff, ast, poff, pst, f) for f in filters}
    infos = []
    for env in MODES:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with Scorer(aoff, ast, n_nodes, **kw) as sc:
            for f in filters:
                got = sc.evaluate_paths(poff, pst, f)
                for name, g, e in zip(("bad", "good", "unaligned"), got, exp[f]):
                    assert np.array_equal(g, e), (env, f, name, np.flatnonzero(g != e)[:8])
            infos.append(sc.info())
        for k in env:
            monkeypatch.delenv(k)
    return infos


@pytest.mark.parametrize("seed", [11, 12])
def test_every_length_in_both_kernels(gpu, monkeypatch, seed):
    """Alignments of 1..40 steps cut from one walk (exact-length code up to 12
    steps, pair-count code up to 33, run-time loops beyond), 120 paths: enough
    for k_scan2 in the automatic mode."""
    rnd = random.Random(seed)
    alns, paths = walk_case(rnd, 30, 160, 5000, 120, 40)
    check_modes(monkeypatch, alns, paths, 32)


def test_very_long_alignments(gpu, monkeypatch):
    """Alignments of 100..220 steps: beyond what k_tile counts as a run of agreeing
    positions (127), so every path enters such windows itself; the reference path's
    indices stop at 16 steps; the overhang test of such pairs is done pair by pair."""
    rnd = random.Random(15)
    n_nodes = 50
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(700)]
    paths = [walk[:rnd.randint(230, 700)] for _ in range(70)]
    for p in paths[::7]:
        p[rnd.randrange(0, 200)] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
    alns = []
    for _ in range(3000):
        m = rnd.choice((100, 126, 127, 128, 129, 180, 220))
        s0 = rnd.randrange(0, 700 - m)
        b = list(walk[s0:s0 + m])
        if rnd.random() < 0.3:
            b[rnd.randrange(m)] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    check_modes(monkeypatch, alns, paths, n_nodes, filters=(True,))


def test_tile_boundaries(gpu, monkeypatch):
    """1, 2, 31, 32, 33, 62, 63 and 97 paths: tiles of 31 with a short last one."""
    rnd = random.Random(16)
    alns, paths = walk_case(rnd, 25, 120, 4000, 97, 9)
    aoff, ast = csr(alns)
    monkeypatch.setenv("GFAL_SCAN", "3")
    with Scorer(aoff, ast, 32) as sc:
        for n in (1, 2, 31, 32, 33, 62, 63, 97):
            poff, pst = csr(paths[:n])
            for f in (True, False):
                got = sc.evaluate_paths(poff, pst, f)
                exp = oracle.evaluate_paths(aoff, ast, poff, pst, f)
                for name, g, e in zip(("bad", "good", "unaligned"), got, exp):
                    assert np.array_equal(g, e), (n, f, name)


def test_list_arena_in_slabs(gpu, monkeypatch):
    """GFAL_SCAN3_LIST_MB=1: the per-tile window lists do not fit the arena at once, so the
    tiles go through k_tile_masks / k_tile / k_scan3 in slabs (one tile at a time here; the
    reference array and the cold arguments come with the first slab only)."""
    rnd = random.Random(29)
    alns, paths = walk_case(rnd, 40, 150, 5000, 130, 10)
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    monkeypatch.setenv("GFAL_SCAN", "3")
    monkeypatch.setenv("GFAL_SCAN3_LIST_MB", "1")
    with Scorer(aoff, ast, 48) as sc:
        for f in (True, False):
            got = sc.evaluate_paths(poff, pst, f)
            exp = oracle.evaluate_paths(aoff, ast, poff, pst, f)
            for name, g, e in zip(("bad", "good", "unaligned"), got, exp):
                assert np.array_equal(g, e), (f, name, np.flatnonzero(g != e)[:8])


def test_tiny_alphabet_overhangs(gpu, monkeypatch):
    """Two to three nodes: repeats everywhere, so the table holds few distinct
    windows with many duplicates, nearly every alignment touches the paths'
    first node (the triage runs for most lanes) and many pairs reach the DP."""
    rnd = random.Random(21)
    alns, paths = random_case(rnd, 3, 4000, 130, 9, 30, min_m=1, min_n=2)
    infos = check_modes(monkeypatch, alns, paths, 4)
    assert all(i["dp_pairs"] > 1000 for i in infos)


def test_unrelated_long_paths_take_several_passes(gpu, monkeypatch):
    """Eight unrelated 1000-step paths have ~16 000 distinct windows per length:
    more than one table holds (4096), so a workgroup rebuilds the table and
    passes over its items again for the rest of the tile's paths."""
    rnd = random.Random(31)
    n_nodes = 600
    paths = [[(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(rnd.randint(940, 1000))]
             for _ in range(100)]
    alns = []
    for _ in range(6000):
        p = paths[rnd.randrange(len(paths))]
        m = rnd.randint(1, 9)
        s = rnd.randrange(0, len(p) - m)
        b = list(p[s:s + m])
        if rnd.random() < 0.2:
            b[rnd.randrange(m)] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    check_modes(monkeypatch, alns, paths, n_nodes, filters=(True,))


def test_siblings_and_prefixes_share_their_windows(gpu, monkeypatch):
    """What a search submits: extensions of one parent (all steps but the last
    shared), prefixes of one walk, and walks that differ in one early step --
    the base path of a tile enters the shared windows for all of them."""
    rnd = random.Random(41)
    n_nodes = 40
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(400)]
    paths = []
    for k in range(60):                                   # siblings
        paths.append(walk[:200] + [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2)])
    for k in range(40):                                   # prefixes
        paths.append(walk[:rnd.randint(2, 400)])
    for k in range(40):                                   # one early substitution
        p = list(walk[:rnd.randint(150, 400)])
        p[rnd.randrange(0, 20)] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        paths.append(p)
    alns = []
    for _ in range(8000):
        m = rnd.randint(1, 14)
        s = rnd.randrange(0, 400 - m)
        b = list(walk[s:s + m])
        if rnd.random() < 0.15:
            b[rnd.randrange(m)] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    check_modes(monkeypatch, alns, paths, n_nodes)


def test_substituted_steps_with_and_without_support(gpu, monkeypatch):
    """k_tile's two shortcuts for positions with windows of their own.  The batch is 100
    prefixes of one walk; the FIRST path of the second and third tile carries a substituted
    step (the other paths' windows there are the reference's: copied, not looked up), other
    paths carry their own.  Half of the substituted junctions also occur in alignments, on
    either strand (the pair filter must let those windows through: they are contents), the
    other half in none (skipped without a lookup)."""
    rnd = random.Random(23)
    V = 60
    walk = [(rnd.randrange(V) << 1) | rnd.randrange(2) for _ in range(160)]
    lens = sorted({rnd.randint(30, 160) for _ in range(400)}, reverse=True)[:100]
    paths = [list(walk[:n]) for n in lens]
    subs = {}                      # path index (by descending length: the scorer's order) -> position
    for k in (31, 62, 5, 17, 40, 41, 70, 93):
        q = rnd.randint(3, len(paths[k]) - 4)
        paths[k][q] = ((V + k % 3) << 1) | rnd.randrange(2)      # a node the walk never visits
        subs[k] = q
    alns = []
    for _ in range(3000):          # pieces of the walk, some reverse-complemented
        m = rnd.randint(1, 12)
        st = rnd.randrange(0, len(walk) - m)
        b = list(walk[st:st + m])
        alns.append([x ^ 1 for x in reversed(b)] if rnd.random() < 0.5 else b)
    for k, q in subs.items():      # windows over half of the substituted steps, both strands
        if k % 2:
            continue
        for _ in range(20):
            lo = rnd.randint(max(0, q - 6), q)
            hi = rnd.randint(q + 1, min(len(paths[k]), q + 7))
            b = list(paths[k][lo:hi])
            alns.append([x ^ 1 for x in reversed(b)] if rnd.random() < 0.5 else b)
    rnd.shuffle(paths)
    check_modes(monkeypatch, alns, paths, V + 4)


def test_tiles_whose_paths_start_differently(gpu, monkeypatch):
    """The triage's shortcut (only alignments on the tile's first node can
    overhang) needs one first step per tile; tiles without one take the exact
    test for every open lane.  Also a first step of neither orientation
    (GFAL_STEP_OTHER: the search's source before its first extension)."""
    rnd = random.Random(51)
    alns, paths = random_case(rnd, 6, 5000, 128, 7, 24, min_m=1, min_n=2)
    for k in range(0, len(paths), 5):
        paths[k][0] = GFAL_STEP_OTHER | (paths[k][0] & ~1)
    check_modes(monkeypatch, alns, paths, 8)


def test_dedup_and_shards_through_k_scan2_and_k_scan3(gpu, monkeypatch):
    """Weighted lanes (gfal_scorer_create_dedup) and a 3-way sharded set give
    the counters of the plain scorer under k_scan2 and k_scan3 as well."""
    rnd = random.Random(61)
    alns, paths = walk_case(rnd, 20, 100, 6000, 110, 10)
    alns = alns + alns[:2000] + alns[:500]                 # copies: weights up to 3
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    for mode in ("2", "3"):
        monkeypatch.setenv("GFAL_SCAN", mode)
        with Scorer(aoff, ast, 32, dedup=True) as sc:
            got = sc.evaluate_paths(poff, pst, True)
            assert sc.info()["n_lanes"] < len(alns)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)
        acc = [np.zeros(len(paths), np.uint64) for _ in range(3)]
        for k in range(3):
            with Scorer(aoff, ast, 32, shard=(k, 3)) as sc:
                for a, part in zip(acc, sc.evaluate_paths(poff, pst, True)):
                    a += part
        for a, e in zip(acc, exp):
            assert np.array_equal(a, e.astype(np.uint64))


@pytest.mark.parametrize("n_nodes", [15000, 24000])
def test_many_nodes_need_no_chain_images(gpu, n_nodes):
    """15 000 / 24 000 nodes: k_scan's path images (a first-occurrence table per
    node) no longer fit the LDS; k_scan2 only keeps node masks per node and takes
    every length.  (Round 1 answered GFAL_E_RANGE above ~11 000 nodes; the reference
    has no limit, include/nodetable.h:11-43; this build's is ~25 600 nodes inside the
    tangle, refused at create.)"""
    rnd = random.Random(71)
    walk = [(v << 1) | rnd.randrange(2) for v in rnd.sample(range(n_nodes), n_nodes)]
    paths = [walk[s:s + rnd.randint(50, 600)] for s in (rnd.randrange(0, n_nodes - 600) for _ in range(100))]
    alns = []
    for _ in range(4000):
        m = rnd.randint(1, 12)
        s = rnd.randrange(0, n_nodes - m)
        b = list(walk[s:s + m])
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    # make sure every node occurs in some alignment (all of them become local nodes)
    for s in range(0, n_nodes - 8, 8):
        alns.append(walk[s:s + 8])
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    with Scorer(aoff, ast, n_nodes) as sc:
        assert sc.info()["n_local_nodes"] >= n_nodes - 1000
        got = sc.evaluate_paths(poff, pst, True)
        few = sc.evaluate_paths(poff[:9], pst[:poff[8]], True)     # a small batch as well
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    for g, f, e in zip(got, few, exp):
        assert np.array_equal(g, e) and np.array_equal(f, e[:8])


def test_scorer_groups_sum_with_rccl_or_on_the_host(gpu, monkeypatch):
    """gfal_group_*: the shards of a set scored together.  One shard per device:
    the counters are summed by an RCCL all-reduce on the devices (a one-GPU box
    can only form the one-rank group, which still runs ncclAllReduce); shards that
    share a device, or GFAL_GROUP_HOST_SUM=1: summed on the host.  Same integers."""
    from gfalign_amd.scorer import Group
    rnd = random.Random(81)
    alns, paths = walk_case(rnd, 20, 100, 5000, 120, 10)
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    exp = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    monkeypatch.setenv("GFAL_GROUP_RCCL_SINGLE", "1")       # (a group of one normally skips the all-reduce)
    with Scorer(aoff, ast, 32) as sc, Group([sc]) as g:
        assert g.uses_rccl
        for _ in range(2):
            got = g.evaluate_paths(poff, pst, True)
            for a, e in zip(got, exp):
                assert np.array_equal(a, e)
        g.begin(poff, pst, False)                           # the two halves, with host work in between
        other = oracle.evaluate_paths(aoff, ast, poff, pst, False)
        for a, e in zip(g.end(), other):
            assert np.array_equal(a, e)
    monkeypatch.delenv("GFAL_GROUP_RCCL_SINGLE")
    with Scorer(aoff, ast, 32) as sc, Group([sc]) as g:
        assert not g.uses_rccl
        for a, e in zip(g.evaluate_paths(poff, pst, True), exp):
            assert np.array_equal(a, e)
    shards = [Scorer(aoff, ast, 32, shard=(k, 3)) for k in range(3)]
    try:
        with Group(shards) as g:
            assert not g.uses_rccl                       # three shards on one device
            got = g.evaluate_paths(poff, pst, True)
    finally:
        for s in shards:
            s.close()
    for a, e in zip(got, exp):
        assert np.array_equal(a, e)
    monkeypatch.setenv("GFAL_GROUP_HOST_SUM", "1")
    with Scorer(aoff, ast, 32) as sc, Group([sc]) as g:
        assert not g.uses_rccl
        got = g.evaluate_paths(poff, pst, False)
    for a, e in zip(got, oracle.evaluate_paths(aoff, ast, poff, pst, False)):
        assert np.array_equal(a, e)
