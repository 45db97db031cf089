"""GPU parity of search-mode scoring (gfal_group_score_children): a candidate
scored from its parent gives the counters of the full evaluation, bit for bit.

Checkers: the oracle (reference src/eval.cpp:67-108 restated) on the full child
paths, and the product's own full scorer on the larger cases.  The identities
behind the kernel are fuzzed on the CPU in tests/test_incr_model.py.
"""
import random

import numpy as np
import pytest

import oracle
from gfalign_amd.scorer import GFAL_STEP_OTHER, Group, Scorer, ScorerError
from helpers import csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from gfalign_amd import scorer
    if scorer.device_count() < 1:
        pytest.fail("these tests need an MI355X: the product path has no CPU fallback")


def walk_alignments(rnd, walk, n_nodes, n_aln, max_m, mutate=0.2, zero=0.01):
    alns = []
    for _ in range(n_aln):
        m = 0 if rnd.random() < zero else rnd.randint(1, max_m)
        s = rnd.randrange(0, len(walk) - m + 1)
        b = list(walk[s:s + m])
        if b and rnd.random() < mutate:
            b[rnd.randrange(len(b))] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    return alns


class Tree:
    """Host side of a search: paths by slot, batches of (parent, step, slot)."""

    def __init__(self, group, n_slots):
        self.g = group
        self.g.store_reserve(n_slots)
        self.paths = {}          # slot -> full path
        self.free = list(range(n_slots - 1, -1, -1))

    def store(self, paths):
        slots = [self.free.pop() for _ in paths]
        off, st = csr(paths)
        res = self.g.score_store(off, st, slots)
        for s, p in zip(slots, paths):
            self.paths[s] = list(p)
        return slots, res

    def children(self, batch, keep=True):
        """batch: list of (parent, step) with parent = ('slot', s) or ('batch', j)."""
        parent, step, slot, full = [], [], [], []
        for par, st in batch:
            if par[0] == "slot":
                parent.append(par[1])
                base = self.paths[par[1]]
            else:
                parent.append(~par[1])
                base = full[par[1]]
            step.append(st)
            full.append(base + [st])
            slot.append(self.free.pop() if keep else -1)
        res = self.g.score_children(parent, step, slot, max(len(p) for p in full))
        for s, p in zip(slot, full):
            if s >= 0:
                self.paths[s] = p
        return slot, full, res


def expect(alns, paths):
    aoff, ast = csr(alns)
    poff, pst = csr(paths)
    return oracle.evaluate_paths(aoff, ast, poff, pst, True)


def assert_same(got, exp, what):
    for name, g, e in zip(("bad", "good", "unaligned"), got, exp):
        assert np.array_equal(g, e), (what, name, np.flatnonzero(np.asarray(g) != np.asarray(e))[:8],
                                      np.asarray(g)[:8], np.asarray(e)[:8])


@pytest.mark.parametrize("seed,n_nodes,max_m", [(1, 2, 4), (2, 3, 5), (3, 6, 6), (4, 25, 8), (5, 40, 3)])
def test_children_equal_the_full_evaluation(gpu, seed, n_nodes, max_m):
    """Chains and siblings, repeated and new nodes, tiny alphabets (palindromic
    windows, windows that occur again), zero-step alignments."""
    rnd = random.Random(seed)
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(90)]
    alns = walk_alignments(rnd, walk, n_nodes, 3000, max_m)
    aoff, ast = csr(alns)
    with Scorer(aoff, ast, n_nodes + 2) as sc, Group([sc]) as g:
        tree = Tree(g, 400)
        roots = [walk[s:s + max_m + rnd.randrange(3)] for s in (0, 5, 11)]
        slots, res = tree.store(roots)
        assert_same(res, expect(alns, roots), "stored roots")
        frontier = [(s, list(p)) for s, p in zip(slots, roots)]
        for rnd_no in range(6):
            batch, bases = [], []
            for s, p in frontier:
                nxt = walk[(walk.index(p[-1]) + 1) % len(walk)] if p[-1] in walk else walk[0]
                # the walk's own continuation, a random step, the reverse of the last step
                for st in {nxt, (rnd.randrange(n_nodes + 1) << 1) | rnd.randrange(2), p[-1] ^ 1}:
                    batch.append((("slot", s), st))
                    bases.append(p)
            # a chain below the first child, inside the same batch
            k0 = len(batch)
            chain_from = 0
            for d in range(4):
                st = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
                batch.append((("batch", chain_from), st))
                chain_from = len(batch) - 1
            slot, full, res = tree.children(batch)
            assert_same(res, expect(alns, full), "round %d" % rnd_no)
            assert len(full[k0 + 3]) == len(full[0]) + 4
            pick = rnd.sample(range(len(full)), min(4, len(full)))
            frontier = [(slot[k], full[k]) for k in pick]


def test_other_orientation_and_nodes_without_alignments(gpu):
    rnd = random.Random(9)
    n_nodes = 8
    walk = [(rnd.randrange(6) << 1) | rnd.randrange(2) for _ in range(60)]
    alns = walk_alignments(rnd, walk, 6, 1500, 5)
    aoff, ast = csr(alns)
    with Scorer(aoff, ast, n_nodes) as sc, Group([sc]) as g:
        tree = Tree(g, 64)
        root = [GFAL_STEP_OTHER | (walk[0] & ~1)] + walk[1:7]      # the search's first step: orientation '0'
        slots, res = tree.store([root])
        assert_same(res, expect(alns, [root]), "root")
        batch = [(("slot", slots[0]), walk[7]),
                 (("slot", slots[0]), (7 << 1) | 1),                # a node no alignment has
                 (("slot", slots[0]), GFAL_STEP_OTHER | (3 << 1)),  # a step that equals nothing
                 (("batch", 1), walk[8]), (("batch", 2), walk[8]), (("batch", 3), (7 << 1))]
        _, full, res = tree.children(batch)
        assert_same(res, expect(alns, full), "children")


def test_first_node_list_longer_than_the_bitmaps(gpu, monkeypatch):
    """GFAL_BITS_MAX_LIST caps the verdict bitmaps: a path whose first node sits in more
    alignments than that keeps no bitmap (its children recompute, nothing is written past
    a slot's words), paths on other first nodes go on inheriting."""
    monkeypatch.setenv("GFAL_BITS_MAX_LIST", "96")
    rnd = random.Random(77)
    n_nodes, max_m = 12, 5
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(120)]
    hub = walk[0]
    # many alignments carry the hub node, few carry the other roots' first nodes
    alns = walk_alignments(rnd, walk, n_nodes, 2500, max_m)
    alns += [[walk[3], hub, walk[1]][:rnd.randint(2, 3)] for _ in range(400)]
    aoff, ast = csr(alns)
    with Scorer(aoff, ast, n_nodes + 2) as sc, Group([sc]) as g:
        tree = Tree(g, 300)
        roots = [walk[0:max_m + 1], walk[7:7 + max_m + 2], walk[20:20 + max_m + 1]]
        slots, res = tree.store(roots)
        assert_same(res, expect(alns, roots), "stored roots")
        frontier = [(s, list(p)) for s, p in zip(slots, roots)]
        for rnd_no in range(5):
            batch = []
            for s, p in frontier:
                i = (walk.index(p[-1]) + 1) % len(walk)
                for st in {walk[i], hub, p[-1] ^ 1}:
                    batch.append((("slot", s), st))
            chain_from = 0
            for d in range(3):
                batch.append((("batch", chain_from), walk[(5 * d + rnd_no) % len(walk)]))
                chain_from = len(batch) - 1
            slot, full, res = tree.children(batch)
            assert_same(res, expect(alns, full), "round %d" % rnd_no)
            frontier = [(slot[k], full[k]) for k in range(0, len(full), 3)][:5]


@pytest.mark.parametrize("n_shards", [1, 2])
def test_a_batch_beyond_the_worklist_limit_is_split(gpu, monkeypatch, n_shards):
    """GFAL_DEBUG_WL_NO_GROW stands for a batch whose exact-DP need is beyond what the lists
    may grow to: gfal_group_score_end runs it in halves (plain, stored and children batches;
    a parent inside a children batch is named by its slot once an earlier piece scored it)."""
    monkeypatch.setenv("GFAL_DEBUG_WL_CAPACITY", "1")       # = one entry per alignment
    monkeypatch.setenv("GFAL_DEBUG_WL_NO_GROW", "1")
    rnd = random.Random(5)
    n_nodes, max_m = 3, 7
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(100)]
    alns = walk_alignments(rnd, walk, n_nodes, 3000, max_m, mutate=0.5)
    aoff, ast = csr(alns)
    scs = [Scorer(aoff, ast, n_nodes + 2, shard=(k, n_shards)) for k in range(n_shards)]
    try:
        with Group(scs) as g:
            roots = [walk[s:s + max_m + 3] for s in range(0, 40, 3)]
            poff, pst = csr(roots)
            res = g.evaluate_paths(poff, pst, True)
            assert_same(res, expect(alns, roots), "plain batch")
            assert sum(sc.info()["n_overflow_reruns"] for sc in scs) >= 1
            tree = Tree(g, 400)
            slots, res = tree.store(roots)
            assert_same(res, expect(alns, roots), "stored batch")
            batch = []
            for s, p in zip(slots, roots):
                for st in {walk[(walk.index(p[-1]) + 1) % len(walk)], p[-1] ^ 1, p[0]}:
                    batch.append((("slot", s), st))
            # a chain of children inside the batch, its links spread over it (they cross the halves)
            base, batch, links = batch, [], []
            every = max(2, len(base) // 6)
            for k, entry in enumerate(base):
                batch.append(entry)
                if k % every == every - 1:
                    batch.append((("batch", links[-1] if links else 1), walk[(7 * k) % len(walk)]))
                    links.append(len(batch) - 1)
            assert len(links) >= 4
            before = sum(sc.info()["n_overflow_reruns"] for sc in scs)
            slot, full, res = tree.children(batch)
            assert_same(res, expect(alns, full), "children batch")
            assert sum(sc.info()["n_overflow_reruns"] for sc in scs) > before
            # the pieces kept their paths: a later batch may name them
            batch2 = [(("slot", slot[k]), walk[k % len(walk)]) for k in links]
            _, full2, res2 = tree.children(batch2, keep=False)
            assert_same(res2, expect(alns, full2), "grandchildren")
    finally:
        for sc in scs:
            sc.close()


@pytest.mark.parametrize("dedup", [False, True])
def test_shards_and_dedup(gpu, dedup):
    """Two shards of one set on one device (counters added on the host) and the
    weighted-lane scorer: the store and the deltas are per shard."""
    rnd = random.Random(21)
    n_nodes = 12
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(120)]
    alns = walk_alignments(rnd, walk, n_nodes, 6000, 7, mutate=0.1)
    aoff, ast = csr(alns)
    scs = [Scorer(aoff, ast, n_nodes, shard=(k, 2), dedup=dedup) for k in range(2)]
    try:
        with Group(scs) as g:
            tree = Tree(g, 256)
            slots, res = tree.store([walk[:9], walk[3:14]])
            assert_same(res, expect(alns, [walk[:9], walk[3:14]]), "roots")
            batch = [(("slot", slots[0]), walk[9]), (("slot", slots[1]), walk[14]), (("batch", 0), walk[10]),
                     (("batch", 2), walk[11]), (("slot", slots[0]), walk[0]), (("batch", 4), walk[1] ^ 1)]
            slot, full, res = tree.children(batch)
            assert_same(res, expect(alns, full), "children")
            # the children as parents of a later batch
            batch2 = [(("slot", slot[3]), walk[12]), (("slot", slot[5]), walk[2])]
            _, full2, res2 = tree.children(batch2, keep=False)
            assert_same(res2, expect(alns, full2), "grandchildren")
    finally:
        for s in scs:
            s.close()


def test_long_paths_many_candidates_against_the_full_scorer(gpu):
    """A tangle-sized case: 60 000 alignments of up to 24 steps over 300 nodes,
    parents of 40..400 steps, 300 children in one batch (so that k_child runs
    with one chunk and with several); checked against the product's full scorer
    and, on a sample, the oracle."""
    rnd = random.Random(33)
    n_nodes = 300
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(450)]
    # revisit stretches so that windows repeat
    walk[200:230] = walk[20:50]
    walk[300:320] = [x ^ 1 for x in reversed(walk[60:80])]
    alns = walk_alignments(rnd, walk, n_nodes, 60000, 24, mutate=0.1, zero=0.001)
    aoff, ast = csr(alns)
    with Scorer(aoff, ast, n_nodes) as sc, Group([sc]) as g:
        tree = Tree(g, 2048)
        cuts = [40, 77, 150, 199, 229, 310, 399]
        roots = [walk[:c] for c in cuts]
        slots, res = tree.store(roots)
        poff, pst = csr(roots)
        assert_same(res, g.evaluate_paths(poff, pst, True), "roots vs full")
        batch = []
        for s, c in zip(slots, cuts):
            for _ in range(42):
                r = rnd.random()
                st = walk[c] if r < 0.4 else (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
                batch.append((("slot", s), st))
        for k in range(6):                                   # chains in the batch
            batch.append((("batch", k * 42), walk[cuts[k] + 1]))
        slot, full, res = tree.children(batch)
        poff, pst = csr(full)
        assert_same(res, g.evaluate_paths(poff, pst, True), "children vs full")
        # the next generations: parents that were children themselves carry the DP verdicts
        # of the alignments on their first node, and their children inherit them unless the
        # path's tail comes back to such an alignment (steps near the start do: recomputed)
        gen, gen_full = slot, full
        for g_no in range(3):
            batch2, picks = [], rnd.sample(range(len(gen_full) - 6), 60)
            for k in picks:
                nxt = walk[len(gen_full[k])] if len(gen_full[k]) < len(walk) else walk[3]
                for st in (nxt, walk[rnd.randrange(0, 6)] ^ rnd.randrange(2), (rnd.randrange(n_nodes) << 1) | 1):
                    batch2.append((("slot", gen[k]), st))
            for k in range(5):
                batch2.append((("batch", 3 * k), walk[rnd.randrange(0, 40)]))
                batch2.append((("batch", len(batch2) - 1), walk[rnd.randrange(0, 40)] ^ 1))
            gen, gen_full, res2 = tree.children(batch2)
            poff2, pst2 = csr(gen_full)
            assert_same(res2, g.evaluate_paths(poff2, pst2, True), "generation %d vs full" % (g_no + 2))
        sample = rnd.sample(range(len(full)), 12)
        exp = expect(alns[:8000], [full[k] for k in sample])
        with Scorer(aoff[:8001], ast[:aoff[8000]], n_nodes) as small, Group([small]) as g2:
            t2 = Tree(g2, 64)
            par = [full[k][:-1] for k in sample]
            s2, _ = t2.store(par)
            _, f2, r2 = t2.children([(("slot", s), full[k][-1]) for s, k in zip(s2, sample)], keep=False)
            assert_same(r2, exp, "sample vs oracle")
        for chunks in ("1", "7"):
            import os
            os.environ["GFAL_CHILD_CHUNKS"] = chunks
            try:
                _, full3, res3 = tree.children(batch[:50], keep=False)
            finally:
                del os.environ["GFAL_CHILD_CHUNKS"]
            assert_same(res3, [r[:50] for r in res], "chunks " + chunks)


def test_bad_batches_are_refused(gpu):
    rnd = random.Random(2)
    walk = [(rnd.randrange(9) << 1) | rnd.randrange(2) for _ in range(40)]
    alns = walk_alignments(rnd, walk, 9, 500, 6, zero=0.0)
    aoff, ast = csr(alns)
    with Scorer(aoff, ast, 9) as sc, Group([sc]) as g:
        with pytest.raises(ScorerError):                       # no store yet
            g.score_children([0], [2], [-1], 10)
        tree = Tree(g, 8)
        m = max(len(b) for b in alns)
        slots, _ = tree.store([walk[:m - 1], walk[:m + 2]])
        with pytest.raises(ScorerError):                       # parent shorter than the longest alignment
            g.score_children([slots[0]], [walk[m - 1]], [-1], 40)
        with pytest.raises(ScorerError):                       # slot outside the store
            g.score_children([99], [2], [-1], 40)
        with pytest.raises(ScorerError):                       # in-batch parent that is not earlier
            g.score_children([slots[1], ~1], [2, 2], [-1, -1], 40)
        with pytest.raises(ScorerError):                       # child longer than stated
            g.score_children([slots[1]], [2], [-1], m + 2)
        # and the scorer still works
        _, full, res = tree.children([(("slot", slots[1]), walk[m + 2])], keep=False)
        assert_same(res, expect(alns, full), "after the refusals")


def test_short_list_dp_kernel_hands_a_long_list_back(gpu):
    """After a call with a short exact-DP list the next children call runs one DP launch
    on the unsorted list (k_dp_small); if its own list turns out long, the kernel reports
    it and the blocking call runs the batch again with the sort and the register
    kernels.  45 000 two-step alignments [v, a0] are start-overhang candidates of every
    path through v that starts with a0: a chain of 40 new nodes makes 130 000 pairs."""
    a0 = 0
    alns = []
    for v in range(1, 301):
        alns += [[(v << 1), (a0 << 1)]] * 150
    alns += [[(a0 << 1), (1 << 1)], [(2 << 1) | 1, (1 << 1) | 1]] * 3
    aoff, ast = csr(alns)
    with Scorer(aoff, ast, 512) as sc, Group([sc]) as g:
        tree = Tree(g, 128)
        root = [(a0 << 1), (1 << 1), (2 << 1)]
        slots, res = tree.store([root])
        assert_same(res, expect(alns, [root]), "root")
        _, full, res = tree.children([(("slot", slots[0]), (400 << 1))], keep=False)     # a short list
        assert_same(res, expect(alns, full), "short list")
        before = sc.info()["n_overflow_reruns"]
        batch = [(("slot", slots[0]), (3 << 1))]
        for k in range(1, 40):
            batch.append((("batch", k - 1), ((3 + k) << 1)))
        _, full, res = tree.children(batch, keep=False)
        assert_same(res, expect(alns, full), "long list")
        info = sc.info()
        assert info["dp_pairs"] > 32768 and info["n_overflow_reruns"] == before + 1
        _, full, res = tree.children(batch[:5], keep=False)                               # and on it goes
        assert_same(res, expect(alns, full), "after the re-run")
