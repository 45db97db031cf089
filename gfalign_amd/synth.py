"""Synthetic tangles for the path-scoring benchmark and parity tests.

Follows the frozen recipe of SURVEY.md section 8(d) (BASELINE.json configs 2-5):

* graph: ``V`` segments ``utig4-<i>``; a "truth" walk ``T`` from ``utig4-0+`` to
  ``utig4-<V-1>+`` over a random subset of the nodes, each visited
  ``1 + Geometric(0.8)`` times, capped at 1000 steps (the reference's MAX_N,
  include/alignments.h:246); ``L`` lines = consecutive pairs of ``T`` plus
  ``0.5 V`` random extra edges; node list = nodes of ``T`` with multiplicity.
* alignments: 85 % sub-walks of ``T`` (``m = 2 + min(Poisson(2.5), 30)``), 10 %
  sub-walks with one step replaced by another node of ``T`` (bad), 5 % with a
  node outside the node list (filtered); each reverse-complemented with
  probability 1/2.
* candidate batch: ``P`` prefixes of ``T`` with length uniform in ``[2, n_T]``,
  30 % with one substituted step.

All randomness comes from a counter-based SplitMix64 (``seed`` + stream id), so
the data is identical on every machine and numpy version.  Steps use the packed
encoding of include/gfalign_scorer.h: ``(node_id << 1) | minus``.
"""
import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class Rng:
    """SplitMix64 used as a counter-based generator: draw(stream, n)."""

    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self._next_stream = 0

    def bits(self, n):
        stream = np.uint64(self._next_stream)
        self._next_stream += 1
        with np.errstate(over="ignore"):
            base = _mix(self.seed * _GOLDEN + stream * _M2 + np.uint64(1))
            ctr = np.arange(1, n + 1, dtype=np.uint64)
            return _mix(base + ctr * _GOLDEN)

    def uniform(self, n):
        return (self.bits(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def below(self, n, bound):
        """Integers in [0, bound) (bound scalar or array); bias < 2^-32."""
        hi = (self.bits(n) >> np.uint64(32)).astype(np.uint64)
        b = np.asarray(bound, dtype=np.uint64)
        return ((hi * b) >> np.uint64(32)).astype(np.int64)


def _poisson_cdf(lam, kmax):
    p = math.exp(-lam)
    cdf = [p]
    for k in range(1, kmax + 1):
        p *= lam / k
        cdf.append(cdf[-1] + p)
    return np.array(cdf)


def make_truth_walk(V, n_T, rng):
    """Packed steps of the truth walk T (int32) and the node ids on it."""
    n_T = min(n_T, 1000)
    K = max(2, min(V, int(round(n_T / 1.25))))
    # K distinct nodes: source 0, destination V-1, and K-2 others
    order = np.argsort(rng.bits(V - 2), kind="stable") + 1 if V > 2 else np.zeros(0, np.int64)
    inner = order[:max(0, K - 2)]
    nodes = np.concatenate([[0], inner, [V - 1]]).astype(np.int64)
    extra = np.floor(np.log(np.maximum(rng.uniform(len(nodes)), 1e-300)) / math.log(0.2)).astype(np.int64)
    mult = 1 + extra
    # trim multiplicities so the walk fits the cap
    while mult.sum() > 1000:
        mult[np.argmax(mult)] -= 1
    visits = np.repeat(nodes, mult)
    # one visit of the source first, one of the destination last, rest shuffled
    first = np.flatnonzero(visits == 0)[0]
    last = np.flatnonzero(visits == V - 1)[-1]
    mask = np.ones(len(visits), bool)
    mask[[first, last]] = False
    mid = visits[mask]
    mid = mid[np.argsort(rng.bits(len(mid)), kind="stable")]
    walk = np.concatenate([[0], mid, [V - 1]])
    minus = (rng.bits(len(walk)) & np.uint64(1)).astype(np.int64)
    minus[0] = 0
    minus[-1] = 0
    T = ((walk << 1) | minus).astype(np.int32)
    return T, nodes, mult


def _subwalks(T, starts, lens):
    """CSR of T[starts[k] : starts[k]+lens[k]]."""
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    idx = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], lens) + np.repeat(starts, lens)
    return off, T[idx].astype(np.int32)


def make_alignments(T, V, on_walk_nodes, N, rng):
    """aln_off (int32), aln_steps (packed int32) for N alignments."""
    n_T = len(T)
    cdf = _poisson_cdf(2.5, 30)
    m = 2 + np.minimum(np.searchsorted(cdf, rng.uniform(N)), 30).astype(np.int64)
    m = np.minimum(m, n_T)
    starts = rng.below(N, (n_T - m + 1))
    off, steps = _subwalks(T, starts, m)
    kind = rng.uniform(N)          # <0.85 clean, <0.95 substituted, else outside
    sub_at = off[:-1] + rng.below(N, m)
    # substitution by another node of T (random orientation)
    repl_T = T[rng.below(N, n_T)].astype(np.int64) ^ (rng.bits(N) & np.uint64(1)).astype(np.int64)
    sel = (kind >= 0.85) & (kind < 0.95)
    steps[sub_at[sel]] = repl_T[sel].astype(np.int32)
    # a node outside the node list
    outside = np.setdiff1d(np.arange(V, dtype=np.int64), on_walk_nodes)
    sel = kind >= 0.95
    if len(outside):
        pick = outside[rng.below(N, len(outside))]
        repl_o = (pick << 1) | (rng.bits(N) & np.uint64(1)).astype(np.int64)
        steps[sub_at[sel]] = repl_o[sel].astype(np.int32)
    else:
        rng.below(N, 1), rng.bits(N)  # keep stream numbering independent of V
    # reverse-complement half of them
    flip = (rng.bits(N) & np.uint64(1)).astype(bool)
    flip_steps = np.repeat(flip, m)
    pos_in = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], m)
    mirror = np.repeat(off[:-1] + m - 1, m) - pos_in
    rc = steps[mirror] ^ 1
    steps = np.where(flip_steps, rc, steps).astype(np.int32)
    return off.astype(np.int32), steps


def make_candidates(T, P, rng):
    """path_off, path_steps: P prefixes of T, 30 % with one substituted step."""
    n_T = len(T)
    lens = 2 + rng.below(P, n_T - 1)           # uniform in [2, n_T]
    off, steps = _subwalks(T, np.zeros(P, np.int64), lens)
    sub = rng.uniform(P) < 0.30
    at = off[:-1] + 1 + rng.below(P, lens - 1)  # never the source step
    repl = T[rng.below(P, n_T)].astype(np.int64) ^ (rng.bits(P) & np.uint64(1)).astype(np.int64)
    steps[at[sub]] = repl[sub].astype(np.int32)
    return off.astype(np.int32), steps


class Tangle:
    """One synthetic workload (graph + alignments + candidate batch)."""

    def __init__(self, V, n_T, N, P, seed):
        rng = Rng(seed)
        self.V, self.N, self.P, self.seed = V, N, P, seed
        self.T, self.walk_nodes, self.walk_mult = make_truth_walk(V, n_T, rng)
        self.n_T = len(self.T)
        self.aln_off, self.aln_steps = make_alignments(self.T, V, self.walk_nodes, N, rng)
        self.path_off, self.path_steps = make_candidates(self.T, P, rng)
        self._rng = rng

    @property
    def S(self):
        return int(self.aln_off[-1])

    def algorithmic_bytes(self):
        """sum over candidates of 4S + 4(N+1) + 4n + 12."""
        n_total = int(self.path_off[-1])
        return self.P * (4 * self.S + 4 * (self.N + 1) + 12) + 4 * n_total

    # ---- text forms, for the CLI / file-format tests (small cases) ----
    def extra_edges(self):
        E = self.V // 2
        a = self._rng.below(E, self.V)
        b = self._rng.below(E, self.V)
        oa = self._rng.bits(E) & np.uint64(1)
        ob = self._rng.bits(E) & np.uint64(1)
        return [(int(x), int(p), int(y), int(q)) for x, p, y, q in zip(a, oa, b, ob)]

    def write_gfa(self, path):
        seen, lines = set(), ["H\tVN:Z:1.0"]
        for i in range(self.V):
            lines.append("S\tutig4-%d\t*\tLN:i:1000" % i)
        edges = [(int(s) >> 1, int(s) & 1, int(t) >> 1, int(t) & 1)
                 for s, t in zip(self.T[:-1], self.T[1:])] + self.extra_edges()
        for a, oa, b, ob in edges:
            if (a, oa, b, ob) in seen:
                continue
            seen.add((a, oa, b, ob))
            lines.append("L\tutig4-%d\t%s\tutig4-%d\t%s\t0M" % (a, "+-"[oa], b, "+-"[ob]))
        with open(path, "w") as f:
            f.write("\n".join(lines) + "\n")

    def write_nodelist(self, path):
        with open(path, "w") as f:
            for node, mult in zip(self.walk_nodes, self.walk_mult):
                f.write("utig4-%d\t%d\n" % (node, mult))

    def write_gaf(self, path):
        with open(path, "w") as f:
            for k in range(self.N):
                st = self.aln_steps[self.aln_off[k]:self.aln_off[k + 1]]
                p = "".join("%sutig4-%d" % ("<" if s & 1 else ">", s >> 1) for s in st)
                L = 1000 * len(st)
                f.write("read%d\t%d\t0\t%d\t+\t%s\t%d\t0\t%d\t%d\t%d\t60\n"
                        % (k, L, L, p, L, L, L, L))


# BASELINE.json configs; "smoke" is a seconds-scale stand-in.
CONFIGS = {
    "smoke":   dict(V=60, n_T=50, N=2000, P=48, seed=7),
    "config2": dict(V=500, n_T=400, N=100_000, P=1_000, seed=1),
    "config3": dict(V=2000, n_T=900, N=1_000_000, P=10_000, seed=2),
    "config5": dict(V=5000, n_T=1000, N=10_000_000, P=10_000, seed=3),
}


def make(name_or_kwargs):
    kw = CONFIGS[name_or_kwargs] if isinstance(name_or_kwargs, str) else name_or_kwargs
    return Tangle(**kw)
