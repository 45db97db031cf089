"""Pure-Python model of the parent -> child scoring rule (scorer.hip k_child).

Test-only.  `gfalign search` extends a path by one step at a time
(reference src/eval.cpp:146-162), so a candidate is always `parent + [s]`.
With the filter on (src/eval.cpp:81-91) the counters of src/eval.cpp:92-98
split into

    Pass(A)  alignments all of whose nodes are on A            (bad + good)
    G1(A)    of those: zero-step, longer than A, or a contiguous subpath of A
             on either strand                                  (good without DP)
    G2(A)    of the rest: the start-overhang pairs the exact DP accepts

and for A' = A + [s], len(A) >= the longest alignment:

    Pass(A') = Pass(A) + #{B : node(s) in B, nodes(B) on A'}   if node(s) is new
    G1(A')   = G1(A) + sum over M in (Lmax, Mmax] of mult(W_M) + mult(rc(W_M))
               (once if W_M == rc(W_M)),  W_M = the last M steps of A',
               Lmax = the longest W_M that already occurs in A on either strand
    G2(A')   is recomputed: its candidates all contain the node of A'[0]

`tests/test_incr_model.py` checks the identities against the direct rule
(tests/kernel_model.py, itself fuzzed against the oracle).
"""
from kernel_model import has_overhang, rc, traceback_score


def contains(A, B):
    m = len(B)
    return any(A[i:i + m] == B for i in range(len(A) - m + 1))


def found(A, B):
    return contains(A, B) or contains(A, rc(B))


def direct(A, alns):
    """(Pass, G1, G2, candidates) of path A by the plain rule."""
    nodes = {a >> 1 for a in A}
    n = len(A)
    Pass = G1 = G2 = 0
    cand = []
    for k, B in enumerate(alns):
        if any((b >> 1) not in nodes for b in B):
            continue
        Pass += 1
        if len(B) == 0 or len(B) > n or found(A, B):
            G1 += 1
            continue
        fw, rv = has_overhang(A, B), has_overhang(A, rc(B))
        if fw or rv:
            cand.append(k)
            if (fw and traceback_score(A, B) == 0) or (rv and traceback_score(A, rc(B)) == 0):
                G2 += 1
    return Pass, G1, G2, cand


class Index:
    """What `create` prepares for the children call."""

    def __init__(self, alns):
        self.alns = alns
        self.mult = {}
        self.inv = {}
        self.max_m = max([len(B) for B in alns] + [0])
        for k, B in enumerate(alns):
            self.mult[tuple(B)] = self.mult.get(tuple(B), 0) + 1
            for v in {b >> 1 for b in B}:
                self.inv.setdefault(v, []).append(k)


def longest_old_window(A, s):
    """Lmax: the longest suffix window of A + [s] that occurs in A, either strand."""
    n = len(A)
    best = 0
    for p in range(n):
        if A[p] == s:                      # same strand: compare backwards
            L = 1
            while p - L >= 0 and A[p - L] == A[n - L]:
                L += 1
            best = max(best, L)
        if A[p] == s ^ 1:                  # other strand: rc(window) read forwards
            L = 1
            while p + L < n and A[p + L] == A[n - L] ^ 1:
                L += 1
            best = max(best, L)
    return best


def child(A, s, Pass, G1, idx):
    """(Pass', G1') of A + [s] from the parent's; needs len(A) >= idx.max_m."""
    assert len(A) >= idx.max_m
    A2 = A + [s]
    v = s >> 1
    nodes = {a >> 1 for a in A}
    if v not in nodes:
        nodes.add(v)
        Pass += sum(1 for k in idx.inv.get(v, []) if all((b >> 1) in nodes for b in idx.alns[k]))
    lmax = longest_old_window(A, s)
    for M in range(lmax + 1, min(len(A2), idx.max_m) + 1):
        W = tuple(A2[len(A2) - M:])
        R = tuple(rc(list(W)))
        G1 += idx.mult.get(W, 0)
        if R != W:
            G1 += idx.mult.get(R, 0)
    return Pass, G1


def dp_candidates(A, idx):
    """The DP worklist of A from the inverted list of its first node."""
    nodes = {a >> 1 for a in A}
    out = []
    for k in idx.inv.get(A[0] >> 1, []):
        B = idx.alns[k]
        if len(B) > len(A) or any((b >> 1) not in nodes for b in B):
            continue
        if (has_overhang(A, B) or has_overhang(A, rc(B))) and not found(A, B):
            out.append(k)
    return out


def dp_status(A, B):
    """None: B is no DP candidate of A; else whether the DP accepts it (what G2 counts)."""
    nodes = {a >> 1 for a in A}
    if len(B) == 0 or len(B) > len(A) or any((b >> 1) not in nodes for b in B) or found(A, B):
        return None
    fw, rv = has_overhang(A, B), has_overhang(A, rc(B))
    if not (fw or rv):
        return None
    return (fw and traceback_score(A, B) == 0) or (rv and traceback_score(A, rc(B)) == 0)


def inherits(A2, d, B):
    """k_child's test: may B's DP status be taken from the ancestor d steps up the path A2?
    Yes if none of the last len(B) + d steps of A2 is on a node of B: the d new rows and the
    len(B) rows before them only subtract, so the table was in its steady state at the
    ancestor and stays there (DESIGN.md, row skipping), and no new node or window concerns B."""
    nb = {b >> 1 for b in B}
    return all((a >> 1) not in nb for a in A2[max(0, len(A2) - (len(B) + d)):])
