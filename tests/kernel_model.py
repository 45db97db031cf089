"""Pure-Python model of the decision rule the HIP kernels implement.

Test-only.  Mirrors gfalign_amd/csrc/scorer.hip (eval_item + k_dp) step for
step so the *design* can be fuzzed against the oracle on a machine without a
GPU: subpath search over the occurrences of B[0]'s node, the filter, the
"m > n" rule, the overhang triage and the exact DP with traceback-exit
propagation.
"""


def traceback_score(A, B):
    """scorer.hip traceback_score(): NW fill + exit propagation, one row."""
    n, m = len(A), len(B)
    dp = [(-j if j <= n else 0) for j in range(m + 1)]
    ex = list(dp)
    for i in range(1, n + 1):
        diag_dp, diag_x = dp[0], ex[0]
        dp[0], ex[0] = 0, 0
        left_dp, left_x = 0, 0
        for j in range(1, m + 1):
            up_dp, up_x = dp[j], ex[j]
            sub = 0 if A[i - 1] == B[j - 1] else -1
            d = diag_dp + sub
            u = up_dp + (-1 if j < m else 0)
            l = left_dp - 1
            v = max(d, u, l)
            if v == d:
                x = diag_x
            elif up_dp >= left_dp:
                x = up_x
            else:
                x = left_x
            dp[j], ex[j] = v, x
            diag_dp, diag_x = up_dp, up_x
            left_dp, left_x = v, x
    return dp[m] - ex[m]


def traceback_score_skip(A, B, window=None):
    """scorer.hip traceback_score_skip(): the same fill, computing only the
    first `window` rows and the rows r .. r + window after every row r whose
    path step has a node that occurs in B (window = m by default; the kernel
    uses its column count MC >= m, any larger window is exact too).  All other
    rows leave the state unchanged: after j rows without a matching step the
    columns 1..j hold dp = -j with exit value 0 and the last column keeps its
    value.  Returns (score, rows computed).  Needs m <= n (the kernels only
    run the DP then)."""
    n, m = len(A), len(B)
    assert m <= n
    w = m if window is None else window
    assert w >= m
    nodes = {b >> 1 for b in B}
    dp = [-j for j in range(m + 1)]
    ex = list(dp)
    remaining, rows = w, 0
    for i in range(1, n + 1):
        if A[i - 1] is not None and (A[i - 1] >> 1) in nodes:
            remaining = w + 1
        if remaining == 0:
            continue
        remaining -= 1
        rows += 1
        diag_dp, diag_x = 0, 0          # column 0 is 0 in every row
        left_dp, left_x = 0, 0
        for j in range(1, m + 1):
            up_dp, up_x = dp[j], ex[j]
            sub = 0 if A[i - 1] == B[j - 1] else -1
            d = diag_dp + sub
            u = up_dp + (-1 if j < m else 0)
            l = left_dp - 1
            v = max(d, u, l)
            if v == d:
                x = diag_x
            elif up_dp >= left_dp:
                x = up_x
            else:
                x = left_x
            dp[j], ex[j] = v, x
            diag_dp, diag_x = up_dp, up_x
            left_dp, left_x = v, x
    return dp[m] - ex[m], rows


def has_overhang(A, B):
    m = len(B)
    for ln in range(1, min(m - 1, len(A)) + 1):
        if A[:ln] == B[m - ln:]:
            return True
    return False


def rc(B):
    return [s ^ 1 for s in reversed(B)]


def decide(A, B, filter):
    """'good' | 'bad' | 'filtered' for packed path A and packed alignment B.

    Path steps may be None (orientation that equals nothing); nodeA gives the
    node ids for the filter.
    """
    steps, nodes = A
    n, m = len(steps), len(B)
    if m == 0:
        return "good"
    occ = {}
    for i, v in enumerate(nodes):
        occ.setdefault(v, []).append(i)
    b0 = B[0]
    found = False
    for pos in occ.get(b0 >> 1, []):
        a = steps[pos]
        if a is None:
            continue
        d = (a & 1) ^ (b0 & 1)
        if not d:
            if pos + m <= n and all(steps[pos + t] == B[t] for t in range(m)):
                found = True
        else:
            if pos >= m - 1 and all(steps[pos - t] == (B[t] ^ 1) for t in range(m)):
                found = True
        if found:
            break
    if found:
        return "good"
    if filter and any((s >> 1) not in occ for s in B):
        return "filtered"
    if m > n:
        return "good"
    a0 = steps[0]
    cand = any(B[t] == a0 for t in range(1, m)) or any((B[t] ^ 1) == a0 for t in range(m - 1))
    if not cand:
        return "bad"
    for Bo in (B, rc(B)):
        if has_overhang(steps, Bo) and traceback_score(steps, Bo) == 0:
            return "good"
    return "bad"


def evaluate(paths, alns, filter):
    out = []
    for A in paths:
        bad = good = una = 0
        nodes = set(A[1])
        for B in alns:
            if filter:
                una += sum(1 for s in B if (s >> 1) not in nodes)
            r = decide(A, B, filter)
            if r == "good":
                good += 1
            elif r == "bad":
                bad += 1
        out.append((bad, good, una))
    return out
