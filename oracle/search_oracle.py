"""Python restatement of the reference search loop -- TEST INFRASTRUCTURE.

Follows reference src/eval.cpp:110-193 (dijkstra) line by line, with the
scoring call (src/eval.cpp:162) served by the C oracle.  The pieces the
reference takes from its absent gfalibs submodule follow the conventions of
SURVEY.md Appendix C.3 (uIds in S-line order; adjacency in L-line order with
reverse edges; min-heap on the key with FIFO among equal keys), pinned by
validateFiles/test.6.tst.  Pure-Python loops: small cases only.
"""
import heapq
import itertools

import numpy as np

from . import evaluate_paths


def read_gfa(path):
    headers, ids, links = [], {}, []
    with open(path) as f:
        for line in f:
            cols = line.rstrip("\n").split("\t")
            if cols[0] == "S" and cols[1] not in ids:
                ids[cols[1]] = len(headers)
                headers.append(cols[1])
            elif cols[0] == "L":
                links.append(cols)
    adj = [[] for _ in headers]
    flip = {"+": "-", "-": "+"}
    for l in links:
        if l[1] not in ids or l[3] not in ids:
            continue
        a, b = ids[l[1]], ids[l[3]]
        adj[a].append((l[2], b, l[4]))
        rev = (flip[l[4]], a, flip[l[2]])
        if rev not in adj[b]:
            adj[b].append(rev)
    return headers, ids, adj


def read_gaf_paths(path, ids):
    """GAF -> packed CSR (unknown headers alias to uId 0, src/alignments.cpp:86)."""
    alns = []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")[5]
            steps, i = [], 0
            while i < len(p):
                j = i + 1
                while j < len(p) and p[j] not in "<>":
                    j += 1
                steps.append((ids.get(p[i + 1:j], 0) << 1) | (0 if p[i] == ">" else 1))
                i = j
            alns.append(steps)
    off = np.zeros(len(alns) + 1, np.int32)
    off[1:] = np.cumsum([len(a) for a in alns])
    st = np.concatenate(alns).astype(np.int32) if off[-1] else np.zeros(0, np.int32)
    return off, st


def pack(step):
    sid, o = step
    if o == "+":
        return sid << 1
    if o == "-":
        return (sid << 1) | 1
    return 0x40000000 | (sid << 1)


def search(gfa, node_file, source, destination, gaf=None, max_steps=100000,
           min_nodes=0, return_all_paths=False):
    """Returns the stdout lines of `gfalign search`."""
    headers, ids, adj = read_gfa(gfa)
    aoff, ast = (read_gaf_paths(gaf, ids) if gaf else (np.zeros(1, np.int32), np.zeros(0, np.int32)))
    # include/nodetable.h:16-54
    records, node_count = {}, 0
    with open(node_file) as f:
        for line in f:
            cols = line.rstrip("\n").split("\t")
            count = 1
            if len(cols) > 1:
                count = int(cols[1])
                if count < 1:
                    continue
            node_count += count
            if cols[0] not in ids:
                raise SystemExit("Error: node not in graph (pIUd: %s)" % cols[0])
            records.setdefault(cols[0], [ids[cols[0]], count])
    for name in (source, destination):
        records.setdefault(name, [ids.get(name, 0), 1])
        node_count += 1
    dest_uid = records[destination][0]

    out = []
    counter = itertools.count()
    first = ([(records[source][0], "0")], {k: v[1] for k, v in records.items()})
    heap = [(0, next(counter), first)]
    best_alt, best_uniques, path_counter, steps = 2 ** 31 - 1, 0, 0, 0
    while heap and steps < max_steps:
        _, _, (upath, ubudget) = heapq.heappop(heap)
        last_id, last_or = upath[-1]
        for or0, vid, or1 in adj[last_id]:
            if last_or != "0" and last_or != or0:
                continue
            name = headers[vid]
            if ubudget.get(name, 0) <= 0:
                continue
            npath = list(upath)
            if npath[-1][1] == "0":
                npath[-1] = (npath[-1][0], or0)
            npath.append((vid, or1))
            uniques = len({s[0] for s in npath})
            pst = np.array([pack(s) for s in npath], np.int32)
            bad, good, _ = evaluate_paths(aoff, ast, [0, len(pst)], pst, True)
            bad, good = int(bad[0]), int(good[0])
            alt = bad - good - uniques
            if vid != dest_uid:
                nbudget = dict(ubudget)
                nbudget[name] -= 1
                heapq.heappush(heap, (alt, next(counter), (npath, nbudget)))
            else:
                path_counter += 1
                ham = False
                if len(npath) + 2 == node_count:
                    seen = {}
                    for s in npath:
                        seen[s[0]] = seen.get(s[0], 0) + 1
                    ham = all(seen.get(uid) == cnt for uid, cnt in records.values())
                show = False
                if uniques >= min_nodes and (best_uniques < uniques or
                                             (best_uniques == uniques and best_alt > alt)):
                    best_alt, best_uniques, show = alt, uniques, True
                if return_all_paths or show:
                    out.append("%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s" % (
                        path_counter, bad, good, alt, len(npath), uniques,
                        "T" if ham else "F",
                        ",".join(headers[s[0]] + s[1] for s in npath)))
        steps += 1
    if steps >= max_steps:
        out.append("Reached maximum number of steps (%d)" % steps)
    return out
