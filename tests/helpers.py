"""Shared helpers for the test-suite (fixtures parsing, random cases)."""
import json
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REF_FILES = os.path.join(GOLDEN, "reference_testfiles")


def csr(lists):
    off = np.zeros(len(lists) + 1, np.int32)
    if lists:
        off[1:] = np.cumsum([len(x) for x in lists])
    steps = (np.concatenate([np.asarray(x, np.int32) for x in lists])
             if lists and off[-1] else np.zeros(0, np.int32))
    return off, steps.astype(np.int32)


def parse_gaf_path(text, ids):
    """'>1<4' -> packed steps, headers mapped through `ids` (dict)."""
    out, i = [], 0
    while i < len(text):
        neg = text[i] == "<"
        j = i + 1
        while j < len(text) and text[j] not in "<>":
            j += 1
        out.append((ids[text[i + 1:j]] << 1) | int(neg))
        i = j
    return out


def parse_path_string(text, ids):
    """'1+,2-' -> packed steps."""
    return [(ids[c[:-1]] << 1) | int(c[-1] == "-") for c in text.split(",")]


def load_appendix_c():
    with open(os.path.join(GOLDEN, "survey_appendix_c.json")) as f:
        return json.load(f)


def random3_alignments():
    """The four reads of testFiles/random3.gaf as packed CSR (+ names)."""
    ids = {str(k + 1): k for k in range(5)}
    alns, names = [], []
    with open(os.path.join(REF_FILES, "random3.gaf")) as f:
        for line in f:
            cols = line.rstrip("\n").split("\t")
            names.append(cols[0])
            alns.append(parse_gaf_path(cols[5], ids))
    return ids, names, alns


def random_case(rnd, n_nodes, n_aln, n_paths, max_m, max_n, min_m=1, min_n=1):
    alns = [[(rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
             for _ in range(rnd.randint(min_m, max_m))] for _ in range(n_aln)]
    paths = [[(rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
              for _ in range(rnd.randint(min_n, max_n))] for _ in range(n_paths)]
    return alns, paths


def walk_case(rnd, n_nodes, walk_len, n_aln, n_paths, max_m):
    """Alignments and paths cut from one random walk (many exact hits,
    start-overhangs and near misses)."""
    walk = [(rnd.randrange(n_nodes) << 1) | rnd.randrange(2) for _ in range(walk_len)]
    alns = []
    for _ in range(n_aln):
        m = rnd.randint(1, max_m)
        s = rnd.randrange(0, max(1, walk_len - m + 1))
        b = list(walk[s:s + m])
        r = rnd.random()
        if r < 0.15 and b:
            b[rnd.randrange(len(b))] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        if rnd.random() < 0.5:
            b = [x ^ 1 for x in reversed(b)]
        alns.append(b)
    paths = []
    for _ in range(n_paths):
        s = rnd.randrange(0, walk_len - 1)
        e = rnd.randint(s + 1, walk_len)
        p = list(walk[s:e])
        if rnd.random() < 0.3:
            p[rnd.randrange(len(p))] = (rnd.randrange(n_nodes) << 1) | rnd.randrange(2)
        paths.append(p)
    return alns, paths
