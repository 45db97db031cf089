"""Regenerates tests/golden/kernel_cases.json from the CPU oracle.

Run from the repo root:  python tests/golden/make_kernel_cases.py
Inputs are seeded; expected counters come from oracle/ (the C restatement of
reference src/eval.cpp:67-108).  Committed so that the GPU box can check that
the oracle it compiles agrees with the one that ran in the build container.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
from helpers import csr, random_case, walk_case  # noqa: E402


def main():
    cases = []
    specs = [("tiny-alphabet", 11, lambda r: random_case(r, 3, 40, 12, 6, 8)),
             ("walk", 12, lambda r: walk_case(r, 12, 40, 60, 12, 9)),
             ("long-aln", 13, lambda r: walk_case(r, 30, 120, 30, 8, 40)),
             ("single-node", 14, lambda r: random_case(r, 1, 20, 6, 5, 6))]
    for name, seed, gen in specs:
        alns, paths = gen(random.Random(seed))
        aoff, ast = csr(alns)
        poff, pst = csr(paths)
        entry = {"name": name, "seed": seed, "n_nodes": 64,
                 "aln_off": aoff.tolist(), "aln_steps": ast.tolist(),
                 "path_off": poff.tolist(), "path_steps": pst.tolist()}
        for flt in (True, False):
            bad, good, una = oracle.evaluate_paths(aoff, ast, poff, pst, flt)
            entry["filter" if flt else "nofilter"] = {
                "bad": bad.tolist(), "good": good.tolist(), "unaligned": una.tolist()}
        fw, rc = oracle.pair_scores(aoff, ast, paths[0])
        entry["pair_scores_path0"] = {"fw": fw.tolist(), "rc": rc.tolist()}
        cases.append(entry)
    with open(os.path.join(HERE, "kernel_cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
