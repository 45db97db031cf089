"""The "real workload" candidate batch of SURVEY.md 8(d): the first P candidate
paths that `gfalign search` itself scores on a synthetic tangle.  Runs the CLI
with GFALIGN_DUMP_BATCHES, times the scorer on that batch and checks a sample
against the oracle.   usage: search_batch.py <config> <P> [budget] [skip]
(skip: leave out the first <skip> candidates, i.e. take the batch from deeper in the search)"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth, build
from gfalign_amd.scorer import Scorer


def load_batches(path, want, skip=0):
    raw = np.fromfile(path, dtype=np.int32)
    at, offs, steps, total = 0, [np.zeros(1, np.int32)], [], 0
    n = 0
    seen = 0
    while at < len(raw) and n < want:
        P, S = int(raw[at]), int(raw[at + 1])
        off = raw[at + 2: at + 3 + P]
        st = raw[at + 3 + P: at + 3 + P + S]
        at += 3 + P + S
        seen += P
        if seen <= skip:
            continue
        take = min(P, want - n)
        offs.append(off[1:take + 1] + total)
        steps.append(st[:off[take]])
        total += int(off[take])
        n += take
    return np.concatenate(offs).astype(np.int32), np.concatenate(steps).astype(np.int32)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    budget = sys.argv[3] if len(sys.argv) > 3 else "20000"
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    t = synth.make(cfg)
    d = "/tmp/sb_" + cfg
    os.makedirs(d, exist_ok=True)
    t.write_gfa(d + "/g.gfa"); t.write_nodelist(d + "/nodes.tsv"); t.write_gaf(d + "/a.gaf")
    cli = build.build_cli()
    dump = d + "/batches.bin"
    if os.path.exists(dump):
        os.remove(dump)
    subprocess.run([cli, "search", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-n", d + "/nodes.tsv",
                    "-s", "utig4-0", "-d", "utig4-%d" % (t.V - 1), "-m", budget],
                   env=dict(os.environ, GFALIGN_DUMP_BATCHES=dump), stdout=subprocess.DEVNULL, check=True)
    off, steps = load_batches(dump, want, skip)
    P = len(off) - 1
    lens = np.diff(off)
    print("search batch: %d candidate paths, length mean %.0f max %d" % (P, lens.mean(), lens.max()))
    with Scorer(t.aln_off, t.aln_steps, t.V) as sc:
        sc.set_profiling(True)
        for _ in range(2):
            bad, good, una = sc.evaluate_paths(off, steps, True)
        sc.set_profiling(False); sc.set_profiling(True)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            sc.evaluate_paths(off, steps, True)
        wall = (time.perf_counter() - t0) / reps
        i = sc.info()
        print("%s search batch P=%d: %.2f ms per batch (%.0f paths/s); scan %.2f ms, sort+dp %.2f ms, dp pairs %d (%.0f per path)" % (
            cfg, P, wall * 1e3, P / wall, i["scan_ms"], i["dp_ms"], i["dp_pairs"], i["dp_pairs"] / P))
    # parity: every path of the batch against all alignments with the fast CPU
    # checker (oracle/gfalign_fast.c, itself pinned to the oracle), and a small
    # sample with the oracle proper
    import oracle
    eb, eg, eu = oracle.fast_evaluate_paths(t.aln_off, t.aln_steps, off, steps, True, threads=16)
    ok = np.array_equal(bad, eb) and np.array_equal(good, eg) and np.array_equal(una, eu)
    print("parity, all %d paths x %d alignments vs oracle/gfalign_fast.c: %s" % (P, t.N, "OK" if ok else "MISMATCH"))
    pick = np.linspace(0, P - 1, 6).astype(int)
    soff = [0]; sst = []
    for p in pick:
        sst.append(steps[off[p]:off[p + 1]]); soff.append(soff[-1] + len(sst[-1]))
    n_sub = min(t.N, 50000)
    a_off = t.aln_off[:n_sub + 1]; a_st = t.aln_steps[:a_off[-1]]
    with Scorer(a_off, a_st, t.V) as sc:
        got = sc.evaluate_paths(np.asarray(soff, np.int32), np.concatenate(sst).astype(np.int32), True)
    exp = oracle.evaluate_paths(a_off, a_st, np.asarray(soff, np.int32), np.concatenate(sst).astype(np.int32), True)
    ok2 = all(np.array_equal(g, e) for g, e in zip(got, exp))
    print("oracle parity on 6 paths x %d alignments: %s" % (n_sub, "OK" if ok2 else "MISMATCH"))
    ok = ok and ok2
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
