#!/usr/bin/env python3
"""Benchmark of the path-scoring hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3]

One "step" = one pass of the hot path over one batch: every candidate path of
the workload scored against every alignment (reference src/eval.cpp:67-108 once
per candidate), with alignments, candidates and counters resident in HBM when
the timed region starts.  N > 1 (launched by torch.distributed.run, one rank
per GPU): the alignments are sharded over the ranks, every rank scores the whole
batch against its shard, and the per-path counters are summed with one RCCL
all-reduce inside the step -- the same total work for every N ("strong").

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how `roofline`,
`cpu_baseline` and `search_mode` are defined.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def _load_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def search_mode(t, device):
    """The metric is paths scored per second IN SEARCH MODE: what `gfalign search`
    itself submits, through the blocking C ABI (host buffers in, counters out).
    Runs the CLI once on the workload's tangle (-m 20000, default speculation),
    then replays the candidate batches it scored:
      first_10k / deep_10k  the first 10 000 candidates and candidates 50 001..60 000
                            of the search as one batch each,
      stream                every batch of the search in order, one blocking call each
                            (the search's own call pattern without its host work),
      cli                   the search itself: scored candidates / search-loop seconds.
    """
    import re
    import subprocess
    import tempfile
    from gfalign_amd import build
    from gfalign_amd.scorer import Scorer
    out = {}
    with tempfile.TemporaryDirectory(prefix="gfalign_bench_") as d:
        t.write_gfa(d + "/g.gfa")
        t.write_nodelist(d + "/nodes.tsv")
        t.write_gaf(d + "/a.gaf")
        cli = build.build_cli()
        dump = d + "/batches.bin"
        stdout = {}
        # cli: the search as shipped -- candidates scored from their parents on the device
        # (gfal_group_score_children; same counters, an algorithmic shortcut reported here,
        # next to the headline figure).  cli_full: every candidate evaluated in full
        # (GFALIGN_INCREMENTAL=0); its batches are the ones replayed below.
        for key, env in (("cli", {}), ("cli_full", {"GFALIGN_INCREMENTAL": "0", "GFALIGN_DUMP_BATCHES": dump})):
            t0 = time.perf_counter()
            p = subprocess.run([cli, "search", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-n", d + "/nodes.tsv",
                                "-s", "utig4-0", "-d", "utig4-%d" % (t.V - 1), "-m", "20000", "--verbose",
                                "--device", str(device)],
                               env=dict(os.environ, **env), capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": "gfalign search failed: " + p.stderr[-300:]}
            stdout[key] = p.stdout
            m1 = re.search(r"search ([0-9.]+) s \(candidates ([0-9.]+) s, scoring ([0-9.]+) s", p.stderr)
            m2 = re.search(r"scored (\d+) candidate paths in (\d+) batches, (\d+) of them in full", p.stderr)
            if m1 and m2:
                scored, nb = int(m2.group(1)), int(m2.group(2))
                out[key] = {"command": "gfalign search -m 20000 (default speculation; GAF parse and scorer "
                                       "creation not counted)" + (" GFALIGN_INCREMENTAL=0" if env else ""),
                            "scored_paths": scored, "scored_in_full": int(m2.group(3)), "batches": nb,
                            "search_loop_s": float(m1.group(1)), "candidates_s": float(m1.group(2)),
                            "scoring_s": float(m1.group(3)), "process_wall_s": wall,
                            "paths_per_s": scored / float(m1.group(1))}
        if stdout["cli"] != stdout["cli_full"]:
            sys.exit("PARITY FAILURE: gfalign search prints different rows with and without GFALIGN_INCREMENTAL")
        raw = np.fromfile(dump, dtype=np.int32)
    batches, at = [], 0
    while at < len(raw):
        P, S = int(raw[at]), int(raw[at + 1])
        batches.append((raw[at + 2: at + 3 + P].copy(), raw[at + 3 + P: at + 3 + P + S].copy()))
        at += 3 + P + S

    def merged(skip, want):
        offs, steps, total, n, seen = [np.zeros(1, np.int32)], [], 0, 0, 0
        for off, st in batches:
            P = len(off) - 1
            seen += P
            if seen <= skip or n >= want:
                continue
            take = min(P, want - n)
            offs.append(off[1:take + 1] + total)
            steps.append(st[:off[take]])
            total += int(off[take])
            n += take
        return np.concatenate(offs).astype(np.int32), np.concatenate(steps).astype(np.int32)

    with Scorer(t.aln_off, t.aln_steps, t.V, device=device) as sc:
        for name, skip in (("first_10k", 0), ("deep_10k", 50000)):
            off, st = merged(skip, 10000)
            if len(off) < 2:
                continue
            sc.evaluate_paths(off, st, True)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                sc.evaluate_paths(off, st, True)
            dt = (time.perf_counter() - t0) / reps
            out[name] = {"paths": len(off) - 1, "mean_path_len": float(np.diff(off).mean()),
                         "ms_per_batch": 1e3 * dt, "paths_per_s": (len(off) - 1) / dt}
        for off, st in batches[:8]:
            sc.evaluate_paths(off, st, True)
        t0 = time.perf_counter()
        n = 0
        for off, st in batches:
            sc.evaluate_paths(off, st, True)
            n += len(off) - 1
        dt = time.perf_counter() - t0
        out["stream"] = {"batches": len(batches), "paths": n, "mean_batch": n / max(1, len(batches)),
                         "ms_per_batch": 1e3 * dt / max(1, len(batches)), "paths_per_s": n / dt}
    out["note"] = ("first_10k / deep_10k / stream: blocking gfal_scorer_score (host buffers, PCIe copies and one "
                   "sync per call included) on the candidate batches gfalign search itself scored on this "
                   "workload's tangle, every candidate in full; cli: the search as shipped (children scored "
                   "from their parents, byte-identical rows); cli_full: the same search with that switched off")
    return out


def cpu_baseline(t, budget_pairs=3.2e6):
    """Time the CPU oracle (port of the reference algorithm) on a bounded
    sample of the SAME workload: 16 candidates spread over the length range
    against the first alignments, sized for roughly 10-30 s on one core."""
    import oracle
    n_paths = 16
    n_aln = int(min(t.N, budget_pairs // n_paths))
    order = np.argsort(np.diff(t.path_off), kind="stable")
    pick = order[np.linspace(0, t.P - 1, n_paths).astype(int)]
    paths = [t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick]
    poff = np.zeros(n_paths + 1, np.int32)
    poff[1:] = np.cumsum([len(p) for p in paths])
    pst = np.concatenate(paths).astype(np.int32)
    aoff = t.aln_off[:n_aln + 1]
    ast = t.aln_steps[:aoff[-1]]
    oracle.lib()
    t0 = time.perf_counter()
    bad, good, una = oracle.evaluate_paths(aoff, ast, poff, pst, True)
    dt = time.perf_counter() - t0
    pairs_per_s = n_paths * n_aln / dt
    return {
        "value": pairs_per_s / t.N,           # candidate paths / s on the full N
        "unit": "paths/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d candidates (length quantiles) x first %d alignments of %s, "
                  "%.1f s on 1 core, scaled to N=%d; oracle/gfalign_oracle.c "
                  "(full NW + traceback, fw+rc, filter on)" % (n_paths, n_aln, t.name, dt, t.N),
    }, (pick, n_aln, bad, good, una)


def cpu_fast(t, bad, good, una, n_paths=64):
    """The kernels' own decision rule as a multi-threaded CPU program
    (oracle/gfalign_fast.c, SURVEY.md 8(d) "cpu-fast"): 64 candidates spread over
    the length range against ALL alignments, on one core and on all cores.  Its
    counters are also compared with the HIP counters of the timed run."""
    import oracle
    order = np.argsort(np.diff(t.path_off), kind="stable")
    pick = order[np.linspace(0, t.P - 1, n_paths).astype(int)]
    paths = [t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick]
    poff = np.zeros(n_paths + 1, np.int32)
    poff[1:] = np.cumsum([len(p) for p in paths])
    pst = np.concatenate(paths).astype(np.int32)
    oracle.fast_lib()
    # a GPU box gives one GPU's job a share of 16 host cores
    cores = min(16, len(os.sched_getaffinity(0)))
    res = {}
    for label, thr, sel in (("one_core", 1, np.arange(2, n_paths, 4)), ("all_cores", cores, np.arange(n_paths))):
        sub = [paths[k] for k in sel]
        o = np.zeros(len(sub) + 1, np.int32)
        o[1:] = np.cumsum([len(p) for p in sub])
        st = np.concatenate(sub).astype(np.int32)
        t0 = time.perf_counter()
        b, g, u = oracle.fast_evaluate_paths(t.aln_off, t.aln_steps, o, st, True, threads=thr)
        dt = time.perf_counter() - t0
        res[label] = len(sub) / dt
        if not (np.array_equal(b, bad[pick[sel]]) and np.array_equal(g, good[pick[sel]])
                and np.array_equal(u, una[pick[sel]])):
            sys.exit("PARITY FAILURE: HIP counters differ from oracle/gfalign_fast.c")
    return {"value": res["all_cores"], "unit": "paths/s", "cores": cores,
            "value_one_core": res["one_core"], "kind": "port-fast",
            "sample": "oracle/gfalign_fast.c (filter bitmap, occurrence lists, overhang test, DP only "
                      "where needed; OpenMP over blocks of alignments): %d candidates (length "
                      "quantiles) x all %d alignments of %s; every fourth of them on one core" % (n_paths, t.N, t.name)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search-mode", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gfalign_amd import shard, synth
    from gfalign_amd.scorer import Scorer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "(one rank per GPU)" % args.gpus)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback on the product path)")
    # Rehearsal on a one-GPU box (GFALIGN_BENCH_REHEARSAL=1): every rank uses
    # cuda:0 and the counters are all-reduced over gloo through the host.  The
    # driver's multi-GPU runs never set it: one rank per GPU, RCCL over xGMI.
    rehearsal = os.environ.get("GFALIGN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    t = synth.make(args.workload)
    t.name = args.workload
    # every rank hands the scorer the whole set and keeps its share of the
    # scorer's own sorted order (gfal_scorer_create_sharded): the shards partition
    # the set, balance by construction and do 1/world of the unsharded work each
    sc = Scorer(t.aln_off, t.aln_steps, t.V, device=local_rank, shard=(rank, world))

    P = t.P
    total_steps = int(t.path_off[-1])
    max_len = int(np.diff(t.path_off).max())
    d_off = torch.from_numpy(t.path_off.astype(np.int32)).to(dev)
    d_steps = torch.from_numpy(t.path_steps.astype(np.int32)).to(dev)
    d_counts = torch.zeros(3 * P, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        sc.score_device(d_off.data_ptr(), d_steps.data_ptr(), P, total_steps, max_len,
                        True, d_counts.data_ptr(), stream.cuda_stream)
        if rehearsal and world > 1:
            host = d_counts.cpu()
            shard.all_reduce_counts(host)
            d_counts.copy_(host)
        else:
            shard.all_reduce_counts(d_counts)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    sc.sync_status()

    sc.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    sc.sync_status()
    info = sc.info()
    sc.set_profiling(False)

    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    counts = d_counts.cpu().numpy().view(np.uint32)
    bad, good, una = counts[:P], counts[P:2 * P], counts[2 * P:]

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = P * args.steps / elapsed
        # Dominant kernels: the scan (k_scan2, plus k_scan for the rare alignment
        # lengths), timed live with HIP events on the caller's stream around every
        # launch of the timed region (gfal_scorer_set_profiling).
        #
        # Roofline.  The scan re-uses every alignment load for all paths of a tile
        # out of LDS and the alignment set lives in L2 / Infinity Cache: it moves
        # ~0.5 % of its algorithmic bytes through HBM, so HBM bandwidth does not
        # bound it.  What bounds it is instruction issue on the 1024 SIMDs:
        #   achieved = VALU wave-instructions of one step (SQ_INSTS_VALU of the scan
        #              kernels, rocprofv3 PMC pass committed under profiles/) / the
        #              live scan time,
        #   peak     = the best sustained VALU issue rate measured on this chip by
        #              tools/valu_rate.hip (simple 2-operand integer ops; the kernel's
        #              3-operand / compare / readlane ops issue at ~60 % of that).
        # The algorithmic-bytes figure of SURVEY.md 8(d) and the physical HBM traffic
        # stay in the object as `hbm_*` fields.
        S_r, N_r = int(info["n_steps"]), int(info["n_aln"])
        alg_bytes = P * (4 * S_r + 4 * (N_r + 1) + 12) + 4 * total_steps
        scan_s = info["scan_ms"] * 1e-3
        alg_gbs = alg_bytes / scan_s / 1e9 if scan_s > 0 else None
        traffic = issue = None
        peaks = _load_json("issue_peaks.json")
        if world == 1:
            tr = _load_json("traffic_%s.json" % args.workload)
            traffic = tr.get("hbm_bytes_per_launch") if tr else None
            issue = _load_json("issue_%s.json" % args.workload)
        achieved = peak = frac = None
        if issue and peaks and scan_s > 0:
            achieved = issue["valu_wave_insts_per_step"] / scan_s / 1e9           # G wave-inst/s, whole chip
            peak = peaks["valu_vop2_ginst_per_s_simd"] * peaks["n_simds"]
            frac = achieved / peak
        out = {
            "metric": "candidate paths scored/sec in search mode",
            "value": value,
            "unit": "paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": "%s: synthetic %d-node tangle, %d GAF alignments (S=%d steps), "
                            "%d candidate paths, filter on" % (args.workload, t.V, t.N, t.S, P),
                "parallelism": "alignments sharded over %d GPU(s), all-reduce of int32[3P]" % world,
                "tile_paths": info["tile_paths"],
                "workgroups": info["n_workgroups"],
                "dp_pairs_per_step": info["dp_pairs"],
            },
            "roofline": {
                "bound": "valu-issue",
                "kernel": "k_scan2 (one launch per alignment-length group, side by side) + k_scan (rare lengths): span from the first start to the last end",
                "achieved": achieved,
                "peak": peak,
                "unit": "G wave-instructions/s",
                "frac": frac,
                "traffic": traffic,
                "kernel_ms": info["scan_ms"],
                "dp_kernel_ms": info["dp_ms"],
                "call_ms": info["total_ms"],
                "valu_wave_insts_per_step": issue["valu_wave_insts_per_step"] if issue else None,
                "salu_wave_insts_per_step": issue["salu_wave_insts_per_step"] if issue else None,
                "peak_vop3_class": (peaks["valu_vop3_ginst_per_s_simd"] * peaks["n_simds"]) if peaks else None,
                # the same against the measured rate of a synthetic loop with the scan's own mix
                # (4 VALU : 2 SALU : 1/8 LDS, tools/valu_rate.hip): how close the kernel is to what
                # its instruction mix can issue at all; `frac` above is against simple two-operand ops
                "peak_scan_mix": (peaks["valu_scan_mix_ginst_per_s_simd"] * peaks["n_simds"]) if peaks else None,
                "frac_of_scan_mix_peak": (achieved / (peaks["valu_scan_mix_ginst_per_s_simd"] * peaks["n_simds"]))
                                         if (achieved and peaks) else None,
                "hbm_algorithmic_bytes_per_launch": alg_bytes,
                "hbm_algorithmic_gbs": alg_gbs,
                "hbm_algorithmic_frac_of_8tbs": (alg_gbs / HBM_PEAK_GBS) if alg_gbs else None,
                "hbm_physical_gbs": (traffic / scan_s / 1e9) if traffic and scan_s > 0 else None,
                "note": "not HBM-bound: the scan moves ~0.5 % of its algorithmic bytes (SURVEY.md 8(d): the "
                        "alignment set re-read per candidate) through HBM -- tiles of candidate paths share "
                        "each load from LDS, the set sits in L2 / Infinity Cache -- so the algorithmic rate "
                        "exceeds the HBM peak and says nothing about the binding unit, which is SIMD "
                        "instruction issue; counters and measured peaks: profiles/ (DESIGN.md section 5)",
            },
        }
        out["config"]["counter_checksum"] = int(bad.astype(np.uint64).sum() * 3 +
                                                 good.astype(np.uint64).sum() * 5 +
                                                 una.astype(np.uint64).sum() * 7)
        if world == 1 and not args.no_cpu_baseline:
            base, (pick, n_aln, ebad, egood, euna) = cpu_baseline(t)
            out["cpu_baseline"] = base
            # the same sample doubles as an end-of-run parity check
            with Scorer(t.aln_off[:n_aln + 1], t.aln_steps[:t.aln_off[n_aln]], t.V,
                        device=local_rank) as chk:
                paths = [t.path_steps[t.path_off[k]:t.path_off[k + 1]] for k in pick]
                poff = np.zeros(len(paths) + 1, np.int32)
                poff[1:] = np.cumsum([len(p) for p in paths])
                gb, gg, gu = chk.evaluate_paths(poff, np.concatenate(paths), True)
            if not (np.array_equal(gb, ebad) and np.array_equal(gg, egood)
                    and np.array_equal(gu, euna)):
                sys.exit("PARITY FAILURE: HIP counters differ from the oracle on the sample")
            out["config"]["parity_sample"] = "bit-exact vs oracle on the cpu_baseline sample"
            # an honest CPU competitor next to the reference-faithful baseline; also a
            # second checker, on full-length alignment sets
            out["cpu_fast"] = cpu_fast(t, bad, good, una)
            # the dedup scorer (identical alignments collapsed into weighted lanes): an
            # algorithmic shortcut that changes the byte count, so it is reported
            # here, next to the headline figure and never inside it
            with Scorer(t.aln_off, t.aln_steps, t.V, device=local_rank, dedup=True) as dd:
                d2 = torch.zeros(3 * P, dtype=torch.int32, device=dev)
                for _ in range(2):
                    dd.score_device(d_off.data_ptr(), d_steps.data_ptr(), P, total_steps, max_len,
                                    True, d2.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    dd.score_device(d_off.data_ptr(), d_steps.data_ptr(), P, total_steps, max_len,
                                    True, d2.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t0) / args.steps
                if not torch.equal(d2, d_counts):
                    sys.exit("PARITY FAILURE: the dedup scorer's counters differ")
                out["dedup"] = {"value": P / dt, "unit": "paths/s", "ms_per_step": 1e3 * dt,
                                "resident_alignments": dd.info()["n_lanes"],
                                "note": "gfal_scorer_create_dedup: identical alignments collapsed into "
                                        "weighted lanes, same counters; reported separately"}
            out["config"]["parity_sample"] += " and vs oracle/gfalign_fast.c on 64 paths x all alignments"
        if world == 1 and not args.no_search_mode:
            out["search_mode"] = search_mode(t, local_rank)
        print(json.dumps(out))
    sc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
