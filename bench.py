#!/usr/bin/env python3
"""Benchmark of the path-scoring hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3]

One "step" = one pass of the hot path over one batch: every candidate path of
the workload scored against every alignment (reference src/eval.cpp:67-108 once
per candidate), with alignments, candidates and counters resident in HBM when
the timed region starts.  N > 1: one rank per GPU -- started by
torch.distributed.run, or, when WORLD_SIZE is not set, by this script itself
(`python bench.py --gpus N` spawns its N rank processes before anything touches
a GPU) -- the alignments are sharded over the ranks, every rank scores the
whole batch against its shard, and the per-path counters are summed with one
RCCL all-reduce inside the step: the same total work for every N ("strong").

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how `roofline`,
`cpu_baseline` and `search_mode` are defined.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# the same guide: a SIMD issues one wave64 VALU instruction per 2 cycles (4 x SIMD-32 ... 64 lanes in two
# passes), 1024 SIMDs, 2.4 GHz -> 1228.8 G wave-instructions/s chip-wide
VALU_PEAK_GINST = 1024 * 2.4 / 2.0
L2_PEAK_GBS = 34500.0   # aggregate L2 read bandwidth (4 MiB per XCD), same guide


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
            m3 = re.search(r"needed (\d+) of the scored candidates", p.stderr)
            if m1 and m2:
                scored, nb = int(m2.group(1)), int(m2.group(2))
                needed = int(m3.group(1)) if m3 else None
                out[key] = {"command": "gfalign search -m 20000 (default speculation; GAF parse and scorer "
                                       "creation not counted)" + (" GFALIGN_INCREMENTAL=0" if env else ""),
                            "scored_paths": scored, "scored_in_full": int(m2.group(3)), "batches": nb,
                            "search_loop_s": float(m1.group(1)), "candidates_s": float(m1.group(2)),
                            "scoring_s": float(m1.group(3)), "process_wall_s": wall,
                            "paths_per_s": scored / float(m1.group(1)),
                            # the candidates whose scores the search went on to use (extensions of popped
                            # entries = the reference's evaluatePath calls, src/eval.cpp:146-162); the rest
                            # of scored_paths is speculation that was never popped
                            "needed_paths": needed,
                            "needed_paths_per_s": (needed / float(m1.group(1))) if needed else None,
                            "speculation_waste": (1.0 - needed / scored) if needed and scored else None}
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


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes of this very
    script (children, never a re-exec; the parent has not imported torch or touched a GPU),
    one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run
    would.  Rank 0's output is ours; any rank failing fails the run."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies leaves the others waiting at a rendezvous or a collective: end them
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        if procs[0].poll() is not None and all(p.poll() is not None for p in procs[1:]):
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(codes):
        sys.exit("bench.py: rank exit codes %s" % codes)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)      # (0.5 s of timed steps at 0.9 ms each)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search-mode", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)          # (does not return)

    import torch
    import torch.distributed as dist
    from gfalign_amd import shard, synth
    from gfalign_amd.scorer import Scorer, load_library

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
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

    # the timed region: HIP events around the dominant kernel only (level 2: two events per
    # step; the full set of phase events costs a step ~3 %)
    sc.set_profiling(2)
    # (no cyclic collection of the interpreter inside the timed steps: a generation-2 pass is
    # ~40 ms of host time, scripts/small_shard_probe.py)
    gc.collect()
    gc.disable()
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
    gc.enable()
    sc.sync_status()
    info = sc.info()
    # the phases of a step (scan phase, exact DP, whole call): a few more steps with every
    # event on, outside the timed region
    sc.set_profiling(True)
    for _ in range(min(args.steps, 20)):
        step()
    torch.cuda.synchronize(dev)
    phases = sc.info()
    sc.set_profiling(False)
    for k in ("scan_ms", "dp_ms", "total_ms"):
        info[k] = phases[k]

    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    counts = d_counts.cpu().numpy().view(np.uint32)
    bad, good, una = counts[:P], counts[P:2 * P], counts[2 * P:]

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = P * args.steps / elapsed
        # Dominant kernel: k_scan3, the walk over every alignment for every tile of 31
        # candidate paths, timed live with HIP events on the caller's
        # stream around its launches of the timed region (gfal_scorer_set_profiling).
        #
        # Roofline.  SURVEY.md 8(d)'s algorithmic bytes (the alignment set re-read per
        # candidate) divided by the kernel time exceed the HBM peak many times over: a
        # tile of paths shares every load and the set lives in L2 / Infinity Cache, so
        # HBM does not bound the kernel (hbm_* fields, PMC traffic).  What bounds an
        # integer compare-and-count kernel is instruction issue:
        #   achieved = VALU wave-instructions k_scan3 issues per step (SQ_INSTS_VALU, rocprofv3
        #              PMC pass, profiles/issue_<workload>.json) / the live kernel time,
        #   peak     = one wave64 VALU instruction per 2 cycles per SIMD (the guide),
        #   useful   = the same without the instructions of a prologue-only run
        #              (GFAL_DEBUG_SCAN2=1 PMC pass): the item loop's share.
        # The counters are tied to the build they were taken on: another library build
        # id -> the fractions are null rather than stale.
        S_r, N_r = int(info["n_steps"]), int(info["n_aln"])
        alg_bytes = P * (4 * S_r + 4 * (N_r + 1) + 12) + 4 * total_steps
        kernel_ms = info["scan_kernel_ms"] if info["scan_kernel_ms"] > 0 else info["scan_ms"]
        kernel_s = kernel_ms * 1e-3
        alg_gbs = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else None
        lib_id = load_library().gfal_build_id().decode()
        traffic = issue = None
        note_counters = None
        peaks = _load_json("issue_peaks.json")
        if world == 1:
            issue = _load_json("issue_%s.json" % args.workload)
            tr = _load_json("traffic_%s.json" % args.workload)
            if issue and issue.get("build_id") != lib_id:
                note_counters = ("profiles/issue_%s.json was taken on build %s, this library is %s: "
                                 "instruction-issue figures withheld" % (args.workload, issue.get("build_id"), lib_id))
                issue = None
            if tr and tr.get("build_id") == lib_id:
                traffic = tr.get("hbm_bytes_per_launch")
        achieved = frac = useful = frac_useful = None
        if issue and kernel_s > 0:
            achieved = issue["valu_wave_insts_per_step"] / kernel_s / 1e9           # G wave-inst/s, whole chip
            frac = achieved / VALU_PEAK_GINST
            if issue.get("useful_valu_wave_insts_per_step") is not None:
                useful = issue["useful_valu_wave_insts_per_step"] / kernel_s / 1e9
                frac_useful = useful / VALU_PEAK_GINST
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
                "note": "value = the full evaluation of a 10 000-path batch (the reference's evaluatePath per "
                        "candidate); the search itself: search_mode.cli",
                "tile_paths": info["tile_paths"],
                "workgroups": info["n_workgroups"],
                "dp_pairs_per_step": info["dp_pairs"],
            },
            "roofline": {
                "bound": "valu-issue",
                "kernel": "k_scan3 (every alignment against every tile of 31 candidate paths)",
                "achieved": achieved,
                "peak": VALU_PEAK_GINST,
                "unit": "G wave-instructions/s",
                "frac": frac,
                "traffic": traffic,
                "kernel_ms": kernel_ms,
                "useful_achieved": useful,
                "frac_useful": frac_useful,
                "valu_wave_insts_per_step": issue["valu_wave_insts_per_step"] if issue else None,
                "useful_valu_wave_insts_per_step": issue.get("useful_valu_wave_insts_per_step") if issue else None,
                "salu_wave_insts_per_step": issue["salu_wave_insts_per_step"] if issue else None,
                "counters_build_id": issue.get("build_id") if issue else None,
                "library_build_id": lib_id,
                "counters_note": note_counters,
                # the phases of a step (HIP events, 20 further steps after the timed region): window
                # preparation + scan kernels, exact DP, whole call
                "scan_phase_ms": info["scan_ms"],
                "dp_kernel_ms": info["dp_ms"],
                "call_ms": info["total_ms"],
                # measured on this chip (tools/valu_rate.hip): simple two-operand ops / three-operand,
                # compare-to-SGPR, readlane class -- what the guide's 2-cycle figure becomes in practice
                "peak_measured_vop2": (peaks["valu_vop2_ginst_per_s_simd"] * peaks["n_simds"]) if peaks else None,
                "peak_measured_vop3_class": (peaks["valu_vop3_ginst_per_s_simd"] * peaks["n_simds"]) if peaks else None,
                "hbm_algorithmic_bytes_per_launch": alg_bytes,
                "hbm_algorithmic_gbs": alg_gbs,
                "hbm_algorithmic_frac_of_8tbs": (alg_gbs / HBM_PEAK_GBS) if alg_gbs else None,
                "hbm_physical_gbs": (traffic / kernel_s / 1e9) if traffic and kernel_s > 0 else None,
                "hbm_physical_frac_of_8tbs": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic and kernel_s > 0 else None,
                "note": "not HBM-bound: the kernel moves a fraction of a percent of its algorithmic bytes "
                        "(SURVEY.md 8(d): the alignment set re-read per candidate) through HBM -- 31 candidate "
                        "paths share each load, the set sits in L2 / Infinity Cache -- so the algorithmic rate "
                        "exceeds the HBM peak and says nothing about the binding unit, which is instruction "
                        "issue; counters: profiles/ (DESIGN.md section 5)",
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
