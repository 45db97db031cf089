"""Condense gpurun_out/prof_<tag>_* (scripts/collect_profiles.sh) into profiles/."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
try:
    build_id = open("gpurun_out/build_id_%s.txt" % tag).read().strip()
except OSError:
    build_id = None
# longest key first: "k_scan3" / "k_scan2" before "k_scan"
KEYS = ("k_scan3", "k_scan2", "k_scan", "k_tile_masks", "k_tile", "k_overhang", "k_dp_regs<4", "k_dp_regs<8", "k_dp_regs<16", "k_dp_regs<32", "k_dp_long",
        "k_dp_sys", "k_len_sort_block", "k_prep", "k_wl_scatter", "k_wl_offsets", "k_unpermute",
        "read_u16", "read_b128")


def kernel_key(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def summ(d):
    out = {}
    files = sorted(glob.glob("gpurun_out/prof_%s_%s/*/*_counter_collection.csv" % (tag, d)),
                   key=os.path.getmtime)[-1:]      # newest run only
    for f in files:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = kernel_key(r["Kernel_Name"])
            if n:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        # a bench step launches k_prep once, k_scan2 once per alignment-length group that
        # occurs and k_scan once: per-step figures are sums over a step's dispatches
        steps = max([len(v) for v in acc.get("k_prep", {}).values()] + [1])
        for n, c in acc.items():
            out[n] = {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v),
                          "per_step": sum(v) / steps}
                      for k, v in c.items()}
    return out


stats = sorted(glob.glob("gpurun_out/prof_%s_stats/*/*_kernel_stats.csv" % tag),
               key=os.path.getmtime)[-1]
shutil.copy(stats, "profiles/%s_bench_config3_kernel_stats.csv" % tag)
pmc = {
    "_how": "scripts/collect_profiles.sh: one rocprofv3 --kernel-trace --pmc pass per counter "
            "group over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-search-mode` "
            "(cd /tmp; TMPDIR=/tmp); means per kernel dispatch, 1x MI355X, workload config3",
    "calibration (tools/fetch_calib, 1 GiB streamed once per kernel)": {
        **summ("calib"),
        "note": "FETCH_SIZE is in KiB and reads 1/2 of the bytes for 2 B/lane and 16 B/lane "
                "coalesced streams (gfx950 rule of MI355X_MICROARCH.md): bytes = FETCH_SIZE*1024*2"},
    "FETCH_SIZE": summ("fetch"), "WRITE_SIZE": summ("write"), "L2": summ("tcc"),
    "SQ_1": summ("sq1"), "SQ_2": summ("sq2"), "GRBM": summ("grbm"),
    "SQ_1 prologue only (GFAL_DEBUG_SCAN2=1)": summ("sq1p"),
    "build_id": build_id,
}
json.dump(pmc, open("profiles/%s_pmc_config3.json" % tag, "w"), indent=1)


def scan_span():
    """The scan kernels of one step run side by side (k_scan2 once per alignment-length
    group on its own stream, k_scan): what compares with bench.py's live `kernel_ms` is
    the span from the first start to the last end, per step, from the kernel trace of
    the --stats run; the per-kernel durations in the stats table overlap."""
    tr = sorted(glob.glob("gpurun_out/prof_%s_stats/*/*_kernel_trace.csv" % tag), key=os.path.getmtime)[-1]
    rows = sorted((r for r in csv.DictReader(open(tr))), key=lambda r: int(r["Start_Timestamp"]))
    spans, busy, cur = [], [], None
    for r in rows:
        k = kernel_key(r["Kernel_Name"])
        if k == "k_prep":
            cur = [None, None, 0]
        elif k in ("k_scan3",) and cur is not None:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            cur[0] = s if cur[0] is None else min(cur[0], s)
            cur[1] = e if cur[1] is None else max(cur[1], e)
            cur[2] += e - s
        elif k == "k_wl_offsets" and cur is not None and cur[0] is not None:
            spans.append(cur[1] - cur[0])
            busy.append(cur[2])
            cur = None
    return {"steps": len(spans), "scan_span_us_mean": sum(spans) / len(spans) / 1e3,
            "scan_span_us_min": min(spans) / 1e3, "scan_span_us_max": max(spans) / 1e3,
            "sum_of_kernel_durations_us_mean": sum(busy) / len(busy) / 1e3,
            "source": "kernel trace of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 "
                      "--no-cpu-baseline --no-search-mode` (scripts/collect_profiles.sh)"}


span = scan_span()
json.dump(span, open("profiles/%s_scan_span.json" % tag, "w"), indent=1)
print(json.dumps(span, indent=1))


def per_step(group, counter, kernels=("k_scan3",)):
    """sum over the dispatches of one bench step of the dominant kernel (k_scan3)"""
    tot = 0.0
    for k in kernels:
        tot += pmc[group].get(k, {}).get(counter, {}).get("per_step", 0.0)
    return tot


fetch, write = per_step("FETCH_SIZE", "FETCH_SIZE"), per_step("WRITE_SIZE", "WRITE_SIZE")
traffic = {
    "workload": "config3", "kernels": "k_scan3", "build_id": build_id,
    "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
    "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
    "method": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py; read side "
              "doubled (gfx950 correction, confirmed by tools/fetch_calib for this kernel's access "
              "widths); Infinity-Cache hits are included in FETCH_SIZE (MI355X_MICROARCH.md), so "
              "this is an upper bound on HBM bytes",
    "source": "profiles/%s_pmc_config3.json" % tag}
json.dump(traffic, open("profiles/traffic_config3.json", "w"), indent=1)
issue = {
    "workload": "config3", "kernels": "k_scan3", "build_id": build_id,
    "valu_wave_insts_per_step": per_step("SQ_1", "SQ_INSTS_VALU"),
    "prologue_valu_wave_insts_per_step": per_step("SQ_1 prologue only (GFAL_DEBUG_SCAN2=1)", "SQ_INSTS_VALU"),
    "useful_valu_wave_insts_per_step": per_step("SQ_1", "SQ_INSTS_VALU") -
                                       per_step("SQ_1 prologue only (GFAL_DEBUG_SCAN2=1)", "SQ_INSTS_VALU"),
    "step_kernels_valu_wave_insts": {k: pmc["SQ_1"].get(k, {}).get("SQ_INSTS_VALU", {}).get("per_step")
                                     for k in ("k_scan3", "k_tile", "k_tile_masks", "k_overhang", "k_scan", "k_prep")},
    "salu_wave_insts_per_step": per_step("SQ_1", "SQ_INSTS_SALU"),
    "lds_wave_insts_per_step": per_step("SQ_1", "SQ_INSTS_LDS"),
    "vmem_rd_wave_insts_per_step": per_step("SQ_1", "SQ_INSTS_VMEM_RD"),
    "wave_quad_cycles_per_step": per_step("SQ_1", "SQ_WAVE_CYCLES"),
    "wait_any_quad_cycles_per_step": per_step("SQ_2", "SQ_WAIT_ANY"),
    "source": "profiles/%s_pmc_config3.json (SQ_INSTS_* count wave-instructions)" % tag}
json.dump(issue, open("profiles/issue_config3.json", "w"), indent=1)

# measured issue peaks (tools/valu_rate.hip)
vr = "gpurun_out/valu_rate_%s.jsonl" % tag
if os.path.exists(vr):
    shutil.copy(vr, "profiles/%s_valu_rate.jsonl" % tag)
    rows = [json.loads(l) for l in open(vr) if l.strip()]

    def best(kind, key):
        return max(r[key] for r in rows if r["kind"] == kind and key in r)
    peaks = {
        "n_simds": 1024, "n_cus": 256,
        "valu_vop2_ginst_per_s_simd": max(best("v_and_b32", "valu_ginst_per_s_simd"),
                                          best("v_add_u32", "valu_ginst_per_s_simd")),
        "valu_vop3_ginst_per_s_simd": max(best("v_alignbit_b32", "valu_ginst_per_s_simd"),
                                          best("v_bfe_u32", "valu_ginst_per_s_simd"),
                                          best("v_cmp_eq_u32->sgpr", "valu_ginst_per_s_simd")),
        "valu_scan_mix_ginst_per_s_simd": best("mix scan(4v:2s:ds/8)", "valu_ginst_per_s_simd"),
        "salu_ginst_per_s_cu": best("salu(and_b64,bcnt1,add)", "salu_ginst_per_s_cu"),
        "lds_b32_ginst_per_s_cu": best("ds_read_b32 x8", "lds_ginst_per_s_cu"),
        "source": "profiles/%s_valu_rate.jsonl (tools/valu_rate.hip: best sustained rate over 1..8 "
                  "waves per SIMD, wave-instructions per second over the span of the launch)" % tag}
    json.dump(peaks, open("profiles/issue_peaks.json", "w"), indent=1)
    print(json.dumps(peaks, indent=1))
if os.path.exists("gpurun_out/workloads_%s.txt" % tag):
    shutil.copy("gpurun_out/workloads_%s.txt" % tag, "profiles/%s_workloads.txt" % tag)
try:
    line = open("gpurun_out/bench_%s_full.json" % tag).read().strip().splitlines()[-1]
    json.loads(line)
    open("profiles/%s_bench_config3.json" % tag, "w").write(line + "\n")
except Exception as e:
    print("no bench json:", e)
print(json.dumps(traffic, indent=1))
print(json.dumps(issue, indent=1))
for r in csv.DictReader(open(stats)):
    k = kernel_key(r["Name"])
    if k:
        print("%-14s calls %3s avg %9.1f us" % (k, r["Calls"], float(r["AverageNs"]) / 1e3))
