"""Condense gpurun_out/prof_<tag>_* (scripts/collect_profiles.sh) into profiles/."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
KEYS = ("k_scan", "k_dp_regs<4", "k_dp_regs<8", "k_dp_regs<16", "k_dp_regs<32", "k_dp_long", "k_dp_sys",
        "k_len_sort_block", "k_prep",
        "k_wl_scatter", "k_wl_offsets", "read_u16", "read_b128")


def summ(d):
    out = {}
    files = sorted(glob.glob("gpurun_out/prof_%s_%s/*/*_counter_collection.csv" % (tag, d)),
                   key=os.path.getmtime)[-1:]      # newest run only
    for f in files:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for k in KEYS:
                if k in n:
                    n = k
            if n in KEYS:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for n, c in acc.items():
            out[n] = {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
                      for k, v in c.items()}
    return out


stats = sorted(glob.glob("gpurun_out/prof_%s_stats/*/*_kernel_stats.csv" % tag),
               key=os.path.getmtime)[-1]
shutil.copy(stats, "profiles/%s_bench_config3_kernel_stats.csv" % tag)
pmc = {
    "_how": "scripts/collect_profiles.sh: one rocprofv3 --kernel-trace --pmc pass per counter "
            "group over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (cd /tmp; "
            "TMPDIR=/tmp); means per kernel dispatch, 1x MI355X, workload config3",
    "calibration (tools/fetch_calib, 1 GiB streamed once per kernel)": {
        **summ("calib"),
        "note": "FETCH_SIZE is in KiB and reads 1/2 of the bytes for 2 B/lane and 16 B/lane "
                "coalesced streams (gfx950 rule of MI355X_MICROARCH.md): bytes = FETCH_SIZE*1024*2"},
    "FETCH_SIZE": summ("fetch"), "WRITE_SIZE": summ("write"), "L2": summ("tcc"),
    "SQ_1": summ("sq1"), "SQ_2": summ("sq2"), "GRBM": summ("grbm"),
}
json.dump(pmc, open("profiles/%s_pmc_config3.json" % tag, "w"), indent=1)
fetch = pmc["FETCH_SIZE"]["k_scan"]["FETCH_SIZE"]["mean_per_dispatch"]
write = pmc["WRITE_SIZE"]["k_scan"]["WRITE_SIZE"]["mean_per_dispatch"]
traffic = {
    "workload": "config3", "kernel": "k_scan", "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
    "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
    "method": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py; read side "
              "doubled (gfx950 correction, confirmed by tools/fetch_calib for this kernel's access "
              "widths); Infinity-Cache hits are included in FETCH_SIZE (MI355X_MICROARCH.md), so "
              "this is an upper bound on HBM bytes",
    "source": "profiles/%s_pmc_config3.json" % tag}
json.dump(traffic, open("profiles/traffic_config3.json", "w"), indent=1)
if os.path.exists("gpurun_out/workloads_%s.txt" % tag):
    shutil.copy("gpurun_out/workloads_%s.txt" % tag, "profiles/%s_workloads.txt" % tag)
try:
    line = open("gpurun_out/bench_%s_full.json" % tag).read().strip().splitlines()[-1]
    json.loads(line)
    open("profiles/%s_bench_config3.json" % tag, "w").write(line + "\n")
except Exception as e:
    print("no bench json:", e)
print(json.dumps(traffic, indent=1))
for r in csv.DictReader(open(stats)):
    for k in KEYS:
        if k in r["Name"]:
            print("%-14s calls %3s avg %9.1f us" % (k, r["Calls"], float(r["AverageNs"]) / 1e3))
