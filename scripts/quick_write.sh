#!/bin/bash
# GPU box: WRITE_SIZE (KiB) and duration of the scan kernels of one bench step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/qw -- python3 bench.py --no-cpu-baseline --no-search-mode --steps 3 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os, re
f = sorted(glob.glob("gpurun_out/qw/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_scan" in r["Kernel_Name"]:
        acc[re.search(r"k_scan2?<[^>]*>", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-60s dispatches %d  WRITE_SIZE %.0f KiB" % (k, len(v), sum(v) / len(v)))
PY
