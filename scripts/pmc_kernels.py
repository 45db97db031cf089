"""Per-kernel MEANS PER DISPATCH of rocprofv3 --pmc counter_collection.csv files
(newest under each given dir), scan kernels first."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0]))):
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    n = len(next(iter(acc[name].values())))
    print("%-28s n=%d" % (name, n))
    print("   " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(c.items())))
