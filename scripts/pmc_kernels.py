"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv (newest under dir)."""
import csv, glob, os, sys, collections
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(files[-1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); calls[name] += 1
for name in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
    print("%-30s calls %5d  " % (name, calls[name]) + "  ".join("%s %.4g" % (c, v) for c, v in sorted(acc[name].items())))
