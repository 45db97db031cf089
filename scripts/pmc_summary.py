"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch
of each counter, per kernel."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
            name = r["Kernel_Name"]
            for key in ("k_scan", "k_dp", "k_prep", "k_pairs"):
                if key in r["Kernel_Name"]:
                    name = key
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("k_scan", "k_dp", "k_prep"):
            if k in acc:
                print(k, {c: sum(v) / len(v) for c, v in acc[k].items()}, "n=", len(next(iter(acc[k].values()))))
