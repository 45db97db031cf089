"""Print a rocprofv3 kernel_stats.csv (newest under the given dir) as a short table."""
import csv, glob, os, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in csv.DictReader(open(files[-1])):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%-44s calls %6s  total %10.3f ms  avg %9.3f us  %5s%%" % (
        name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
