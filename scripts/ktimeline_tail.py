"""Timeline of the last N kernels in a rocprofv3 kernel_trace.csv."""
import csv, glob, os, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[len(rows) - n - skip: len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
    b, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s start %8.1f us  end %8.1f us  dur %7.1f us  grid %s wg %s" % (name, b / 1e3, e / 1e3, (e - b) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
