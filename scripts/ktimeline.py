"""Timeline of the last batch in a rocprofv3 kernel_trace.csv: start offset and
duration of every kernel after the last k_len_hist launch."""
import csv, glob, os, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if ("k_len_sort_block" in r["Kernel_Name"] or "k_len_hist" in r["Kernel_Name"]))
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
    b, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s start %8.1f us  end %8.1f us  dur %7.1f us  queue %s" % (name, b / 1e3, e / 1e3, (e - b) / 1e3, r.get("Queue_Id", "?")))
