"""Print the kernels around the class-B simulation launches of a rocprofv3 --kernel-trace csv (start/end in us relative to the first row shown)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "L2Geom<319>" in r["Kernel_Name"]]
for i in idx[4:6]:
    t0 = int(rows[max(0, i - 8)]["Start_Timestamp"])
    for r in rows[max(0, i - 8): i + 6]:
        print("%9.1f %9.1f  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    print("--")
