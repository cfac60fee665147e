"""GPU idle time between kernels of a rocprofv3 --kernel-trace csv: total and by (previous kernel -> next kernel) pair."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# restrict to the last step: from the last k_sketch_tiles-run start
starts = [i for i, r in enumerate(rows) if "k_expand_tiles" in r["Kernel_Name"]]
lo = starts[-1] if starts else 0
rows = rows[lo:]
end = int(rows[0]["Start_Timestamp"]); t0 = end
gaps = collections.Counter(); cnt = collections.Counter(); busy = 0
prev = "start"
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        k = (prev[:40], r["Kernel_Name"].split("(")[0][:40]); gaps[k] += s - end; cnt[k] += 1
    if e > end:
        busy += e - max(s, end); end = e; prev = r["Kernel_Name"].split("(")[0]
print("window %.1f ms busy %.1f ms idle %.1f ms" % ((end - t0) / 1e6, busy / 1e6, (end - t0 - busy) / 1e6))
for k, v in gaps.most_common(25):
    print("%8.2f ms  n=%4d  %s -> %s" % (v / 1e6, cnt[k], k[0], k[1]))
