"""Per-kernel sums of a rocprofv3 --pmc counter_collection CSV.  usage: python tools/pmc_summary.py <counter_collection.csv>"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:60]:
    print(k + " " + " ".join("%s=%.7g(n=%d)" % (c, x, cnt[(k, c)]) for c, x in sorted(v.items())))
