#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" [bench args]   (PMC pass: counters only, kernel-trace, no other tracing)
TAG=$1; CTRS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/pmc -o pmc --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-e2e --no-verify "$@" > $OUT/pmc.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out=sys.argv[1]
f=glob.glob(out+'/pmc/*counter_collection.csv')
if not f: print('no counter file', glob.glob(out+'/pmc/*')); sys.exit()
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'].split('(')[0][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    cnt[(k,r['Counter_Name'])]+=1
with open(out+'/pmc_summary.txt','w') as o:
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1].values()))[:60]:
        line=k+' '+' '.join('%s=%.7g(n=%d)'%(c,x,cnt[(k,c)]) for c,x in sorted(v.items()))
        print(line); o.write(line+'\n')
PY
