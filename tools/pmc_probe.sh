#!/bin/bash
# rocprofv3 PMC pass over a probe script; prints per-kernel counter sums.  usage: bash tools/pmc_probe.sh TAG "COUNTERS..." script.py [args]
TAG=$1; CNT=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -o p -- python3 $R/"$@" > $OUT/log.txt 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:34s} {v / n[(k, c)]:16.1f} per dispatch")
PY
