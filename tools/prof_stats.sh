#!/bin/bash
# rocprofv3 --kernel-trace --stats over the bench command; prints the top kernels.  usage: bash tools/prof_stats.sh TAG [bench args...]
TAG=${1:-x}; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" > $OUT/stats.log 2>&1
cd $R
F=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv
[ -n "$KEEP_TRACE" ] || rm -rf $OUT/stats   # the kernel trace is tens of MB: gpurun merges at most 64 MiB back
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f}us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
tail -c 400 $OUT/stats.log
