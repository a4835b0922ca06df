#!/bin/bash
# kernel trace of the bench (WORKLOAD=cfg2|cfg4|cfg5|ref, default cfg2) + per-queue timeline of one steady-state step.  usage: bash tools/timeline_run.sh [TAG]
TAG=${1:-tl}; R=$PWD; OUT=$R/gpurun_out/tl_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $R/bench.py --workload ${WORKLOAD:-cfg2} --no-cpu-baseline --no-secondary --steps 12 --warmup 6 > $OUT/log.txt 2>&1
cd $R; F=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $F 3 > $OUT/timeline.txt 2>&1
rm -rf $OUT/tr; tail -c 300 $OUT/log.txt; tail -3 $OUT/timeline.txt
