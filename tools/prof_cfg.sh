#!/bin/bash
# rocprofv3 kernel stats of ONE secondary workload of tools/bench_cfgs.py.  usage: bash tools/prof_cfg.sh cfg4|cfg5|ref-default [TAG]
W=${1:-cfg4}; TAG=${2:-$W}
R=$PWD; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export M3L_CFGS_ONLY=$W
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/bench_cfgs.py > $OUT/log.txt 2>&1
cd $R; cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
[ -n "$KEEP_TRACE" ] || rm -rf $OUT/stats
grep "samples/s" $OUT/log.txt
python3 tools/show_stats.py $OUT/kernel_stats.csv 40 2>/dev/null || head -40 $OUT/kernel_stats.csv
