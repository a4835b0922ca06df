R=$PWD; OUT=$R/gpurun_out/prof_ref; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export M3L_CFGS_ONLY=${1:-ref-default}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/bench_cfgs.py > $OUT/log.txt 2>&1
cd $R; cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; tail -2 $OUT/log.txt
