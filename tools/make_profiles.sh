#!/bin/bash
# Reproduces the files under profiles/ on a GPU box:   gpurun -- 'bash tools/make_profiles.sh r01'
# (1) rocprofv3 --kernel-trace --stats over the default bench command; (2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), as
# the micro-arch guide prescribes; (3) tools/pmc_to_traffic.py condenses them.  The program follows `--` directly (no env / bash -c hop).
set -e
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 10 > $OUT/stats.log 2>&1
# PMC passes with every kernel on ONE stream (M3L_WGRAD_INLINE=1): counters of kernels that overlap on two streams are not attributable
export M3L_WGRAD_INLINE=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/mfma.log 2>&1
unset M3L_WGRAD_INLINE
cd $R
python3 tools/pmc_to_traffic.py $OUT $TAG
python3 tools/attn_phase_probe.py $TAG > $OUT/attn_phase.log 2>&1 && cp gpurun_out/${TAG}_attn_phases.json $OUT/final/
# per-kernel-class budget of the step (stand-alone launch times, algorithmic bytes, HBM / MFMA rates): in-library event brackets
python3 tools/kernel_budget.py > $OUT/final/${TAG}_kernel_budget.json 2> $OUT/budget.log || true
# raw traces / counter dumps are tens of MB (gpurun merges at most 64 MiB back): keep only what profiles/ gets
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma
# the bench line quotes the traffic / phase profiles it finds under profiles/: put the fresh ones there first (this tree is the box's copy)
cp $OUT/final/${TAG}_traffic.json $OUT/final/${TAG}_attn_phases.json profiles/ 2>/dev/null || true
python3 bench.py > $OUT/final/${TAG}_bench_n1.json 2> $OUT/bench.log
tail -c 700 $OUT/final/${TAG}_bench_n1.json
