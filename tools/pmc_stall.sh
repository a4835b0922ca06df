#!/bin/bash
# Where the waves of each kernel spend their cycles: SQ wait / issue counters per kernel (one PMC pass, weight gradients inline so that
# counters are attributable).  usage: bash tools/pmc_stall.sh TAG
TAG=${1:-x}
R=$PWD; OUT=$R/gpurun_out/stall_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export M3L_WGRAD_INLINE=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ${PMC_EXTRA:-SQ_INST_CYCLES_VMEM} SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/log.txt 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, re
fs = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)(<[^>]*>|I[\w]*?E(?=v|E))?", k)
        k = (m.group(1) + (m.group(2) or "")) if m else k[:60]
        k = k[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
rows = []
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    rows.append((wc, k, d))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'wave_cyc(M)':>11s} wait_any wait_inst active  valu   lds  ${PMC_EXTRA:-vmem}  mfma_busy/wave_cyc")
for wc, k, d in rows[:22]:
    g = lambda c: d.get(c, 0) / wc
    print(f"{k:70s} {wc/1e6:11.1f} {g('SQ_WAIT_ANY'):8.2f} {g('SQ_WAIT_INST_ANY'):9.2f} {g('SQ_ACTIVE_INST_ANY'):6.2f} {g('SQ_ACTIVE_INST_VALU'):5.2f} {g('SQ_ACTIVE_INST_LDS'):5.2f} {g('${PMC_EXTRA:-SQ_INST_CYCLES_VMEM}'):5.2f} {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/wc:6.3f}")
PY
rm -rf $OUT/p
tail -3 $OUT/log.txt
