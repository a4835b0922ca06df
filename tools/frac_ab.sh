#!/bin/bash
# headline bench under environment settings, printing samples/s and the roofline block's in-step / stand-alone fractions:
#   bash tools/frac_ab.sh "" "M3L_WGRAD_LAYERS=4" ...
R=$PWD
for rep in 1 2; do
  for setting in "$@"; do
    out=$(env $setting python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 10 2>/dev/null)
    python3 - "$setting" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]; sa = r.get("stand_alone") or {}
print(f"[{sys.argv[1] or 'default':28s}] {d['value']:9.1f} samples/s  frac {r['frac']:.3f} ({r['avg_launch_us']:.1f} us, {r['algorithmic_bytes_per_launch']/1e6:.0f} MB, {r['launches_per_step']} launches/step)  alone {sa.get('frac')} ({sa.get('avg_launch_us')} us)", flush=True)
PY
  done
done
