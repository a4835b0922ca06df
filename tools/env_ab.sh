#!/bin/bash
# A/B of environment switches on the headline bench in ONE job (same box, back to back): bash tools/env_ab.sh "VAR=val VAR2=val" "..." ...
# ("" = defaults).  Prints samples/s and ms/step per setting, twice each (order A B A B) so that drift shows.
R=$PWD
for rep in 1 2; do
  for setting in "$@"; do
    out=$(env $setting python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 10 2>/dev/null)
    python3 - "$setting" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"[{sys.argv[1] or 'default':40s}] {d['value']:9.1f} samples/s  {d['ms_per_step']:.3f} ms  host {d['host_enqueue_ms_per_step']:.2f} ms", flush=True)
PY
  done
done
