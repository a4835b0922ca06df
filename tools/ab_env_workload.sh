#!/bin/bash
# A/B of environment settings on a secondary workload: bash tools/ab_env_workload.sh WORKLOAD BATCH "ENV1" "ENV2" ...   ("" = defaults)
W=$1; B=$2; shift 2
for rep in 1 2; do
  for setting in "$@"; do
    out=$(env $setting python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 5 --workload $W --batch $B 2>/dev/null)
    python3 -c "
import json,sys; d=json.loads(sys.argv[2]); print(f'$W B=$B [{sys.argv[1] or \"default\":30s}] {d[\"value\"]:9.1f} samples/s {d[\"ms_per_step\"]:.3f} ms host {d[\"host_enqueue_ms_per_step\"]:.2f}')" "$setting" "$out"
  done
done
