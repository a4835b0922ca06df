for w in "cfg4 64" "cfg4 256" "cfg5 128" "cfg5 512" "ref 512"; do
  set -- $w
  for mr in 0 1 100000000; do
    out=$(M3L_ROWTILE_MIN_ROWS=$mr python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 5 --workload $1 --batch $2 2>/dev/null)
    python3 -c "
import json,sys; d=json.loads(sys.argv[1]); print('$1 B=$2 min_rows=$mr:', d['value'], d['ms_per_step'])" "$out"
  done
done
