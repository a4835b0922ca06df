"""headline bench + the in-step average duration of one profiled kernel class (in-library HIP-event brackets).  usage: python tools/prof_stats_lite.py [class]"""
import json, subprocess, sys
cls = sys.argv[1] if len(sys.argv) > 1 else "enc_fwd_mega"
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-secondary", "--steps", "40", "--warmup", "15", "--roofline-kernel", cls], capture_output=True, text=True)
d = json.loads(out.stdout.strip().split("\n")[-1])
r = d.get("roofline") or {}
print(d["value"], d["ms_per_step"], cls, r.get("avg_launch_us"), "alone", (r.get("stand_alone") or {}).get("avg_launch_us"))
