"""Timeline of one steady-state step from a rocprofv3 kernel trace: per queue, each kernel's start (us from the step's first kernel),
duration and the idle gap before it.  usage: python tools/step_timeline.py <kernel_trace.csv> [step_index_from_end=3]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the mask_rank kernel that follows an adam kernel
starts = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
a, b = starts[-back - 1] + 1, starts[-back] + 1
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
last_end = {}
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(\w+_kernel)", n)
    return (m.group(1) if m else n)[:28]
print(f"step of {len(seg)} kernels, {(int(seg[-1]['End_Timestamp']) - t0) / 1000:.1f} us")
busy = {}
for r in seg:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1000 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    busy[q] = busy.get(q, 0) + (e - s) / 1000
    print(f"q{q:>2} {(s - t0) / 1000:8.1f} {(e - s) / 1000:7.1f} gap {gap:6.1f}  {short(r['Kernel_Name'])}")
print("busy us per queue:", {k: round(v, 1) for k, v in busy.items()})
