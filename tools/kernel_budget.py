"""Per-kernel-class budget of one cfg-2 training step (B = 256, bf16): every tracked launch bracketed by HIP events inside the library
(m3l_prof_begin(NULL, 1)), weight gradients on the compute stream so that every kernel has the GPU to itself.  Per class: launches per
step, average launch time (event-bracket overhead subtracted), algorithmic bytes per launch as the launcher declares them, the HBM rate
and MFMA rate that follow, and the class's share of the step.  -> JSON on stdout (tools/make_profiles.sh keeps it as
profiles/rNN_kernel_budget.json).  usage: python tools/kernel_budget.py [workload]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["M3L_WGRAD_INLINE"] = "1"
import torch  # noqa: E402
import bench  # noqa: E402
from m3l_amd import _lib  # noqa: E402
from m3l_amd.parallel import FlatAdam, GradSync  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
c, B, name = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
lib = _lib.lib()
mae = bench.build_model(c, "bf16", dev)
sync = GradSync(mae)
opt = FlatAdam(sync, lr=1e-4)
torch.manual_seed(1234)
x = bench.synthetic_batch(c, B, dev)


def step():
    sync.zero_grad()
    mae(x).backward()
    sync.finish(defer_scale=True)
    opt.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
STEPS = 20
lib.m3l_prof_begin(None, 1)
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
lib.m3l_prof_end()
ov = lib.m3l_prof_event_overhead_us(torch.cuda.current_stream().cuda_stream, 200)
kinds = {}
for i in range(lib.m3l_prof_count()):
    nm = C.create_string_buffer(96)
    ms_tot, launches, work, byt = C.c_double(), C.c_long(), C.c_double(), C.c_double()
    lib.m3l_prof_get(i, nm, 96, C.byref(ms_tot), C.byref(launches), C.byref(work), C.byref(byt))
    if launches.value:
        k = kinds.setdefault(nm.value.decode().split("[")[0], [0.0, 0, 0.0, 0.0])
        k[0] += ms_tot.value; k[1] += launches.value; k[2] += work.value; k[3] += byt.value
rows, tot = [], 0.0
for k, (ms, n, work, byt) in kinds.items():
    us = max(ms / n * 1e3 - ov, 0.1)
    per_step = us * n / STEPS
    tot += per_step
    rows.append({"class": k, "launches_per_step": round(n / STEPS, 2), "avg_us": round(us, 2), "us_per_step": round(per_step, 1),
                 "algorithmic_MB_per_launch": round(byt / n / 1e6, 1) if byt else None,
                 "TBps": round(byt / n / (us * 1e-6) / 1e12, 2) if byt else None,
                 "frac_of_8TBps": round(byt / n / (us * 1e-6) / 8e12, 3) if byt else None,
                 "mfma_TFLOPs": round(work / n / (us * 1e-6) / 1e12, 1) if work else None})
rows.sort(key=lambda r: -r["us_per_step"])
for r in rows:
    r["share_of_tracked"] = round(r["us_per_step"] / tot, 4)
meta = {}
try:
    meta = json.load(open(os.path.join(ROOT, "m3l_amd", "lib", "build_stamp.json")))
except (OSError, ValueError):
    pass
print(json.dumps({"workload": name, "batch": B, "steps": STEPS, "event_bracket_overhead_us": round(ov, 2), "tracked_us_per_step": round(tot, 1),
                  "note": "stand-alone launch times (weight gradients inline); kernels the library does not bracket (torch's rand / fill, "
                          "the Adam launch) are not in the list", "classes": rows, "_meta": meta}, indent=1))
