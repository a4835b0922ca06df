#!/usr/bin/env python3
"""Condense rocprofv3 outputs (tools/make_profiles.sh) into the small files kept under profiles/:
   <tag>_kernel_stats.csv   the --stats table as rocprofv3 wrote it
   <tag>_pmc_hbm.csv        per kernel: launches, mean FETCH_SIZE / WRITE_SIZE (KB, raw counters)
   <tag>_traffic.json       per bench.py kernel class: HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests of 16-byte-per-lane streams are
tallied as 64 bytes); both counters are in KB.  This is L2-miss traffic and includes Infinity-Cache hits."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "final")
os.makedirs(dst, exist_ok=True)
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))


# The weight-gradient kernel is launched in two populations: GROUPED launches (the four weights of 2-4 transformer layers: one workgroup
# per CU x 768 threads = 196 608 threads at cfg 2; the class bench.py brackets and prices in `roofline`) and single small Linears (patch embed / heads: <= 96
# workgroups).  They are kept apart by grid size so that counter bytes and algorithmic bytes describe the SAME launches.
GROUPED_MIN_THREADS = 150000


def collect(sub, counter, grid_min=0, grid_max=1 << 62):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            if not (grid_min <= int(float(r.get("Grid_Size", 0) or 0)) < grid_max):
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
names = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1]))
tot = sum((fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1]) for k in names) or 1.0
with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "launches", "mean_FETCH_SIZE_KB_raw", "mean_WRITE_SIZE_KB_raw", "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024", "share_of_total"])
    for k in names:
        nf, sf = fetch.get(k, [0, 0.0])
        nw, sw = write.get(k, [0, 0.0])
        w.writerow([k[:120], max(nf, nw, 1), round(sf / max(nf, 1), 1), round(sw / max(nw, 1), 1),
                    int((2 * sf / max(nf, 1) + sw / max(nw, 1)) * 1024), round((2 * sf + sw) / tot, 4)])

# MFMA utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
# summed over the 8 XCDs)
busy, active = collect("mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), collect("mfma", "GRBM_GUI_ACTIVE")
if busy and active:
    with open(os.path.join(dst, f"{tag}_mfma_util.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "launches", "avg_kernel_cycles(GRBM_GUI_ACTIVE/8)", "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES/(cycles*1024 SIMDs)"])
        for k in sorted(busy, key=lambda k: -busy[k][1]):
            n, sb = busy[k]
            na, sa = active.get(k, [0, 0.0])
            if not na or sa <= 0:
                continue
            cyc = sa / na / 8.0
            w.writerow([k[:110], n, int(cyc), round((sb / n) / (cyc * 1024.0), 4)])

classes = {"wgrad": "wgrad_kernel", "gemm_nt_glds64": "gemm_nt_glds_kernel", "ln_bwd": "ln_bwd_kernel", "mlp_t192_bwd": "mlp_t192_bwd_kernel",
           "mlp_t192_fwd": "mlp_t192_fwd_kernel"}
# steps covered by a PMC pass = launches of the once-per-step optimizer kernel
steps_f = sum(v[0] for k, v in fetch.items() if "adam_flat" in k) or 1
steps_w = sum(v[0] for k, v in write.items() if "adam_flat" in k) or 1
step_bytes = int((2 * sum(v[1] for v in fetch.values()) / steps_f + sum(v[1] for v in write.values()) / steps_w) * 1024)
traffic = {"_step": {"hbm_bytes_per_step": step_bytes, "steps_in_fetch_pass": steps_f, "steps_in_write_pass": steps_w,
                     "note": "sum over every kernel of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per optimizer step; L2-miss traffic incl. Infinity-Cache hits"}}


def per_launch(sub):
    nf = sum(v[0] for k, v in fetch.items() if sub in k)
    sf = sum(v[1] for k, v in fetch.items() if sub in k)
    nw = sum(v[0] for k, v in write.items() if sub in k)
    sw = sum(v[1] for k, v in write.items() if sub in k)
    return (nf, sf, nw, sw)


# weight gradients, like for like: per GROUPED launch (= bench.py's `roofline` class) and per step over the whole family (grouped + small
# launches + their reduce kernels: the split-M slabs are written by the former and read by the latter)
fg, wg = collect("fetch", "FETCH_SIZE", GROUPED_MIN_THREADS), collect("write", "WRITE_SIZE", GROUPED_MIN_THREADS)
nfg = sum(v[0] for k, v in fg.items() if "wgrad_kernel" in k)
sfg = sum(v[1] for k, v in fg.items() if "wgrad_kernel" in k)
nwg = sum(v[0] for k, v in wg.items() if "wgrad_kernel" in k)
swg = sum(v[1] for k, v in wg.items() if "wgrad_kernel" in k)
fam_f = sum(v[1] for k, v in fetch.items() if "wgrad_" in k) / steps_f
fam_w = sum(v[1] for k, v in write.items() if "wgrad_" in k) / steps_w
if nfg and nwg:
    traffic["wgrad_grouped"] = {"hbm_bytes_per_launch": int((2 * sfg / nfg + swg / nwg) * 1024), "fetch_size_kb": round(sfg / nfg, 1),
                                "write_size_kb": round(swg / nwg, 1), "launches": nfg, "launches_per_step": round(nfg / steps_f, 2),
                                "note": "wgrad_kernel<*> launches of >= %d threads only (the grouped per-layer launches bench.py brackets as `wgrad`)" % GROUPED_MIN_THREADS}
    traffic["wgrad_family_per_step"] = {"hbm_bytes_per_step": int((2 * fam_f + fam_w) * 1024), "write_bytes_per_step": int(fam_w * 1024),
                                        "note": "every wgrad_kernel<*> and wgrad_reduce_kernel<*> launch of a step (grouped and small): operands read, "
                                                "split-M slabs written and read back, dW written"}

for cls, sub in classes.items():
    if cls == "wgrad":
        continue
    nf, sf, nw, sw = per_launch(sub)
    if nf and nw:
        traffic[cls] = {"hbm_bytes_per_launch": int((2 * sf / nf + sw / nw) * 1024), "fetch_size_kb": round(sf / nf, 1), "write_size_kb": round(sw / nw, 1),
                        "launches": nf, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 2 "
                                                "--no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md, KB -> bytes x1024; L2-miss traffic incl. Infinity-Cache hits"}
# the three largest kernels of the --kernel-trace --stats pass with their own HBM rates (PMC bytes per launch / average duration)
if st:
    rows = list(csv.DictReader(open(st[0])))
    ktot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    top = []
    for r in rows[:3]:
        name = r["Name"]
        key = name.split("(")[0].split("<")[0].split("::")[-1].strip()
        key = next((sub for sub in ("enc_fwd_mega_kernel", "attn_t192_bwd_kernel", "attn_t192_fwd_kernel", "qkv_bwd_t192_kernel",
                                    "wgrad_reduce_kernel", "wgrad_kernel", "ln_bwd_kernel", "mlp_block_bwd_kernel", "mlp_block_fwd_kernel", "attn_block_bwd_kernel",
                                    "attn_block_fwd_kernel", "mlp_t192_bwd_kernel", "mlp_t192_fwd_kernel", "gemm_nt_glds_kernel") if sub in name), key)
        nf, sf, nw, sw = per_launch(key)
        avg_us = float(r["AverageNs"]) / 1e3
        b = int((2 * sf / nf + sw / nw) * 1024) if nf and nw else None
        top.append({"kernel": key, "share_of_kernel_time": round(float(r["TotalDurationNs"]) / ktot, 4), "avg_us": round(avg_us, 1),
                    "hbm_bytes_per_launch": b, "hbm_frac_of_8TBps": round(b / (avg_us * 1e-6) / 8e12, 4) if b else None})
    traffic["_step"]["top_kernels"] = top
try:      # what code these counters were collected on (bench.py shows it beside every figure it quotes from this file)
    traffic["_meta"] = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "m3l_amd", "lib", "build_stamp.json")))
except (OSError, ValueError):
    traffic["_meta"] = {}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v.get("hbm_bytes_per_launch", v.get("hbm_bytes_per_step")) if isinstance(v, dict) else v) for k, v in traffic.items()}))
