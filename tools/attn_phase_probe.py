"""Per-phase cycle counts of the forward attention block (attn_block.hip) at the cfg-2 encoder shape (B = 256, n = 48, D = 192, 3 heads)
from the kernel's own shader-clock stamps (m3l_set_attn_phase_buffer), with the MFMA utilisation of each phase:
   MFMA FLOP issued in the phase / (phase cycles x 4069 FLOP/cycle/CU)      (2.5 PFLOP/s dense bf16 / 256 CUs / 2.4 GHz)
One workgroup = one sample = one CU.  Writes profiles/<tag>_attn_phases.json.   usage: python tools/attn_phase_probe.py [tag]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from m3l_amd import _lib as L  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
raw = C.CDLL(sys.argv[2] if len(sys.argv) > 2 else L.LIB_PATH)
lib = L.lib()
dev = "cuda:0"
B, n, D, H = 256, 48, 192, 3
M = B * n
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
rn = lambda *s, dt=torch.float32, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)  # noqa: E731
x = rn(M, D)
wqkv, wo = rn(3 * D, D, dt=bf, sc=0.05), rn(D, D, dt=bf, sc=0.05)
bo, lw, lb = rn(D, sc=0.1), rn(D, sc=0.1) + 1, rn(D, sc=0.1)
xn1, o, xn2 = (torch.empty(M, D, device=dev, dtype=bf) for _ in range(3))
qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
lse, x1 = torch.empty(B * H * n, device=dev), torch.empty(M, D, device=dev)
ts = torch.zeros(B, 8, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
fwd = raw._Z18m3l_attn_block_fwdiiiPKfS0_S0_PKvS2_S0_S0_S0_fPvS3_S3_PfS4_S3_P12ihipStream_t


def run():
    rc = fwd(D, B, n, P(x), P(lw), P(lb), P(wqkv), P(wo), P(bo), P(lw), P(lb), C.c_float(1e-5), P(xn1), P(qkv), P(o), P(lse), P(x1), P(xn2), st)
    assert rc == 0


for _ in range(3):
    run()
torch.cuda.synchronize()
raw.m3l_set_attn_phase_buffer.argtypes = [C.c_void_p]
raw.m3l_set_attn_phase_buffer(P(ts))
acc = torch.zeros(5, dtype=torch.float64)
reps = 10
for _ in range(reps):
    run()
    torch.cuda.synchronize()
    d = (ts[:, 1:6] - ts[:, 0:5]).double().cpu()
    acc += d.median(dim=0).values
raw.m3l_set_attn_phase_buffer(None)
cyc = (acc / reps).tolist()
names = ["ln1", "qkv_proj", "attention", "out_proj", "residual_ln2"]
# MFMA FLOP per sample as ISSUED (16-row tiles, 32-key steps: 48 keys run as 64) and as useful
issued = {"qkv_proj": 2 * 48 * D * 3 * D, "attention": H * 3 * (2 * 2 * 16 * 64 * 64), "out_proj": 2 * 48 * D * D}
useful = {"qkv_proj": 2 * n * D * 3 * D, "attention": H * 2 * 2 * n * n * 64, "out_proj": 2 * n * D * D}
PEAK = 2.5e15 / 256 / 2.4e9
tot = sum(cyc)
out = {"kernel": "attn_block_fwd_kernel<3>", "shape": {"B": B, "n": n, "D": D, "heads": H}, "clock": "s_memtime shader clock, thread 0 of each workgroup, median over the 256 workgroups, mean of 10 launches",
       "peak_flop_per_cycle_per_cu": round(PEAK, 1), "total_cycles": round(tot, 1), "phases": {}}
for nm, c in zip(names, cyc):
    e = {"cycles": round(c, 1), "frac_of_kernel": round(c / tot, 4)}
    if nm in issued:
        e["mfma_util_issued"] = round(issued[nm] / (c * PEAK), 4)
        e["mfma_util_useful"] = round(useful[nm] / (c * PEAK), 4)
    out["phases"][nm] = e
out["kernel_mfma_util_useful"] = round(sum(useful.values()) / (tot * PEAK), 4)
try:
    out["_meta"] = json.load(open(os.path.join(ROOT, "m3l_amd", "lib", "build_stamp.json")))
except (OSError, ValueError):
    out["_meta"] = {}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for path in (os.path.join(ROOT, "gpurun_out", f"{tag}_attn_phases.json"),):
    json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
