"""Standalone timing of the six row-tiled decoder kernels (t192.hip) at the cfg-2 decoder shape (B = 256, n = 192, D = 192, mlp = 768),
back to back on one stream — diagnostic only (calls the library's internal C++ launchers by their mangled names).
usage: python tools/t192_probe.py [lib.so] [kernel-name-substring]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m3l_amd import _lib as L  # noqa: E402

LIBP = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else L.LIB_PATH
ONLY = [a for a in sys.argv[1:] if not a.endswith(".so")]
raw = C.CDLL(LIBP)
dev = "cuda:0"
B, n, D, mlp = int(os.environ.get("PB", "256")), 192, 192, 768
M = B * n
g = torch.Generator(device=dev).manual_seed(0)


def rn(*s, dt=torch.float32, sc=1.0):
    return (torch.randn(*s, device=dev, generator=g) * sc).to(dt)


bf = torch.bfloat16
x, x1, dres = rn(M, D), rn(M, D), rn(M, D)
xn, o, dxt = rn(M, D, dt=bf), rn(M, D, dt=bf), rn(M, D, dt=bf)
qkv, dqkv = rn(M, 3 * D, dt=bf), rn(M, 3 * D, dt=bf)
u, h, du = rn(M, mlp, dt=bf), torch.empty(M, mlp, device=dev, dtype=bf), torch.empty(M, mlp, device=dev, dtype=bf)
lse = rn(B * 3 * n).abs() + 3.0
wqkv, wqkvT = rn(3 * D, D, dt=bf, sc=0.05), rn(D, 3 * D, dt=bf, sc=0.05)
wo, woT = rn(D, D, dt=bf, sc=0.05), rn(D, D, dt=bf, sc=0.05)
w1, w1T = rn(mlp, D, dt=bf, sc=0.05), rn(D, mlp, dt=bf, sc=0.05)
w2, w2T = rn(D, mlp, dt=bf, sc=0.05), rn(mlp, D, dt=bf, sc=0.05)
b1, b2, bo, lw, lb = rn(mlp, sc=0.1), rn(D, sc=0.1), rn(D, sc=0.1), rn(D, sc=0.1) + 1, rn(D, sc=0.1)
xout, x1o, dxo = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
xn_o, dxt_o, o_o = torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf)
qkv_o = torch.empty(M, 3 * D, device=dev, dtype=bf)
tiles = (M + 191) // 192
cs_part, ln_part = torch.empty(M // 16 + 16, mlp, device=dev), torch.empty(M // 48 + 1, 3 * D, device=dev)    # sized for every tile shape
st = torch.cuda.current_stream().cuda_stream
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
F = C.c_float(1e-5)
raw.m3l_set_t192(7)

KERNELS = {
    "attn_t192_fwd": lambda: raw._Z17m3l_attn_t192_fwdiiiPKfS0_S0_PKvfPvS3_S3_PfP12ihipStream_t(
        D, B, n, P(x), P(lw), P(lb), P(wqkv), F, P(xn_o), P(qkv_o), P(o_o), P(lse), C.c_void_p(st)),
    "attn_tail_mlp_t192_fwd": lambda: raw._Z26m3l_attn_tail_mlp_t192_fwdiiiPKvPKfS0_S2_S2_S2_fPfPvS0_S2_S0_S2_S4_S4_S3_P12ihipStream_t(
        D, M, mlp, P(o), P(x), P(wo), P(bo), P(lw), P(lb), F, P(x1o), P(xn_o), P(w1), P(b1), P(w2), P(b2), P(u), P(h), P(xout), C.c_void_p(st)),
    "mlp_t192_fwd": lambda: raw._Z16m3l_mlp_t192_fwdiiiPKvPKfS0_S2_S0_S2_PvS3_PfP12ihipStream_t(
        D, M, mlp, P(xn), P(x1), P(w1), P(b1), P(w2), P(b2), P(u), P(h), P(xout), C.c_void_p(st)),
    "mlp_t192_bwd": lambda: raw._Z16m3l_mlp_t192_bwdiiiPKvPfPKfS3_S0_S0_S0_fPvS4_S1_S1_P12ihipStream_t(
        D, M, mlp, P(dxt), P(dxo), P(x1), P(lw), P(u), P(w2T), P(w1T), F, P(du), P(dxt_o), P(cs_part), P(ln_part), C.c_void_p(st)),
    "attn_t192_bwd": lambda: raw._Z17m3l_attn_t192_bwdiiiPKvS0_S0_PKfS0_PvP12ihipStream_t(
        D, B, n, P(dxt), P(qkv), P(o), P(lse), P(woT), P(qkv_o), C.c_void_p(st)),
    "qkv_bwd_t192": lambda: raw._Z16m3l_qkv_bwd_t192iiiPKvPKfS2_S0_S2_fPfPvS3_P12ihipStream_t(
        D, M, 3 * D, P(dqkv), P(x), P(lw), P(wqkvT), P(dres), F, P(dxo), P(dxt_o), P(ln_part), C.c_void_p(st)),
}
tot = 0.0
out = []
for name, fn in KERNELS.items():
    if ONLY and not any(s in name for s in ONLY):
        continue
    for _ in range(3):
        rc = fn()
        assert rc == 0, (name, rc)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    tot += us
    out.append(f"{name} {us:.1f}")
print(f"{os.path.basename(LIBP)}: " + " | ".join(out) + f" | sum {tot:.1f} us")
