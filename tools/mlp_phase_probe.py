#!/usr/bin/env python3
"""Where mlp_block_fwd's chunk loop spends its cycles: a diagnostic build (-DMB_STAMP=1, bash tools/t192_ablate.sh "1" mlp_block MB_STAMP)
accumulates shader-clock deltas per wave and loop section; this prints their means over the 256 workgroups x 12 compute waves, per chunk.
usage: python tools/mlp_phase_probe.py build_abl/libmlp_block_1.so"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

lib = C.CDLL(sys.argv[1])
dev = "cuda:0"
B, n, D, mlp = 256, 48, 192, 768
M = B * n
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
rn = lambda *s, dt=torch.float32, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)  # noqa: E731
x1, xn = rn(M, D), rn(M, D, dt=bf)
w1, w2 = rn(mlp, D, dt=bf, sc=0.05), rn(D, mlp, dt=bf, sc=0.05)
b1, b2 = rn(mlp, sc=0.1), rn(D, sc=0.1)
u, h, xout = torch.empty(M, mlp, device=dev, dtype=bf), torch.empty(M, mlp, device=dev, dtype=bf), torch.empty(M, D, device=dev)
stamps = torch.zeros(B * 12 * 16, dtype=torch.int64, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
assert lib.m3l_mb_set_stamps(P(stamps)) == 0
fn = lib._Z17m3l_mlp_block_fwdiiiiPKvPKfS0_S2_S0_S2_PvS3_PfP12ihipStream_t
for _ in range(5):
    assert fn(D, mlp, B, n, P(xn), P(x1), P(w1), P(b1), P(w2), P(b2), P(u), P(h), P(xout), st) == 0
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    fn(D, mlp, B, n, P(xn), P(x1), P(w1), P(b1), P(w2), P(b2), P(u), P(h), P(xout), st)
b.record()
torch.cuda.synchronize()
t = stamps.view(B, 12, 16).double().cpu()
names = ["loop head + bias read", "barrier 1 (W1 block / slowest wave)", "fc1 fragments (6 ds_read_b128) landed", "fc1 6 MFMAs",
         "bias + GELU + 8 ds_write_b16 + drain", "barrier 2", "u / h chunk -> global (issue)", "fc2 fragments (8 ds_read_b128) landed", "fc2 6 MFMAs"]
NC = mlp // 64
tot = t[:, :, 10].mean().item()
print(f"kernel (instrumented) {a.elapsed_time(b) / 20 * 1e3:.1f} us/launch; chunk loop {tot:.0f} cycles per wave = {tot / NC:.0f} per chunk")
for i, nm in enumerate(names):
    m = t[:, :, i].mean().item()
    print(f"  {nm:42s} {m / NC:8.1f} cycles/chunk  {m / tot * 100:5.1f} %   (min wave {t[:, :, i].mean(0).min().item() / NC:7.1f}, max wave {t[:, :, i].mean(0).max().item() / NC:7.1f})")
