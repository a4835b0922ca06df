"""EXPERIMENT (EXPERIMENTS.md 5.3): the encoder's feed-forward forward (B = 256 samples x 48 tokens, D = 192, mlp = 768) with each 48-token
tile split over `nslice` independent 4-wave workgroups (mlp_split_fwd_kernel, t192.hip) against the per-sample block kernel
(mlp_block_fwd).  Checks u, h and the sum of the partial outputs against torch, then times both (stand-alone, back to back).
usage: python tools/mlp_split_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m3l_amd import _lib as L  # noqa: E402

raw = C.CDLL(L.LIB_PATH)
dev = "cuda:0"
B, n, D, mlp = int(os.environ.get("PB", "256")), 48, 192, 768
M = B * n
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16


def rn(*s, dt=torch.float32, sc=1.0):
    return (torch.randn(*s, device=dev, generator=g) * sc).to(dt)


xn2, x1 = rn(M, D, dt=bf), rn(M, D)
w1, w2 = rn(mlp, D, dt=bf, sc=0.05), rn(D, mlp, dt=bf, sc=0.05)
b1, b2 = rn(mlp, sc=0.1), rn(D, sc=0.1)
u, h = torch.empty(M, mlp, device=dev, dtype=bf), torch.empty(M, mlp, device=dev, dtype=bf)
u2, h2 = torch.empty_like(u), torch.empty_like(h)
xout = torch.empty(M, D, device=dev)
ypart = torch.zeros(6, M, D, device=dev)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def block():
    rc = raw._Z17m3l_mlp_block_fwdiiiiPKvPKfS0_S2_S0_S2_PvS3_PfP12ihipStream_t(D, mlp, B, n, P(xn2), P(x1), P(w1), P(b1), P(w2), P(b2), P(u2), P(h2),
                                                                              P(xout), C.c_void_p(st))
    assert rc == 0


def split(nslice, nst, dw=1, abl=0):
    rc = raw.m3l_mlp_split_fwd_probe(M, mlp, nslice, nst, dw, abl, P(xn2), P(w1), P(b1), P(w2), P(u), P(h), P(ypart), C.c_void_p(st))
    assert rc == 0, L.last_error() if hasattr(L, "last_error") else rc


def timeit(f, iters=200):
    for _ in range(20):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


# reference (fp32 math on the bf16 operands, as the kernels accumulate)
ur = (xn2.float() @ w1.float().t() + b1).to(bf)
block()
torch.cuda.synchronize()
for nslice, nst in ((3, 2), (3, 3), (2, 3), (1, 3), (4, 2), (6, 2)):
    u.zero_(); h.zero_(); ypart.zero_()
    split(nslice, nst)
    torch.cuda.synchronize()
    y = ypart[:nslice].sum(0) + b2 + x1
    eu = float((u.float() - ur.float()).abs().max())
    eh = float((h.float() - h2.float()).abs().max())
    ey = float((y - xout).abs().max())
    print(f"nslice {nslice} nst {nst}: max|u - ref| {eu:.3e}  max|h - block| {eh:.3e}  max|y - block xout| {ey:.3e}", flush=True)
for rep in range(2):
    print(f"mlp_block_fwd                  {timeit(block):7.2f} us")
    for nslice, nst, dw in ((3, 2, 1), (3, 2, 2), (2, 3, 1), (2, 3, 2), (1, 3, 1), (1, 3, 2)):
        print(f"split nslice {nslice} nst {nst} dw {dw}:  " + "  ".join(f"abl{a}={timeit(lambda: split(nslice, nst, dw, a)):6.2f}" for a in (0, 1, 2, 4, 8, 3, 7, 15, 16, 48, 32)), flush=True)
