#!/usr/bin/env python3
"""mlp_t192_bwd alone at the cfg-2 decoder shape under the phase-offset / priority knobs (m3l_set_t192_stagger): microseconds per launch
(interleaved rounds in one process, median and min) and bit-equality of every output with the baseline kernel.
usage: python tools/t192_stagger_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m3l_amd import _lib as L  # noqa: E402

raw = C.CDLL(L.LIB_PATH)
lib = L.lib()
dev = "cuda:0"
B, n, D, mlp = 256, 192, 192, 768
M = B * n
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16


def rn(*s, dt=torch.float32, sc=1.0):
    return (torch.randn(*s, device=dev, generator=g) * sc).to(dt)


x1, dx0 = rn(M, D), rn(M, D)
dxt, u = rn(M, D, dt=bf), rn(M, mlp, dt=bf)
w1T, w2T = rn(D, mlp, dt=bf, sc=0.05), rn(mlp, D, dt=bf, sc=0.05)
lw = rn(D, sc=0.1) + 1
st = torch.cuda.current_stream().cuda_stream
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
F = C.c_float(1e-5)
raw.m3l_set_t192(7)
NSET = 3                                   # output sets rotated so that back-to-back launches do not hit the same lines
outs = [dict(dx=torch.empty(M, D, device=dev), du=torch.empty(M, mlp, device=dev, dtype=bf), dxt=torch.empty(M, D, device=dev, dtype=bf),
             cs=torch.empty(M // 16 + 16, mlp, device=dev), ln=torch.empty(M // 48 + 1, 3 * D, device=dev)) for _ in range(NSET)]


def launch(o):
    o["dx"].copy_(dx0)                      # dx is read (the residual gradient) and overwritten in place
    rc = raw._Z16m3l_mlp_t192_bwdiiiPKvPfPKfS3_S0_S0_S0_fPvS4_S1_S1_P12ihipStream_t(
        D, M, mlp, P(dxt), P(o["dx"]), P(x1), P(lw), P(u), P(w2T), P(w1T), F, P(o["du"]), P(o["dxt"]), P(o["cs"]), P(o["ln"]), C.c_void_p(st))
    assert rc == 0


def timed(reps=10):
    for o in outs:
        o["dx"].copy_(dx0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        o = outs[i % NSET]
        raw._Z16m3l_mlp_t192_bwdiiiPKvPfPKfS3_S0_S0_S0_fPvS4_S1_S1_P12ihipStream_t(
            D, M, mlp, P(dxt), P(o["dx"]), P(x1), P(lw), P(u), P(w2T), P(w1T), F, P(o["du"]), P(o["dxt"]), P(o["cs"]), P(o["ln"]), C.c_void_p(st))
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


CONFIGS = [("baseline kernel", 0, 0), ("stg kernel, nobody leads", 0x1000, 0), ("stg, everybody leads", 0x1fff, 0),
           ("stg, waves 4-7 lead", 0x10f0, 0), ("stg, waves 0-3 lead", 0x100f, 0), ("stg, waves 8-11 lead", 0x1f00, 0),
           ("stg, waves 0-5 lead", 0x103f, 0), ("stg, waves 6-11 lead", 0x1fc0, 0), ("stg, odd waves lead", 0x1aaa, 0),
           ("stg, waves 0-7 lead", 0x10ff, 0), ("stg, waves 4-11 lead", 0x1ff0, 0),
           ("stg 4-7 lead + prio 4-7", 0x10f0, 0x0f0), ("stg 4-7 lead + prio 0-3,8-11", 0x10f0, 0xf0f), ("stg 4-11 lead + prio 4-11", 0x1ff0, 0xff0),
           ("baseline order + prio 4-11", 0x1000, 0xff0), ("baseline order + prio 8-11", 0x1000, 0xf00)]
lib.m3l_set_t192_stagger(0, 0)
launch(outs[0])
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in outs[0].items()}
for name, stg, prio in CONFIGS[1:]:
    lib.m3l_set_t192_stagger(stg, prio)
    launch(outs[1])
    torch.cuda.synchronize()
    tiles = (M + 191) // 192
    same = all(torch.equal(outs[1][k][: (tiles if k in ("cs", "ln") else M)], ref[k][: (tiles if k in ("cs", "ln") else M)]) for k in ref)
    assert same, f"{name}: outputs differ from the baseline kernel"
print("every configuration is bit-identical to the baseline kernel", flush=True)
times = {c[0]: [] for c in CONFIGS}
for rnd in range(7):
    for name, stg, prio in CONFIGS:
        lib.m3l_set_t192_stagger(stg, prio)
        timed(3)
        times[name].append(timed(12))
for name, _, _ in CONFIGS:
    t = sorted(times[name])
    print(f"{name:34s} median {t[len(t) // 2]:7.1f} us   min {t[0]:7.1f} us", flush=True)
