#!/usr/bin/env python3
"""The grouped weight-gradient launch alone (wgrad.hip through m3l_op_gemm_tn_grouped) at the cfg-2 shapes: the decoder's two-layer
group (M = 49152, 8 problems) and the encoder's four-layer group (M = 12288, 16 problems).  Prints per launch: microseconds (HIP
events over `reps` back-to-back launches, operands rotated over `nset` buffer sets so that the Infinity Cache does not serve them),
algorithmic GB/s (both operands once + dW once), TFLOP/s, and the max error against torch (fp32 matmul of the bf16 operands).

usage: python tools/wgrad_probe.py [reps]      (environment switches of wgrad.hip apply: M3L_WGRAD_WAVES, M3L_WGRAD_NT, ...)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from m3l_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def layer_shapes(D, HD, mlp):
    # (N, K) of dW = Y^T X with Y [M, N], X [M, K]: fc2, fc1, qkv, out-proj (the order mae_plan.hip pushes them)
    return [(D, mlp), (mlp, D), (3 * HD, D), (D, HD)]


def run(name, M, layers, D=192, HD=192, mlp=768, nset=3, check=True):
    shapes = layer_shapes(D, HD, mlp) * layers
    cnt = len(shapes)
    g = torch.Generator(device="cpu").manual_seed(0)
    sets = []
    for _ in range(nset):
        Ys = [(torch.randn(M, n, generator=g) * 0.5).to(torch.bfloat16).to(dev) for n, _ in shapes]
        Xs = [(torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev) for _, k in shapes]
        sets.append((Ys, Xs))
    outs = [torch.empty(n, k, dtype=torch.float32, device=dev) for n, k in shapes]
    Ns = (C.c_int * cnt)(*[n for n, _ in shapes])
    Ks = (C.c_int * cnt)(*[k for _, k in shapes])
    wsb = lib.m3l_op_gemm_tn_grouped_ws_bytes(1, cnt, M, Ns, Ks)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch(s):
        Ys, Xs = sets[s % nset]
        L.check(lib.m3l_op_gemm_tn_grouped(1, cnt, M, L.ptr_array(Ys), Ns, L.ptr_array(Xs), Ks, Ns, Ks, L.ptr_array(outs), L.ptr(ws), wsb, st),
                "m3l_op_gemm_tn_grouped")
    launch(0)
    torch.cuda.synchronize()
    err = 0.0
    if check:
        Ys, Xs = sets[0]
        for i in range(cnt):
            ref = Ys[i].float().t() @ Xs[i].float()
            err = max(err, float((outs[i] - ref).abs().max()) / float(ref.abs().max()))
    for s in range(3):
        launch(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in range(reps):
        launch(s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    byt = sum(2.0 * M * (n + k) + 4.0 * n * k for n, k in shapes)
    fl = sum(2.0 * M * n * k for n, k in shapes)
    print(f"{name:28s} M={M:6d} problems={cnt:2d}  {us:8.1f} us/launch (kernel + slab reduce)  {byt / us / 1e3:7.1f} GB/s algorithmic  "
          f"{fl / us / 1e6:6.1f} TFLOP/s  ws {wsb / 1e6:.0f} MB  max rel err {err:.2e}", flush=True)
    return us


if __name__ == "__main__":
    run("decoder 2-layer group", 49152, 2)
    run("encoder 4-layer group", 12288, 4)
    run("decoder 4-layer group", 49152, 4, check=False)
