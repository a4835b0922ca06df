#!/usr/bin/env python3
"""Micro-benchmark of the individual HIP kernels through the C ABI (device timers around back-to-back launches).
Usage on the GPU box:  python tools/bench_kernels.py [gemm_nt|gemm_tn|attn|ln|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from m3l_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
S = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def gemm_nt(code=1):
    T = torch.bfloat16 if code else torch.float32
    print(f"--- gemm_nt dtype={'bf16' if code else 'f32'}   (M,N,K)  epilogue   us   TFLOP/s   GB/s(min traffic)")
    for (M, N, K, epi) in [(12288, 576, 192, "t"), (49152, 576, 192, "t"), (12288, 192, 192, "res"), (49152, 192, 192, "res"),
                           (12288, 768, 192, "gelu"), (49152, 768, 192, "gelu"), (12288, 192, 768, "res"), (49152, 192, 768, "res"),
                           (12288, 768, 192, "dgelu"), (49152, 768, 192, "dgelu"), (12288, 192, 576, "t"), (49152, 192, 576, "t"),
                           (8192, 8192, 8192, "t")]:
        A = torch.randn(M, K, device=dev).to(T)
        W = torch.randn(N, K, device=dev).to(T)
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        o32 = torch.empty(M, N, device=dev)
        ot = torch.empty(M, N, device=dev, dtype=T)
        op = torch.empty(M, N, device=dev, dtype=T)
        u = torch.randn(M, N, device=dev).to(T)
        es = 2 if code else 4
        if epi == "t":
            f = lambda: L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, L.ptr(ot), None, None, 0, N, S())
            byt = M * K * es + N * K * es + M * N * es
        elif epi == "res":
            f = lambda: L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, L.ptr(bias), L.ptr(res), L.ptr(o32), None, None, None, 0, N, S())
            byt = M * K * es + N * K * es + M * N * 8
        elif epi == "gelu":
            f = lambda: L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, L.ptr(bias), None, None, L.ptr(ot), L.ptr(op), None, 1, N, S())
            byt = M * K * es + N * K * es + 2 * M * N * es
        else:
            f = lambda: L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, L.ptr(ot), None, L.ptr(u), 0, N, S())
            byt = M * K * es + N * K * es + 2 * M * N * es
        us = timeit(f, iters=20 if M * N * K > 1e11 else 50)
        print(f"({M:6d},{N:5d},{K:5d}) {epi:6s} {us:9.1f} us {2.0 * M * N * K / us / 1e6:8.1f} TF/s {byt / us / 1e3:8.1f} GB/s")


def gemm_tn(code=1):
    T = torch.bfloat16 if code else torch.float32
    print(f"--- gemm_tn dtype={'bf16' if code else 'f32'}")
    for (M, N, K) in [(12288, 576, 192), (49152, 576, 192), (12288, 192, 192), (49152, 192, 192), (12288, 768, 192), (49152, 768, 192),
                      (12288, 192, 768), (49152, 192, 768)]:
        Y = torch.randn(M, N, device=dev).to(T)
        X = torch.randn(M, K, device=dev).to(T)
        nb = L.lib().m3l_op_gemm_tn_ws_bytes(M, N, K)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        out = torch.empty(N, K, device=dev)
        f = lambda: L.lib().m3l_op_gemm_tn(code, L.ptr(Y), N, L.ptr(X), K, M, N, K, L.ptr(ws), nb, L.ptr(out), K, S())
        us = timeit(f)
        es = 2 if code else 4
        print(f"({M:6d},{N:5d},{K:5d}) {us:9.1f} us {2.0 * M * N * K / us / 1e6:8.1f} TF/s {(M * (N + K) * es) / us / 1e3:8.1f} GB/s")


def attn(code=1):
    T = torch.bfloat16 if code else torch.float32
    print(f"--- attention dtype={'bf16' if code else 'f32'}")
    for (B, n, H) in [(256, 48, 3), (256, 192, 3)]:
        qkv = torch.randn(B * n, 3 * H * 64, device=dev).to(T)
        o = torch.empty(B * n, H * 64, device=dev, dtype=T)
        dO = torch.randn(B * n, H * 64, device=dev).to(T)
        dqkv = torch.empty_like(qkv)
        lse = torch.empty(B, H, n, device=dev)
        ds = torch.empty(B, H, n, device=dev)
        f = lambda: L.lib().m3l_op_attn_fwd(code, L.ptr(qkv), L.ptr(o), L.ptr(lse), B, n, H, S())
        us = timeit(f)
        fl = 4.0 * B * H * n * n * 64
        es = 2 if code else 4
        print(f"fwd B={B} n={n} H={H}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s {(B * n * H * 64 * 4 * es) / us / 1e3:8.1f} GB/s")
        f = lambda: L.lib().m3l_op_attn_bwd(code, L.ptr(qkv), L.ptr(o), L.ptr(dO), L.ptr(lse), L.ptr(ds), L.ptr(dqkv), B, n, H, S())
        us = timeit(f)
        print(f"bwd B={B} n={n} H={H}: {us:8.1f} us {2.5 * fl / us / 1e6:7.1f} TF/s")


def ln():
    print("--- layernorm")
    for M in (12288, 49152):
        D = 192
        x = torch.randn(M, D, device=dev)
        g = torch.randn(D, device=dev)
        b = torch.randn(D, device=dev)
        y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        f = lambda: L.lib().m3l_layernorm_fwd(1, L.ptr(x), M, D, L.ptr(g), L.ptr(b), 1e-5, L.ptr(y), None, S())
        us = timeit(f)
        print(f"ln_fwd M={M}: {us:7.1f} us {(M * D * 6) / us / 1e3:8.1f} GB/s")
        dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
        dx = torch.empty(M, D, device=dev)
        dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
        ws = torch.empty(L.lib().m3l_layernorm_ws_bytes(D), dtype=torch.uint8, device=dev)
        f = lambda: L.lib().m3l_layernorm_bwd(1, L.ptr(dy), L.ptr(x), M, D, L.ptr(g), 1e-5, L.ptr(dx), L.ptr(dx), L.ptr(ws), L.ptr(dg), L.ptr(db), S())
        us = timeit(f)
        print(f"ln_bwd M={M}: {us:7.1f} us {(M * D * 14) / us / 1e3:8.1f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    L.lib()
    if which in ("gemm_nt", "all"):
        gemm_nt(1)
    if which in ("gemm_tn", "all"):
        gemm_tn(1)
    if which in ("attn", "all"):
        attn(1)
    if which in ("ln", "all"):
        ln()
    if which == "f32":
        gemm_nt(0)
        gemm_tn(0)
        attn(0)
