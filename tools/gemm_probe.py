"""Where does the NT GEMM's time go?  Same launch with outputs removed (through the C ABI: null out pointers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from m3l_amd import _lib as L
dev = torch.device("cuda:0"); S = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (M, N, K) in [(49152, 576, 192), (49152, 768, 192), (12288, 576, 192)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
    ot = torch.empty(M, N, device=dev, dtype=torch.bfloat16); o32 = torch.empty(M, N, device=dev)
    none = timeit(lambda: L.lib().m3l_op_gemm_nt(1, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, None, None, None, 0, N, S()))
    t = timeit(lambda: L.lib().m3l_op_gemm_nt(1, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, L.ptr(ot), None, None, 0, N, S()))
    f = timeit(lambda: L.lib().m3l_op_gemm_nt(1, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, L.ptr(o32), None, None, None, 0, N, S()))
    print(f"({M},{N},{K}): no-output {none:.1f} us, bf16 out {t:.1f} us, f32 out {f:.1f} us")
