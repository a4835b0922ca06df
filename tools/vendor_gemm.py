"""Our NT GEMM (PLAIN epilogue: bf16 out, no bias) against torch's bf16 matmul (hipBLASLt) at the cfg-2 Linear shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from m3l_amd import _lib as L
dev = torch.device("cuda:0"); S = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
print("shape (M,N,K)            ours us   hipBLASLt us   ratio")
CFG4 = [(7232, 1152, 384), (7232, 384, 384), (7232, 1536, 384), (7232, 384, 1536), (7232, 384, 1152),
        (5120, 768, 256), (5120, 256, 256), (5120, 1024, 256), (5120, 256, 1024)]      # cfg 4 / cfg 5 encoder (B = 64), M3L default encoder (B = 512)
for (M, N, K) in (CFG4 if "--wide" in sys.argv else []) + [(12288, 576, 192), (12288, 192, 192), (12288, 768, 192), (12288, 192, 768), (12288, 192, 576),
                  (49152, 576, 192), (49152, 192, 192), (49152, 768, 192), (49152, 192, 768), (49152, 192, 576)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
    ot = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ours = timeit(lambda: L.lib().m3l_op_gemm_nt(1, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, L.ptr(ot), None, None, 0, N, S()))
    Wt = W.t()
    ven = timeit(lambda: torch.matmul(A, Wt, out=ot))
    print(f"({M:6d},{N:4d},{K:4d})   {ours:8.1f}   {ven:8.1f}   {ours / ven:6.2f}")
