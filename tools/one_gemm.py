import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from m3l_amd import _lib as L
M, N, K = [int(v) for v in sys.argv[1:4]]
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = torch.randn(N, K, device=dev).to(torch.bfloat16)
ot = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
S = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    L.lib().m3l_op_gemm_nt(1, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, L.ptr(ot), None, None, 0, N, S)
torch.cuda.synchronize()
