import torch, time
dev = torch.device("cuda:0")
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): y.copy_(x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"copy {mb} MB: {ms*1e3:.1f} us  {2*mb/1024/ (ms/1e3):.0f} GB/s (read+write)")
# bf16 matmul peak via torch (hipBLASLt) for reference
for (M, N, K) in [(8192, 8192, 8192), (49152, 768, 192), (49152, 192, 768), (12288, 576, 192)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); B = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): C = A @ B.t()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): C = A @ B.t()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"torch bf16 matmul ({M},{N},{K}): {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF/s")
