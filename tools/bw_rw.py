"""HBM read-only / write-only / copy bandwidth (torch elementwise kernels), for pricing write-heavy GEMM epilogues."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (75, 150, 600, 2400):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty(n, device=dev)
    s = t(lambda: y.fill_(1.0)); print(f"{mb} MB write-only (fill): {s*1e6:.1f} us {mb/1e3/s:.0f} GB/s")
    s = t(lambda: y.zero_()); print(f"{mb} MB write-only (memset): {s*1e6:.1f} us {mb/1e3/s:.0f} GB/s")
    s = t(lambda: x.sum()); print(f"{mb} MB read-only (sum): {s*1e6:.1f} us {mb/1e3/s:.0f} GB/s")
    s = t(lambda: y.copy_(x)); print(f"{mb} MB copy: {s*1e6:.1f} us {2*mb/1e3/s:.0f} GB/s r+w")
    xb = x.bfloat16()
    s = t(lambda: torch.add(x, 1.0, out=y)); print(f"{mb} MB x+1: {s*1e6:.1f} us {2*mb/1e3/s:.0f} GB/s r+w")
