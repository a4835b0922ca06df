"""Does de-synchronising the row-tiled decoder kernels pay?  The decoder forward (4 layers, D = 192, n = 192) of B = 256 samples on one
stream against the same work as two half batches on two streams (each kernel then covers half the CUs and the two chains drift apart,
so that the memory-bound heads / tails of one overlap the compute loop of the other).  Forward only, no_grad."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from m3l_amd.pretrain_models import Transformer
dev = torch.device("cuda:0")
torch.manual_seed(0)
tf = Transformer(192, 4, 3, 64, 768); tf.compute_dtype = "bf16"; tf = tf.to(dev)
B, n = 256, 192
x = torch.randn(B, n, 192, device=dev)
xa, xb = x[:128].contiguous(), x[128:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    with torch.no_grad(): return tf(x)
def two():
    with torch.no_grad():
        with torch.cuda.stream(s1): a = tf(xa)
        with torch.cuda.stream(s2): b = tf(xb)
    return a, b
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6, (t1 - t0) / it * 1e6
for r in range(3):
    a, b = timeit(one), timeit(two)
    print(f"one stream B=256: {a[0]:.1f} us (host enqueue {a[1]:.1f})   two streams 2 x B=128: {b[0]:.1f} us (host enqueue {b[1]:.1f})")
ya = one(); a, b = two(); torch.cuda.synchronize()
print("max diff", float((torch.cat([a, b]) - ya).abs().max()))
