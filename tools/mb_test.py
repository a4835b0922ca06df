"""Prototype: does running two half-batches as independent chains on two HIP streams beat one full batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
c = bench.CFG2
mae = bench.build_model(c, "bf16", dev)
B = 256
x = {"image": torch.rand(B, 3, 64, 64, device=dev), "tactile1": torch.rand(B, 3, 32, 32, device=dev), "tactile2": torch.rand(B, 3, 32, 32, device=dev)}
def full():
    mae.zero_grad(set_to_none=True)
    mae(x).backward()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
halves = [{k: v[i * B // 2:(i + 1) * B // 2].contiguous() for k, v in x.items()} for i in range(2)]
def split():
    mae.zero_grad(set_to_none=True)
    cur = torch.cuda.current_stream()
    losses = []
    for s, h in zip(streams, halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            losses.append(mae(h))
    torch.autograd.backward(losses)
    for s in streams:
        cur.wait_stream(s)
def timeit(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full batch  : %.3f ms" % timeit(full))
print("2 x half, 2 streams: %.3f ms" % timeit(split))
