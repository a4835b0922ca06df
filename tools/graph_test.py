"""Whole-step HIP graph (torch.cuda.CUDAGraph over zero_grad + fwd + bwd + Adam): viability and speed vs eager."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from m3l_amd.parallel import FlatAdam, GradSync
dev = torch.device("cuda:0")
c = bench.CFG2
mae = bench.build_model(c, "bf16", dev)
sync = GradSync(mae); opt = FlatAdam(sync, lr=1e-4, capturable=True)
B = 256
x = {"image": torch.rand(B, 3, 64, 64, device=dev), "tactile1": torch.rand(B, 3, 32, 32, device=dev), "tactile2": torch.rand(B, 3, 32, 32, device=dev)}
def step():
    sync.zero_grad(); loss = mae(x); loss.backward(); sync.finish(); opt.step(); return loss
def timeit(f, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(20): step()
torch.cuda.current_stream().wait_stream(s)
print("eager ms/step", timeit(step), timeit(step))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
losses = []
for _ in range(5):
    g.replay(); losses.append(float(loss))
print("replay losses", losses, "step_dev", int(opt._step_dev.item()))
print("graph ms/step", timeit(g.replay), timeit(g.replay))
