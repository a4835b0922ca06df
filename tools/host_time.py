import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from m3l_amd.parallel import FlatAdam, GradSync
dev = torch.device("cuda:0")
c = bench.CFG2
mae = bench.build_model(c, "bf16", dev)
sync = GradSync(mae); opt = FlatAdam(sync, lr=1e-4)
B = 256
x = {"image": torch.rand(B, 3, 64, 64, device=dev), "tactile1": torch.rand(B, 3, 32, 32, device=dev), "tactile2": torch.rand(B, 3, 32, 32, device=dev)}
def step():
    sync.zero_grad(); loss = mae(x); loss.backward(); sync.finish(); opt.step()
from m3l_amd import functional as Fn
# host enqueue time: run steps but sync only at the end; measure python time per step when GPU queue is deep.
# A/B: the step as two library calls (csrc/mae_step.hip, default) vs the five per-module autograd Functions
for fused in (True, False, True, False):
    Fn.FUSED_STEP = fused
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"fused={fused}: host enqueue {1e3*(t1-t0)/20:.2f} ms/step ; total {1e3*(t2-t0)/20:.2f} ms/step ; {B*20/(t2-t0):.0f} samples/s", flush=True)
Fn.FUSED_STEP = True
# forward-only and backward-only host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
