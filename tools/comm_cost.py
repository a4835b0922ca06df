"""Host / device cost of one async RCCL all-reduce call as GradSync issues it (world size 1 rehearsal on a one-GPU box)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat = torch.zeros(7_280_000, device=dev)
for n in (50_000, 1_800_000, 7_280_000):
    for _ in range(5):
        dist.all_reduce(flat[:n], async_op=True).wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ws = []
    for _ in range(100):
        ws.append(dist.all_reduce(flat[:n], async_op=True))
    t1 = time.perf_counter()
    for w in ws: w.wait()
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"n={n}: issue {1e4*(t1-t0):.1f} us/call host, wait {1e4*(t2-t1):.1f} us/call host, drain {1e3*(t3-t2):.2f} ms total")
dist.destroy_process_group()
