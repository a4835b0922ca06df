"""MAEExtractor.forward under no_grad at rollout batch sizes: eager (vt_load + two library calls) against a hipGraph replay of the same call
(static input buffers, one copy in, one replay).  Prints ms per call and checks that the replay reproduces the eager features."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from m3l_amd import MAEExtractor

dev = torch.device("cuda:0")
c = bench.CFGREF
mae = bench.build_model(c, "bf16", dev)
fs = c.get("frame_stack", 1)
ext = MAEExtractor(mae, c["dim"], False, fs).to(dev)


def timeit(f, n=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for B in (1, 8, 32, 128):
    obs = {"image": torch.rand(B, fs, c["image_size"], c["image_size"], 3, device=dev),
           "tactile": torch.rand(B, fs, 3 * c["num_tactiles"], c["tactile_size"], c["tactile_size"], device=dev) * 2 - 1}

    def eager():
        with torch.no_grad():
            return ext(obs)
    ref = eager().clone()
    t_e = timeit(eager)
    static = {k: v.clone() for k, v in obs.items()}
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            ext(static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out = ext(static)

    def replay():
        for k in static:
            static[k].copy_(obs[k])
        g.replay()
        return out
    got = replay().clone()
    t_g = timeit(replay)
    print(f"B={B:4d}  eager {t_e:.3f} ms/call  graph {t_g:.3f} ms/call  max|diff| {float((got - ref).abs().max()):.3e}", flush=True)
