#!/usr/bin/env python3
"""Two data-parallel ranks through the REAL HIP step (chunked transformer backward, GradSync buckets, FlatAdam) on a one-GPU box:
both ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device).  Checks, after 3 optimizer steps:
  * both ranks hold bit-identical parameters;
  * they equal a single-process run on the concatenated batch (mean-of-rank-means == global mean for equal per-rank batches).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/dp_two_rank_check.py
(launched from a shell, not from a process that already initialised the GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from m3l_amd import VTMAE, VTT  # noqa: E402
from m3l_amd.parallel import FlatAdam, GradSync  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 2


def build():
    torch.manual_seed(0)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=6, heads=2, mlp_dim=128, num_tactiles=2)
    return VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=2, decoder_heads=2, num_tactiles=2, compute_dtype="fp32").to(dev)


def data(step, lo, hi):
    g = torch.Generator().manual_seed(100 + step)
    B = 8
    x = {"image": torch.rand(B, 3, 32, 32, generator=g), "tactile1": torch.rand(B, 3, 16, 16, generator=g), "tactile2": torch.rand(B, 3, 16, 16, generator=g)}
    noises = [torch.rand(B, 16, generator=g) for _ in range(3)]
    return {k: v[lo:hi].to(dev) for k, v in x.items()}, [n[lo:hi].to(dev) for n in noises]


def train(mae, sync, lo, hi):
    opt = FlatAdam(sync, lr=1e-3)
    for step in range(3):
        x, noises = data(step, lo, hi)
        sync.zero_grad()
        mae(x, mask_noise=noises).backward()
        sync.finish()
        opt.step()
    torch.cuda.synchronize()
    return sync.flat_params.clone()


mae = build()
sync = GradSync(mae)
assert sync._comm and sync.world == 2
p_dp = train(mae, sync, 4 * rank, 4 * rank + 4)
gathered = [torch.empty_like(p_dp) for _ in range(2)]
dist.all_gather(gathered, p_dp)
same = torch.equal(gathered[0], gathered[1])
if rank == 0:
    ref = build()
    rs = GradSync(ref)
    rs._comm, rs.world = False, 1                      # single-process reference on the full batch, no collectives
    p_ref = train(ref, rs, 0, 8)
    err = float((p_dp - p_ref).abs().max() / p_ref.abs().max())
    print(f"ranks identical: {same}; vs single-process full batch: max rel err {err:.2e}", flush=True)
    assert same and err < 1e-5, (same, err)
dist.barrier()
dist.destroy_process_group()
