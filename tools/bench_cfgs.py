#!/usr/bin/env python3
"""Secondary workloads (not the headline metric): MAE fwd+bwd+Adam at BASELINE cfg 4 / cfg 5 / the reference-default architecture,
and the cfg-5 policy-side extractor (get_embeddings + frozen DINOv2 + fusion MLP).  One line per workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from m3l_amd import VTMAE, VTT, DinoV2Frozen, DinoCatMAEExtractor
from m3l_amd.parallel import FlatAdam, GradSync
dev = torch.device("cuda:0")

ONLY = os.environ.get("M3L_CFGS_ONLY")          # substring of a workload name: run only that one (profiling)

def run(name, enc_kw, mae_kw, B, C, hw_i, hw_t, k, steps=20):
    if ONLY and ONLY not in name:
        return None
    torch.manual_seed(0)
    mae = VTMAE(encoder=VTT(**enc_kw), compute_dtype="bf16", **mae_kw).to(dev)
    sync = GradSync(mae); opt = FlatAdam(sync, lr=1e-4)
    x = {"image": torch.rand(B, C, hw_i, hw_i, device=dev)}
    for i in range(k): x[f"tactile{i+1}"] = torch.rand(B, C, hw_t, hw_t, device=dev)
    def step():
        sync.zero_grad(); mae(x).backward(); sync.finish(); opt.step()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{name}: B={B} {el/steps*1e3:.2f} ms/step  {B*steps/el:.0f} samples/s", flush=True)
    return mae

run("cfg4 ViT-Small 224x224 + 4x64x64 (452 tokens, mask .75)", dict(image_size=224, tactile_size=64, image_patch_size=16, tactile_patch_size=8, dim=384, depth=12, heads=6, mlp_dim=1536, num_tactiles=4),
    dict(decoder_dim=192, masking_ratio=0.75, decoder_depth=4, decoder_heads=3, num_tactiles=4), 64, 3, 224, 64, 4)
run("ref-default 256/4/4 early-conv fs=4 mask .95", dict(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=256, depth=4, heads=4, mlp_dim=512, image_channels=12, tactile_channels=12, num_tactiles=2, frame_stack=4),
    dict(decoder_dim=256, masking_ratio=0.95, decoder_depth=3, decoder_heads=4, num_tactiles=2, early_conv_masking=True, frame_stack=4), 512, 12, 64, 32, 2)
mae = run("cfg5 MAE 70x70 P14 fs=4 384/4/4 dec 384/3/4 mask .8", dict(image_size=70, tactile_size=70, image_patch_size=14, tactile_patch_size=14, dim=384, depth=4, heads=4, mlp_dim=768, image_channels=12, tactile_channels=12, num_tactiles=2, frame_stack=4),
    dict(decoder_dim=384, masking_ratio=0.8, decoder_depth=3, decoder_heads=4, num_tactiles=2, frame_stack=4), 128, 12, 70, 70, 2)
if mae is None:
    sys.exit(0)
dino = DinoV2Frozen().to(dev)
ext = DinoCatMAEExtractor(dino, mae, 384, False, 4).to(dev).eval()
obs = {"image": torch.rand(128, 4, 70, 70, 3, device=dev), "tactile": torch.rand(128, 4, 6, 70, 70, device=dev) * 2 - 1}
with torch.no_grad():
    for _ in range(5): ext(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ext(obs)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    xi = torch.rand(128, 3, 70, 70, device=dev)
    for _ in range(5): dino(xi)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): dino(xi)
    torch.cuda.synchronize(); el2 = time.perf_counter() - t1
print(f"cfg5 extractor forward (get_embeddings + 1-layer tf + frozen DINOv2-S/14-reg + MLP): B=128 {el/20*1e3:.2f} ms  {128*20/el:.0f} obs/s; DINOv2 alone {el2/20*1e3:.2f} ms")
