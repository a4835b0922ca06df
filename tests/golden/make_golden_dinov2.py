#!/usr/bin/env python3
"""Golden vectors for the frozen DINOv2-with-registers encoder of the cfg-5 fusion head.

The reference obtains the network from torch.hub (facebookresearch/dinov2, `dinov2_vits14_reg`): not fetchable here.  The same
published architecture is available locally as `transformers.Dinov2WithRegistersModel` (an installed third-party library, not
reference code), built from a config object with RANDOM weights.  This script runs that implementation on a
small configuration (hidden 64, 2 layers, 1 head of 64, patch 14, 4 registers, position table for 98x98 so that a 70x70 input
exercises the bicubic/antialias interpolation) and records parameters (under the hub checkpoint's names), input and outputs.

    python tests/golden/make_golden_dinov2.py      ->  tests/golden/dinov2_small.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.dinov2_oracle import hf_to_hub_state_dict  # noqa: E402


def main():
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    cfg = Dinov2WithRegistersConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, patch_size=14, image_size=98,
                                    num_register_tokens=4, mlp_ratio=4)
    torch.manual_seed(41)
    m = Dinov2WithRegistersModel(cfg).eval()
    g = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for n, p in m.named_parameters():          # make every bias / scale / token non-trivial
            if p.dim() == 1 or "token" in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g) + (1.0 if ("norm" in n and n.endswith("weight")) or "lambda" in n else 0.0))
    x = torch.rand(2, 3, 70, 70, generator=g)
    with torch.no_grad():
        emb = m.embeddings(x)
        o = m(x)
    out = {"param/" + k: v.numpy() for k, v in hf_to_hub_state_dict(m.state_dict()).items()}
    out["input/x"] = x.numpy()
    out["out/tokens_in"] = emb.numpy()
    out["out/x_norm"] = o.last_hidden_state.numpy()
    out["out/cls"] = o.pooler_output.numpy()
    out["meta"] = np.array([64, 2, 1, 14, 98, 4], dtype=np.int64)   # dim, depth, heads, patch, img_size, registers
    np.savez_compressed(os.path.join(HERE, "dinov2_small.npz"), **out)
    print("dinov2_small:", tuple(o.last_hidden_state.shape), "cls", o.pooler_output[0, :4].tolist())


if __name__ == "__main__":
    main()
