"""CPU restatement of the frozen DINOv2 (ViT with registers) image encoder that the cfg-5 fusion head calls
(`self.dino_model(obs_viso)`, /root/reference/models/pretrain_models_dino_cat_mae.py:886; the model object comes from
`torch.hub.load('facebookresearch/dinov2', 'dinov2_vits14_reg')`, train_dino_cat_mae.py:29).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under m3l_amd/).

The network is a third-party dependency that is absent from /root/reference (fetched by torch.hub at run time; no network
here, and its pretrained weights are unavailable): facebookresearch/dinov2 `DinoVisionTransformer` (vit_small, patch 14,
4 register tokens, LayerScale, `interpolate_antialias=True`, `interpolate_offset=0.0`).  Its published algorithm is
restated below with the hub checkpoint's parameter names.  PINNING: checked against tests/golden/dinov2_small.npz, produced
by the independent implementation that IS installed in this image — `transformers.Dinov2WithRegistersModel` built from a
config object with random weights (tests/golden/make_golden_dinov2.py) — i.e. pinned to a second implementation of the same
published architecture, not to the hub code itself and not to pretrained weights: "parity unpinned" with respect to the
reference's own run (cfg 5 is measured with random-init frozen weights).
"""
from typing import Dict

import torch
import torch.nn.functional as F


def interpolate_pos_embed(pos_embed: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """(1, 1 + G*G, D) -> (1, 1 + gh*gw, D): bicubic, antialias, align_corners=False, explicit size (the *_reg hub models:
    interpolate_offset = 0.0, interpolate_antialias = True).  Identity when the grid already matches."""
    n = pos_embed.shape[1] - 1
    G = int(round(n ** 0.5))
    if G * G == n and gh == G and gw == G:
        return pos_embed
    D = pos_embed.shape[-1]
    patch = pos_embed[:, 1:].reshape(1, G, G, D).permute(0, 3, 1, 2).float()
    patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False, antialias=True)
    return torch.cat((pos_embed[:, :1], patch.permute(0, 2, 3, 1).reshape(1, gh * gw, D).to(pos_embed.dtype)), dim=1)


def prepare_tokens(P: Dict[str, torch.Tensor], x: torch.Tensor, patch: int) -> torch.Tensor:
    """patch conv (stride = kernel = patch) -> [cls | patches] + positions -> registers inserted after cls."""
    B, _, H, W = x.shape
    t = F.conv2d(x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    t = torch.cat((P["cls_token"].expand(B, -1, -1), t), dim=1) + interpolate_pos_embed(P["pos_embed"], H // patch, W // patch)
    return torch.cat((t[:, :1], P["register_tokens"].expand(B, -1, -1), t[:, 1:]), dim=1)


def dinov2_forward(P: Dict[str, torch.Tensor], x: torch.Tensor, *, patch: int, depth: int, heads: int, eps: float = 1e-6):
    """Returns {'tokens_in', 'x_norm' (B, n, D), 'cls' (B, D) = what `model(x)` returns (head = Identity)}."""
    t = prepare_tokens(P, x, patch)
    out = {"tokens_in": t}
    B, n, D = t.shape
    dh = D // heads
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (D,), P[p + "norm1.weight"], P[p + "norm1.bias"], eps)
        qkv = (h @ P[p + "attn.qkv.weight"].t() + P[p + "attn.qkv.bias"]).reshape(B, n, 3, heads, dh).permute(2, 0, 3, 1, 4)
        att = ((qkv[0] * dh ** -0.5) @ qkv[1].transpose(-1, -2)).softmax(-1) @ qkv[2]
        a = att.transpose(1, 2).reshape(B, n, D) @ P[p + "attn.proj.weight"].t() + P[p + "attn.proj.bias"]
        t = t + P[p + "ls1.gamma"] * a
        h = F.layer_norm(t, (D,), P[p + "norm2.weight"], P[p + "norm2.bias"], eps)
        m = F.gelu(h @ P[p + "mlp.fc1.weight"].t() + P[p + "mlp.fc1.bias"]) @ P[p + "mlp.fc2.weight"].t() + P[p + "mlp.fc2.bias"]
        t = t + P[p + "ls2.gamma"] * m
    xn = F.layer_norm(t, (D,), P["norm.weight"], P["norm.bias"], eps)
    out["x_norm"] = xn
    out["cls"] = xn[:, 0]
    return out


def hf_to_hub_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers.Dinov2WithRegistersModel.state_dict() -> facebookresearch/dinov2 hub checkpoint names."""
    out = {"cls_token": sd["embeddings.cls_token"], "mask_token": sd["embeddings.mask_token"],
           "register_tokens": sd["embeddings.register_tokens"], "pos_embed": sd["embeddings.position_embeddings"],
           "patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"],
           "patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"],
           "norm.weight": sd["layernorm.weight"], "norm.bias": sd["layernorm.bias"]}
    i = 0
    while f"encoder.layer.{i}.norm1.weight" in sd:
        s, d = f"encoder.layer.{i}.", f"blocks.{i}."
        for nm in ("norm1", "norm2", "mlp.fc1", "mlp.fc2"):
            out[d + nm + ".weight"], out[d + nm + ".bias"] = sd[s + nm + ".weight"], sd[s + nm + ".bias"]
        for wb in ("weight", "bias"):
            out[d + "attn.qkv." + wb] = torch.cat([sd[s + f"attention.attention.{k}.{wb}"] for k in ("query", "key", "value")], 0)
            out[d + "attn.proj." + wb] = sd[s + "attention.output.dense." + wb]
        out[d + "ls1.gamma"], out[d + "ls2.gamma"] = sd[s + "layer_scale1.lambda1"], sd[s + "layer_scale2.lambda1"]
        i += 1
    return out
