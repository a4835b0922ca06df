"""DINO-style `VTT` encoder — drop-in for /root/reference/models/VTT.py:77-426 (the forward signature the DINO / iBOT
self-distillation stack of the reference calls: `forward(x, masks) -> x_norm_patchtokens`, `forward_features -> dict`).

Same constructor kwargs, attributes and state-dict keys as the reference class: three separate LayerNorm-Linear-LayerNorm
patch embeds (image, tactile_1, tactile_2), one sinusoidal position table over the stacked (3*H/P, W/P) patch grid sliced
per modality (VTT.py:290-292), keep-index masks applied to every modality with the same indices and concatenated over the
mask list on the batch dimension (`tactile_ssl.utils.apply_masks`), optional register tokens, the vit-pytorch Transformer,
and a final LayerNorm(eps=1e-6).  Weights: trunc-normal(0.02) Linear / ones-zeros LayerNorm (VTT.py:222-228,801-809).

All arithmetic runs in the HIP kernels of m3l_amd/csrc: the fused patch-gather + LN/Linear/LN embed (one call per
modality with its own weights and position slice), the row gather, the transformer stack and the LayerNorm.
torch only concatenates the token blocks.
"""
import math
from functools import partial
from typing import Literal

import torch
from torch import nn
from torch.nn.init import trunc_normal_

from . import _lib as L
from . import functional as Fn
from .pretrain_models import Rearrange, Transformer, pair


class SinusoidalEmbed(nn.Module):
    """tactile_ssl/model/layers/patch_embed.py:133-224 for un-normalised integer grid coordinates: num_bands =
    ceil(D / (2 * ndim)), bands = 10000^-linspace(0,1,nb+1)[:-1], features = grid[..., None] * bands ->
    cat(sin, cos) over the band axis -> flatten -> [:D]; cached."""

    def __init__(self, size, stride, embed_dim=768):
        super().__init__()
        size, stride = list(size), list(stride)
        assert len(size) < 4, "Sinusoidal position embeddings only support 1D, 2D and 3D grids."
        assert len(size) == len(stride), "size and stride must have the same length"
        assert embed_dim % 2 == 0, "Embedding dimension must be divisible by 2"
        self.patches_resolution = [s // stride[i] for i, s in enumerate(size)]
        self.embed_dim = embed_dim
        self.num_bands = math.ceil(embed_dim / (2 * len(size)))
        bands = torch.stack([torch.linspace(0, 1.0, steps=self.num_bands + 1)[:-1] for _ in range(len(size))], dim=0)
        self.register_buffer("frequency_bands", 10000 ** -bands)
        self.register_buffer("cached_encoding", None, persistent=False)

    def forward(self, device, normalized_coords: bool = False):
        if self.cached_encoding is not None:
            return self.cached_encoding if self.cached_encoding.device == torch.device(device) else self.cached_encoding.to(device)
        assert not normalized_coords
        axes = [torch.arange(0, r, dtype=torch.float, device=device) for r in self.patches_resolution]
        grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(axes))
        feats = grid[..., None] * self.frequency_bands.to(device)
        enc = torch.cat([feats.sin(), feats.cos()], dim=-1).flatten(-2, -1)
        self.cached_encoding = enc[..., :self.embed_dim].contiguous()
        return self.cached_encoding


def _init_weights_vit_timm(module: nn.Module):
    if isinstance(module, nn.Linear):
        trunc_normal_(module.weight, std=0.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.LayerNorm):
        nn.init.zeros_(module.bias)
        nn.init.ones_(module.weight)


class VTT(nn.Module):
    def __init__(self, *, image_size, tactile_size, image_patch_size, tactile_patch_size, dim, depth, heads, mlp_dim,
                 image_channels=3, tactile_channels=3, dim_head=64, dropout=0., emb_dropout=0, num_tactiles=2, frame_stack=1,
                 pos_embed_fn: Literal["sinusoidal", "learned"] = "sinusoidal", num_register_tokens: int = 0, num_frames: int = 1,
                 compute_dtype="fp32"):
        super().__init__()
        image_height, image_width = pair(image_size)
        tactile_height, tactile_width = pair(tactile_size)
        image_patch_height, image_patch_width = pair(image_patch_size)
        tactile_patch_height, tactile_patch_width = pair(tactile_patch_size)
        self.image_height, self.image_width = image_height, image_width
        self.tactile_height, self.tactile_width = tactile_height, tactile_width
        self.image_patch_height, self.image_patch_width = image_patch_height, image_patch_width
        self.tactile_patch_height, self.tactile_patch_width = tactile_patch_height, tactile_patch_width
        self.image_channels, self.tactile_channels = image_channels, tactile_channels
        self.frame_stack = frame_stack
        assert image_height % image_patch_height == 0 and image_width % image_patch_width == 0, 'Image dimensions must be divisible by the patch size.'
        assert tactile_height % tactile_patch_height == 0 and tactile_width % tactile_patch_width == 0, 'Tactile dimensions must be divisible by the patch size.'
        if pos_embed_fn != "sinusoidal":
            raise NotImplementedError("only pos_embed_fn='sinusoidal' is defined by the reference class (the learned branch is commented out there)")
        self.num_patches_image = (image_height // image_patch_height) * (image_width // image_patch_width)
        self.num_patches_tactile = (tactile_height // tactile_patch_height) * (tactile_width // tactile_patch_width) * num_tactiles
        self.num_patches = self.num_patches_image + self.num_patches_tactile
        image_patch_dim = image_channels * image_patch_height * image_patch_width
        tactile_patch_dim = tactile_channels * tactile_patch_height * tactile_patch_width

        def embed(p, pd):
            return nn.Sequential(Rearrange(p, p), nn.LayerNorm(pd), nn.Linear(pd, dim), nn.LayerNorm(dim))
        self.image_to_patch_embedding = embed(image_patch_height, image_patch_dim)
        self.tactile_to_patch_embedding_1 = embed(tactile_patch_height, tactile_patch_dim)
        self.tactile_to_patch_embedding_2 = embed(tactile_patch_height, tactile_patch_dim)
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.to_latent = nn.Identity()

        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_register_tokens = num_register_tokens
        assert num_register_tokens >= 0
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, dim)) if num_register_tokens else None
        self.pos_embed_fn = pos_embed_fn
        self.num_frames = num_frames
        self.embed_dim = dim
        self.pos_embed = SinusoidalEmbed([image_height * 3, image_width], [image_patch_height, image_patch_height], embed_dim=dim)
        self.norm = norm_layer(self.embed_dim)
        self.head = nn.Identity()
        self.compute_dtype = compute_dtype
        self.transformer.compute_dtype = compute_dtype
        self.init_weights()

    def init_weights(self):
        if self.register_tokens is not None:
            nn.init.normal_(self.register_tokens, std=1e-6)
        for m in self.modules():
            _init_weights_vit_timm(m)

    def interpolate_pos_encoding(self, img_shape, img_dtype, device):
        return self.pos_embed(device).float().unsqueeze(0)

    # ----------------------------------------------------------------------------------------------------------------
    def _embed_one(self, x, seq, height, width, patch, channels, pos_slice):
        """One modality through the fused HIP embed (all patches): LN -> Linear -> LN, + its slice of the position table."""
        n = (height // patch) * (width // patch)
        if pos_slice.shape[0] != n:
            raise RuntimeError(f"The size of tensor a ({n}) must match the size of tensor b ({pos_slice.shape[0]}) at non-singleton dimension 1")
        geom = L.Geom(height, width, patch, channels, height, width, patch, channels, 0, 1, 0)
        dt = Fn.dtype_code(self.compute_dtype)
        zeros_mod = torch.zeros(1, self.embed_dim, device=x.device)
        tensors = [seq[1].weight, seq[1].bias, seq[2].weight, seq[2].bias, seq[3].weight, seq[3].bias] + [None] * 6 + \
                  [zeros_mod, pos_slice.contiguous(), None]
        return Fn.EmbedFn.apply(None, geom, self.embed_dim, dt, None, n, n, x, [], *tensors)

    def prepare_tokens_with_masks(self, x, masks=None):
        pos = self.interpolate_pos_encoding(x[1].shape, x[1].dtype, device=x[1].device)[0]
        n1 = self.num_patches_image
        n2 = self.num_patches_tactile // 2 if self.num_patches_tactile else 0
        embed1 = self._embed_one(x[0], self.image_to_patch_embedding, self.image_height, self.image_width, self.image_patch_height,
                                 self.image_channels, pos[:n1])
        embed2 = self._embed_one(x[1], self.tactile_to_patch_embedding_1, self.tactile_height, self.tactile_width,
                                 self.tactile_patch_height, self.tactile_channels, pos[n1:n2 * 2])       # VTT.py:291
        embed3 = self._embed_one(x[2], self.tactile_to_patch_embedding_2, self.tactile_height, self.tactile_width,
                                 self.tactile_patch_height, self.tactile_channels, pos[n1 * 2:])         # VTT.py:292
        if masks is not None:
            embed1 = torch.cat([Fn.GatherTokensFn.apply(embed1, m) for m in masks], dim=0)
            embed2 = torch.cat([Fn.GatherTokensFn.apply(embed2, m) for m in masks], dim=0)
            embed3 = torch.cat([Fn.GatherTokensFn.apply(embed3, m) for m in masks], dim=0)
        x = torch.cat([embed1, embed2, embed3], dim=-2)
        if self.register_tokens is not None:
            x = torch.cat((self.register_tokens.expand(x.shape[0], -1, -1), x), dim=1)
        return x

    def forward_features(self, x, masks=None):
        x_list = [x['image'], x['tactile1'], x['tactile2']]
        x = self.prepare_tokens_with_masks(x_list, masks)
        x = self.transformer(x)
        x_norm = Fn.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return {
            "x_norm_regtokens": x_norm[:, : self.num_register_tokens],
            "x_norm_patchtokens": x_norm[:, self.num_register_tokens:],
            "x_prenorm": x,
            "masks": masks,
        }

    def forward(self, *args, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret["x_norm_patchtokens"]
