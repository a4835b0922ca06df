"""Frozen DINOv2 image encoder (ViT with register tokens) on the HIP kernels — the `dino_model` of the reference's cfg-5
fusion head: `torch.hub.load('facebookresearch/dinov2', 'dinov2_vits14_reg')` (train_dino_cat_mae.py:29), called as
`self.dino_model(obs_viso)` -> (B, 384) on the middle RGB frame (models/pretrain_models_dino_cat_mae.py:883-886).

Drop-in surface: same constructor sizes as the hub model, the hub checkpoint's state-dict names (`cls_token`, `pos_embed`,
`register_tokens`, `patch_embed.proj.*`, `blocks.{i}.norm1/attn.qkv/attn.proj/ls1.gamma/norm2/mlp.fc1/mlp.fc2/ls2.gamma`,
`norm.*`), `model(x)` returns the final-norm CLS token (head = Identity).  Inference only: every parameter is frozen
(`requires_grad=False`) as in the reference, and the forward is one C-ABI call (`m3l_frozen_vit_fwd`) after the patch
projection GEMM.  PyTorch only re-tiles the image into patches and concatenates the token rows.

The pretrained weights are not available offline; the module is exact for whatever weights are loaded
(tests: transformers' independent implementation of the same architecture with random weights).
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import functional as Fn


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):          # parameter container with the hub checkpoint's names; arithmetic runs in m3l_frozen_vit_fwd
    def __init__(self, dim, hidden, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attn(dim)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, hidden)
        self.ls2 = _LayerScale(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)


class DinoV2Frozen(nn.Module):
    """ViT-S/14 with 4 registers by default (= dinov2_vits14_reg: embed_dim 384, depth 12, 6 heads of 64, img_size 518)."""

    def __init__(self, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, patch_size=14, img_size=518, in_chans=3,
                 num_register_tokens=4, layer_norm_eps=1e-6, compute_dtype="bf16"):
        super().__init__()
        if embed_dim != 64 * num_heads:
            raise NotImplementedError("m3l_amd attention kernels are built for dim_head = 64 (embed_dim == 64 * num_heads)")
        self.embed_dim, self.depth, self.num_heads, self.patch_size = embed_dim, depth, num_heads, patch_size
        self.num_register_tokens, self.eps = num_register_tokens, layer_norm_eps
        self.mlp_dim = int(embed_dim * mlp_ratio)
        grid = img_size // patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + grid * grid, embed_dim))
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))          # checkpoint compatibility; unused at inference
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        self.blocks = nn.ModuleList([_Block(embed_dim, self.mlp_dim, layer_norm_eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=layer_norm_eps)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()
        self.compute_dtype = compute_dtype
        self._prepared = None         # (key, tensors kept alive, pointer array, patch weight, patch bias)
        self._pos_cache = {}

    def set_compute_dtype(self, compute_dtype):
        self.compute_dtype, self._prepared = compute_dtype, None

    # ---- frozen weights -> compute-type operands, once (re-done if a parameter is replaced / modified in place) ---------
    def _key(self):
        return (self.compute_dtype, str(self.cls_token.device)) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _prepare(self):
        key = self._key()
        if self._prepared is not None and self._prepared[0] == key:
            return self._prepared
        T = Fn.tdtype(Fn.dtype_code(self.compute_dtype))
        keep, ptrs = [], []

        def add(t, dtype=torch.float32):
            t = t.detach().to(dtype).contiguous()
            keep.append(t)
            ptrs.append(t)

        for b in self.blocks:
            add(b.norm1.weight); add(b.norm1.bias)
            add(b.attn.qkv.weight, T); add(b.attn.qkv.bias)
            g1, g2 = b.ls1.gamma.detach().float(), b.ls2.gamma.detach().float()
            add(g1[:, None] * b.attn.proj.weight.detach().float(), T); add(g1 * b.attn.proj.bias.detach().float())   # LayerScale folded
            add(b.norm2.weight); add(b.norm2.bias)
            add(b.mlp.fc1.weight, T); add(b.mlp.fc1.bias)
            add(g2[:, None] * b.mlp.fc2.weight.detach().float(), T); add(g2 * b.mlp.fc2.bias.detach().float())
        add(self.norm.weight); add(self.norm.bias)
        # patch projection as an NT GEMM: W[D, (c, kh, kw)] with K padded to a multiple of 8
        w = self.patch_embed.proj.weight.detach().float().flatten(1)
        kp = (w.shape[1] + 7) // 8 * 8
        wp = F.pad(w, (0, kp - w.shape[1])).to(T).contiguous()
        pb = self.patch_embed.proj.bias.detach().float().contiguous()
        self._prepared = (key, keep, L.ptr_array(ptrs), wp, pb, kp)
        self._pos_cache = {}
        return self._prepared

    def _pos(self, gh, gw):
        k = (gh, gw, self.pos_embed.data_ptr(), self.pos_embed._version)
        if k not in self._pos_cache:
            n = self.pos_embed.shape[1] - 1
            G = int(round(n ** 0.5))
            pe = self.pos_embed.detach().float()
            if not (gh == G and gw == G):
                # hub *_reg models: interpolate_offset = 0.0 -> explicit size, bicubic, antialias (one-time weight transform)
                patch = pe[:, 1:].reshape(1, G, G, -1).permute(0, 3, 1, 2)
                patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False, antialias=True)
                pe = torch.cat((pe[:, :1], patch.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)), dim=1)
            self._pos_cache = {k: pe.contiguous()}
        return self._pos_cache[k]

    @torch.no_grad()
    def forward_features(self, x):
        Fn._require_cuda(x, "DinoV2Frozen input")
        Fn._require_cuda(self.cls_token, "DinoV2Frozen parameters")
        B, Cc, H, W = x.shape
        P, D = self.patch_size, self.embed_dim
        assert H % P == 0 and W % P == 0, f"input {H}x{W} is not a multiple of the patch size {P}"
        gh, gw = H // P, W // P
        _, keep, ptrs, wp, pb, kp = self._prepare()
        dt = Fn.dtype_code(self.compute_dtype)
        T = Fn.tdtype(dt)
        # patches in the Conv2d weight's K order (c, kh, kw), zero-padded to the GEMM's K: one HIP gather (m3l_op_patch_cols)
        xf = x.detach().float().contiguous()
        pt = torch.empty(B * gh * gw, kp, dtype=T, device=x.device)
        L.check(L.lib().m3l_op_patch_cols(dt, L.ptr(xf), B, Cc, H, W, P, kp, L.ptr(pt), Fn._stream()), "m3l_op_patch_cols")
        emb = torch.empty(B * gh * gw, D, dtype=torch.float32, device=x.device)
        L.check(L.lib().m3l_op_gemm_nt(dt, L.ptr(pt), kp, L.ptr(wp), kp, B * gh * gw, D, kp, L.ptr(pb), None, L.ptr(emb), None, None,
                                       None, 0, D, Fn._stream()), "patch_embed")
        pos = self._pos(gh, gw)
        R = self.num_register_tokens
        n = 1 + R + gh * gw
        tok = torch.empty(B, n, D, dtype=torch.float32, device=x.device)
        cls, regs = self.cls_token.detach().float().contiguous(), self.register_tokens.detach().float().contiguous()
        L.check(L.lib().m3l_op_vit_tokens(L.ptr(emb), L.ptr(cls), L.ptr(regs) if R else None, L.ptr(pos), B, gh * gw, R, D, L.ptr(tok), Fn._stream()),
                "m3l_op_vit_tokens")
        cfg = L.TfCfg(D, self.depth, self.num_heads, self.mlp_dim, 1, dt)
        ws = Fn._ws(L.lib().m3l_frozen_vit_ws_bytes(C.byref(cfg), B, n), x.device)
        y = torch.empty(B, n, D, dtype=torch.float32, device=x.device)
        L.check(L.lib().m3l_frozen_vit_fwd(C.byref(cfg), float(self.eps), B, n, L.ptr(tok), ptrs, L.ptr(ws), L.ptr(y), Fn._stream()),
                "m3l_frozen_vit_fwd")
        return {"x_norm_clstoken": y[:, 0], "x_norm_regtokens": y[:, 1:1 + R], "x_norm_patchtokens": y[:, 1 + R:], "x_prenorm": None,
                "tokens_in": tok}

    def forward(self, x):
        """(B, 3, H, W) -> (B, embed_dim): the final-norm CLS token, what the hub model's forward returns (head = Identity)."""
        return self.forward_features(x)["x_norm_clstoken"]
