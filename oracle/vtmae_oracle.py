"""CPU ORACLE for the M3L masked multimodal auto-encoder hot path.  *** TEST INFRASTRUCTURE ONLY ***

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file, and
only as the checker / reported CPU baseline.  The product (`m3l_amd/`) never imports it and has no CPU
fallback: it raises when the HIP extension is missing.

What it restates (plain PyTorch on CPU, fp32 or fp64, autograd for the backward):

  vt_load            /root/reference/utils/pretrain_utils.py:7-57
  patchify           /root/reference/models/pretrain_models.py:768,775   (einops Rearrange)
  patch embed        :769-771,776-778  (LayerNorm -> Linear -> LayerNorm, eps 1e-5)
  modality + sincos  :120-143,202-214
  mask sampling      :223-248   (Python-double int() truncation; argsort of injected noise, STABLE ascending)
  gathers            :255-262
  transformer        vit-pytorch==1.6.4 `vit_pytorch.vit.Transformer`  (third party, NOT in /root/reference)
  un-shuffle         :279-307
  heads + loss       :327-340
  get_embeddings     :588-668
  EarlyCNN stem      :37-56,180-191,311-322   (early_conv_masking=True, the reference's default flag)
  SinusoidalEmbed    /root/reference/tactile_ssl/model/layers/patch_embed.py:133-224 (+ create_ndgrid utils/__init__.py:39-69)
  apply_masks        /root/reference/tactile_ssl/utils/__init__.py:25-36
  DINO-style VTT     /root/reference/models/VTT.py:280-360,424-426

PINNING.  Checked (tests/test_oracle_golden.py) against fixtures in tests/golden/*.npz produced by
EXECUTING the reference's own `VTT`/`VTMAE`/`vt_load` code in the build container
(tests/golden/make_golden.py).  The reference has no numeric tests or golden vectors of its own
(SURVEY.md section 4).  The two arithmetic third-party dependencies (vit-pytorch 1.6.4 Transformer,
positional-encodings 6.0.1 PositionalEncoding2D) are absent from the image and from /root/reference;
they are restated from their published algorithm on BOTH sides of the fixture comparison, so that part
is **parity unpinned** (nothing executable pins it to the real wheels).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class OracleCfg:
    image_hw: int
    tactile_hw: int
    image_patch: int
    tactile_patch: int
    dim: int
    depth: int
    heads: int
    mlp: int
    channels: int
    num_tactiles: int
    dec_dim: int
    dec_depth: int
    dec_heads: int
    ratio: float
    dim_head: int = 64
    dec_dim_head: int = 64

    @property
    def n_img(self):
        return (self.image_hw // self.image_patch) ** 2

    @property
    def n_tac(self):
        return (self.tactile_hw // self.tactile_patch) ** 2

    @property
    def n_total(self):
        return self.n_img + self.num_tactiles * self.n_tac


# ----------------------------------------------------------------------------------------------------
# integer / index work (numpy)
# ----------------------------------------------------------------------------------------------------
def mask_counts(ratio: float, n_img: int, n_tac_total: int, num_tactiles: int):
    """pretrain_models.py:223-227 with Python double semantics and int() truncation."""
    n = n_img + n_tac_total
    num_masked = int(ratio * n)
    image_perc = n_img / n
    nm_img = int(num_masked * image_perc)
    nm_tac = (num_masked - nm_img) // num_tactiles if (num_tactiles > 0 and n_tac_total > 0) else 0
    return num_masked, nm_img, nm_tac


def stable_argsort_rows(noise: np.ndarray) -> np.ndarray:
    """Ascending, ties broken by ascending original index (contract of the HIP mask kernel)."""
    return np.argsort(noise, axis=-1, kind="stable").astype(np.int64)


def mask_indices(noises: List[np.ndarray], ratio: float, n_img: int, n_tac: int, num_tactiles: int, perms=None, counts=None):
    """pretrain_models.py:229-248.  noises = [image (B,n_img), tactile_1 (B,n_tac), ...] in RNG order.
    Returns (masked_indices, unmasked_indices, nm_img, nm_tac) as int64 (B, *).

    Tie-break contract: STABLE ascending (== torch.argsort(stable=True)).  The reference calls
    `torch.rand(...).argsort(dim=-1)` WITHOUT stable=True; on float32 ties torch's default sort is not stable
    (observed here on 64-wide rows, and its CUDA kernel differs again), i.e. the reference's own order of tied
    keys is unspecified.  `perms` lets a test inject the permutations a particular reference run produced."""
    _, nm_img, nm_tac = mask_counts(ratio, n_img, n_tac * num_tactiles, num_tactiles)
    if counts is not None:          # reconstruct() has its own count rule (pretrain_models.py:425,433)
        nm_img, nm_tac = counts
    perm = stable_argsort_rows(noises[0]) if perms is None else np.asarray(perms[0], dtype=np.int64)
    masked = [perm[:, :nm_img]]
    unmasked = [perm[:, nm_img:]]
    count = n_img
    for i in range(num_tactiles):
        p = (stable_argsort_rows(noises[1 + i]) if perms is None else np.asarray(perms[1 + i], dtype=np.int64)) + count
        masked.append(p[:, :nm_tac])
        unmasked.append(p[:, nm_tac:])
        count += n_tac
    return np.concatenate(masked, 1), np.concatenate(unmasked, 1), nm_img, nm_tac


# ----------------------------------------------------------------------------------------------------
# float work (torch CPU)
# ----------------------------------------------------------------------------------------------------
def vt_load(obs: Dict[str, np.ndarray], frame_stack: int = 1, image_normalization=(0, 1), tactile_normalization=(-1, 1)) -> Dict[str, torch.Tensor]:
    """utils/pretrain_utils.py:7-57: NHWC -> NCHW, per-sensor channel pick, (x - lo) / (hi - lo) in fp32 (:28-30,47-49)."""
    out = {}
    if "image" in obs:
        img = np.asarray(obs["image"])
        if img.ndim == 3:
            img = img[None]
        assert img.shape[-1] == 3 * frame_stack
        t = torch.tensor(img, dtype=torch.float32).permute(0, 3, 1, 2)
        out["image"] = (t - image_normalization[0]) / (image_normalization[1] - image_normalization[0])
    if "tactile" in obs:
        tac = np.asarray(obs["tactile"])
        if tac.ndim == 3:
            tac = tac[None]
        assert tac.shape[1] in (3 * frame_stack, 6 * frame_stack, 12 * frame_stack)
        n_tactiles = tac.shape[1] // frame_stack
        idx = np.array([i * n_tactiles + c for i in range(frame_stack) for c in range(3)])
        for s in range(n_tactiles // 3):
            t = torch.tensor(tac[:, idx + 3 * s], dtype=torch.float32)
            out[f"tactile{s + 1}"] = (t - tactile_normalization[0]) / (tactile_normalization[1] - tactile_normalization[0])
    return out


def patchify(x: torch.Tensor, p: int) -> torch.Tensor:
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'"""
    B, C, H, W = x.shape
    x = x.reshape(B, C, H // p, p, W // p, p)
    return x.permute(0, 2, 4, 3, 5, 1).reshape(B, (H // p) * (W // p), p * p * C)


def sincos_2d(channels_param: int, gh: int, gw: int, out_ch: int, dtype=torch.float32) -> torch.Tensor:
    """positional-encodings 6.0.1 PositionalEncoding2D(channels_param) applied to a (1,gh,gw,out_ch) tensor,
    flattened to (gh*gw, out_ch).  Interleaved sin/cos, x-code then y-code, truncated to out_ch."""
    ch = int(math.ceil(channels_param / 4) * 2)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))

    def emb(n):
        s = torch.einsum("i,j->ij", torch.arange(n).float(), inv_freq)
        return torch.stack((s.sin(), s.cos()), dim=-1).flatten(-2, -1)
    e = torch.zeros(gh, gw, 2 * ch)
    e[:, :, :ch] = emb(gh).unsqueeze(1)
    e[:, :, ch:] = emb(gw)
    return e[:, :, :out_ch].reshape(gh * gw, -1).to(dtype)


def transformer(x, P: Dict[str, torch.Tensor], prefix: str, depth: int, heads: int, dim_head: int,
                final_norm: bool = True):
    """vit-pytorch 1.6.4 Transformer forward (pre-norm attention + GELU FFN, final LayerNorm)."""
    B, n, D = x.shape
    for i in range(depth):
        a = f"{prefix}layers.{i}.0."
        f = f"{prefix}layers.{i}.1.net."
        h = F.layer_norm(x, (D,), P[a + "norm.weight"], P[a + "norm.bias"], 1e-5)
        qkv = h @ P[a + "to_qkv.weight"].t()
        q, k, v = [t.reshape(B, n, heads, dim_head).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
        dots = (q @ k.transpose(-1, -2)) * (dim_head ** -0.5)
        o = (dots.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, n, heads * dim_head)
        if (a + "to_out.0.weight") in P:
            o = o @ P[a + "to_out.0.weight"].t() + P[a + "to_out.0.bias"]
        x = o + x
        h = F.layer_norm(x, (D,), P[f + "0.weight"], P[f + "0.bias"], 1e-5)
        h = F.gelu(h @ P[f + "1.weight"].t() + P[f + "1.bias"])
        x = h @ P[f + "4.weight"].t() + P[f + "4.bias"] + x
    if final_norm:
        x = F.layer_norm(x, (D,), P[prefix + "norm.weight"], P[prefix + "norm.bias"], 1e-5)
    return x


def _embed(patches, P, key):
    pd = patches.shape[-1]
    h = F.layer_norm(patches, (pd,), P[key + ".1.weight"], P[key + ".1.bias"], 1e-5)
    h = h @ P[key + ".2.weight"].t() + P[key + ".2.bias"]
    return F.layer_norm(h, (h.shape[-1],), P[key + ".3.weight"], P[key + ".3.bias"], 1e-5)


def early_cnn(x, P, prefix: str, key: str):
    """EarlyCNN.forward (pretrain_models.py:37-56): conv 4/2/1 + ReLU, conv 4/2/1 + ReLU, conv 4/2/1 (image) or 3/1/1 (tactile)
    + ReLU, conv 1x1; flatten(2).transpose(1, 2) -> (B, h*w, D)."""
    x = F.relu(F.conv2d(x, P[prefix + "conv1.weight"], P[prefix + "conv1.bias"], stride=2, padding=1))
    x = F.relu(F.conv2d(x, P[prefix + "conv2.weight"], P[prefix + "conv2.bias"], stride=2, padding=1))
    if key == "image":
        x = F.relu(F.conv2d(x, P[prefix + "conv3.weight"], P[prefix + "conv3.bias"], stride=2, padding=1))
    else:
        x = F.relu(F.conv2d(x, P[prefix + "conv3.weight"], P[prefix + "conv3.bias"], stride=1, padding=1))
    return F.conv2d(x, P[prefix + "conv4.weight"], P[prefix + "conv4.bias"]).flatten(2).transpose(1, 2)


def encoder_tokens(P, cfg: OracleCfg, x: Dict[str, torch.Tensor], use_vision=True, use_tactile=True, sincos=True):
    """patchify + embed (or EarlyCNN stem when the parameters hold one) + modality + sincos for ALL patches
    (pretrain_models.py:154-216); sincos=False is `use_sincosmod_encodings=False`: no modality / sincos terms, the learned
    `encoder.pos_embedding[:, 1:num_patches + 1]` added to the concatenated tokens instead (:218-219)."""
    dt = P["mask_token"].dtype
    early = "early_conv_vision.conv1.weight" in P
    toks, img_patches, tac_patches = [], None, None
    if use_vision:
        img_patches = patchify(x["image"].to(dt), cfg.image_patch)
        t = early_cnn(x["image"].to(dt), P, "early_conv_vision.", "image") if early else _embed(img_patches, P, "encoder.image_to_patch_embedding")
        if sincos:
            t = t + P["encoder_modality_embedding.weight"][0] + P["image_enc_pos_embedding"][0]
        toks.append(t)
    if cfg.num_tactiles > 0 and use_tactile:
        tac_patches = torch.cat([patchify(x[f"tactile{i + 1}"].to(dt), cfg.tactile_patch) for i in range(cfg.num_tactiles)], 1)
        if early:
            t = torch.cat([early_cnn(x[f"tactile{i + 1}"].to(dt), P, "early_conv_tactile.", "tactile") for i in range(cfg.num_tactiles)], 1)
        else:
            t = _embed(tac_patches, P, "encoder.tactile_to_patch_embedding")
        if sincos:
            mod = P["encoder_modality_embedding.weight"][1:1 + cfg.num_tactiles].repeat_interleave(cfg.n_tac, 0)
            t = t + mod + P["tactile_enc_pos_embedding"][0]
        toks.append(t)
    tokens = torch.cat(toks, 1)
    if not sincos:
        tokens = tokens + P["encoder.pos_embedding"][:, 1:tokens.shape[1] + 1]
    return tokens, img_patches, tac_patches


def vtmae_forward(P: Dict[str, torch.Tensor], cfg: OracleCfg, x: Dict[str, torch.Tensor],
                  noises: List[torch.Tensor], use_vision=True, use_tactile=True, perms=None, counts=None,
                  sincos=True) -> Dict[str, torch.Tensor]:
    """VTMAE.forward (pretrain_models.py:146-342).  sincos=False = `use_sincosmod_encodings=False`: learned encoder positions
    (:218-219) and `decoder_pos_emb` at every decoder position (:280-281,286-287) instead of the modality + sincos terms.
    Returns every intermediate the fixtures record plus 'loss'."""
    nt = cfg.num_tactiles if use_tactile else 0
    n_img = cfg.n_img if use_vision else 0
    tokens, img_patches, tac_patches = encoder_tokens(P, cfg, x, use_vision, use_tactile, sincos)
    B = tokens.shape[0]
    masked, unmasked, nm_img, nm_tac = mask_indices([np.asarray(z) for z in noises], cfg.ratio, n_img, cfg.n_tac, nt, perms, counts)
    masked_t, unmasked_t = torch.from_numpy(masked), torch.from_numpy(unmasked)
    br = torch.arange(B)[:, None]
    vis = tokens[br, unmasked_t]
    out = {"masked_indices": masked_t, "unmasked_indices": unmasked_t, "tokens_all": tokens, "encoder_in": vis}
    enc = transformer(vis, P, "encoder.transformer.", cfg.depth, cfg.heads, cfg.dim_head)
    out["encoder_out"] = enc
    dec_in = enc @ P["enc_to_dec.weight"].t() + P["enc_to_dec.bias"] if "enc_to_dec.weight" in P else enc
    N = tokens.shape[1]
    full = torch.zeros(B, N, cfg.dec_dim, dtype=tokens.dtype)
    full = full.index_put((br, unmasked_t), dec_in)
    full = full.index_put((br, masked_t), P["mask_token"].expand(B, masked_t.shape[1], -1))
    if sincos:
        add = []
        if use_vision:
            add.append(P["decoder_modality_embedding.weight"][0] + P["image_dec_pos_embedding"][0])
        if nt > 0:
            add.append(P["decoder_modality_embedding.weight"][1:1 + nt].repeat_interleave(cfg.n_tac, 0) + P["tactile_dec_pos_embedding"][0])
        full = full + torch.cat(add, 0)
    else:
        full = full + P["decoder_pos_emb.weight"][:N]       # decoder_pos_emb(unmasked_indices) / (masked_indices): every position j gets row j
    out["decoder_in"] = full
    dec = transformer(full, P, "decoder.", cfg.dec_depth, cfg.dec_heads, cfg.dec_dim_head)
    out["decoder_out"] = dec
    loss = torch.zeros((), dtype=tokens.dtype)
    mi_img, mi_tac = masked_t[:, :nm_img], masked_t[:, nm_img:]
    if "early_conv_vision.conv1.weight" in P:
        # early_conv_masking=True (pretrain_models.py:311-322): predict ALL patches, loss over all of them
        if nt > 0:
            pred_t = dec[:, n_img:] @ P["to_tactiles.weight"].t() + P["to_tactiles.bias"]
            out["pred_tactile"], out["target_tactile"] = pred_t, tac_patches
            loss = loss + 10 * F.mse_loss(pred_t, tac_patches)
        if use_vision:
            pred_i = dec[:, :n_img] @ P["to_pixels.weight"].t() + P["to_pixels.bias"]
            out["pred_pixel"], out["target_pixel"] = pred_i, img_patches
            loss = loss + F.mse_loss(pred_i, img_patches)
        out["loss"] = loss
        return out
    if nt > 0:
        pred_t = dec[br, mi_tac] @ P["to_tactiles.weight"].t() + P["to_tactiles.bias"]
        tgt_t = tac_patches[br, mi_tac - n_img]
        out["pred_tactile"], out["target_tactile"] = pred_t, tgt_t
        loss = loss + 10 * F.mse_loss(pred_t, tgt_t)
    if use_vision:
        pred_i = dec[br, mi_img] @ P["to_pixels.weight"].t() + P["to_pixels.bias"]
        tgt_i = img_patches[br, mi_img]
        out["pred_pixel"], out["target_pixel"] = pred_i, tgt_i
        loss = loss + F.mse_loss(pred_i, tgt_i)
    out["loss"] = loss
    return out


def _frames(p, n, gh, gw, ph, pw):
    """'b (n h w) (p1 p2 c) -> b (n c) (h p1) (w p2)' (pretrain_models.py:464,476)."""
    b, _, pd = p.shape
    c = pd // (ph * pw)
    return p.reshape(b, n, gh, gw, ph, pw, c).permute(0, 1, 6, 2, 4, 3, 5).reshape(b, n * c, gh * ph, gw * pw)


def reconstruct(P, cfg: OracleCfg, x, noises, mask_ratio=None, use_vision=True, use_tactile=True):
    """VTMAE.reconstruct (pretrain_models.py:344-586): masked counts int(r * n_img) / int(r * n_tac_total / k)
    (:425,:433), frames with masked patches = prediction ('*_rec') or 0.5 / inf ('*_masked'), unweighted MSEs."""
    r = cfg.ratio if mask_ratio is None else mask_ratio
    nt = cfg.num_tactiles if use_tactile else 0
    n_img = cfg.n_img if use_vision else 0
    nm_img = int(r * n_img)
    nm_tac = int(r * (nt * cfg.n_tac) / nt) if nt else 0
    o = vtmae_forward(P, cfg, x, noises, use_vision, use_tactile, counts=(nm_img, nm_tac))
    early = "early_conv_vision.conv1.weight" in P
    B = o["masked_indices"].shape[0]
    br = torch.arange(B)[:, None]
    mi_img, mi_tac = o["masked_indices"][:, :nm_img], o["masked_indices"][:, nm_img:] - n_img
    out = {}
    if use_vision:
        g, ph = cfg.image_hw // cfg.image_patch, cfg.image_patch
        patches = patchify(x["image"], ph)
        vis = patches.clone()
        vis[br, mi_img] = 0.5
        rec = o["pred_pixel"] if early else patches.index_put((br, mi_img), o["pred_pixel"])
        out["image_rec"], out["image_masked"] = _frames(rec, 1, g, g, ph, ph), _frames(vis, 1, g, g, ph, ph)
        out["recon_loss_image"] = F.mse_loss(o["pred_pixel"], o["target_pixel"])
    if nt:
        g, ph = cfg.tactile_hw // cfg.tactile_patch, cfg.tactile_patch
        patches = torch.cat([patchify(x[f"tactile{i + 1}"], ph) for i in range(nt)], 1)
        vis = patches.clone()
        vis[br, mi_tac] = float("inf")
        rec = o["pred_tactile"] if early else patches.index_put((br, mi_tac), o["pred_tactile"])
        out["tactile_rec"], out["tactile_masked"] = _frames(rec, nt, g, g, ph, ph), _frames(vis, nt, g, g, ph, ph)
        out["recon_loss_tactile"] = F.mse_loss(o["pred_tactile"], o["target_tactile"])
    return out


def get_embeddings(P, cfg: OracleCfg, x, use_vision=True, use_tactile=True, sincos=True):
    """VTMAE.get_embeddings (pretrain_models.py:588-668): encoder over ALL tokens, no masking."""
    tokens, _, _ = encoder_tokens(P, cfg, x, use_vision, use_tactile, sincos)
    return transformer(tokens, P, "encoder.transformer.", cfg.depth, cfg.heads, cfg.dim_head)


# ----------------------------------------------------------------------------------------------------
# DINO-style VTT path (models/VTT.py) pieces
# ----------------------------------------------------------------------------------------------------
def sinusoidal_embed(grid_hw, embed_dim: int) -> torch.Tensor:
    """tactile_ssl/model/layers/patch_embed.py:133-224 for a 2-D grid with un-normalised integer coords:
    nb = ceil(D/4); bands = 10000^-linspace(0,1,nb+1)[:-1]; feats = grid[...,None]*bands -> cat(sin,cos)
    over the band axis -> flatten (coord-major) -> [:D]."""
    gh, gw = grid_hw
    nb = int(math.ceil(embed_dim / 4))
    bands = 10000.0 ** (-torch.linspace(0, 1, nb + 1)[:-1])
    ys, xs = torch.meshgrid(torch.arange(gh).float(), torch.arange(gw).float(), indexing="ij")
    grid = torch.stack([ys, xs], -1)                       # (gh,gw,2)
    feats = grid[..., None] * bands                        # (gh,gw,2,nb)
    emb = torch.cat([feats.sin(), feats.cos()], dim=-1)    # (gh,gw,2,2nb)
    return emb.flatten(-2).reshape(gh * gw, -1)[:, :embed_dim]


def apply_masks(x: torch.Tensor, masks: List[torch.Tensor]) -> torch.Tensor:
    """tactile_ssl/utils/__init__.py:25-36: gather kept indices per mask, concat over masks on batch."""
    return torch.cat([torch.gather(x, 1, m[:, :, None].expand(-1, -1, x.shape[-1])) for m in masks], 0)


def load_fixture_params(npz, dtype=torch.float32, requires_grad=False) -> Dict[str, torch.Tensor]:
    P = {}
    for k in npz.files:
        if k.startswith("param/"):
            t = torch.tensor(npz[k]).to(dtype)
            if requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            P[k[len("param/"):]] = t
    return P


def cfg_from_meta(meta, ratio) -> OracleCfg:
    m = [int(v) for v in meta]
    return OracleCfg(image_hw=m[0], tactile_hw=m[1], image_patch=m[2], tactile_patch=m[3], dim=m[4], depth=m[5],
                     heads=m[6], mlp=m[7], channels=m[8], num_tactiles=m[9], dec_dim=m[10], dec_depth=m[11],
                     dec_heads=m[12], ratio=float(ratio))


def vtt_dino_forward(P, *, image_patch: int, tactile_patch: int, depth: int, heads: int, x, masks=None,
                     dim_head: int = 64):
    """models/VTT.py:280-360: three separate LN-Linear-LN embeds, one sinusoidal table over the stacked
    (3*H/P, W/P) grid sliced per modality (:290-292), optional keep-index masks applied to each modality with the
    SAME indices (:299-301) and concatenated over the mask list on batch, vit-pytorch Transformer (with its own
    final LayerNorm), then `self.norm` = LayerNorm(eps=1e-6) (:165,206,354)."""
    e1 = _embed(patchify(x["image"], image_patch), P, "image_to_patch_embedding")
    e2 = _embed(patchify(x["tactile1"], tactile_patch), P, "tactile_to_patch_embedding_1")
    e3 = _embed(patchify(x["tactile2"], tactile_patch), P, "tactile_to_patch_embedding_2")
    H, W = x["image"].shape[-2:]
    D = e1.shape[-1]
    pos = sinusoidal_embed((3 * H // image_patch, W // image_patch), D).to(e1.dtype)
    n1, n2 = e1.shape[1], e2.shape[1]
    e1 = e1 + pos[:n1]
    e2 = e2 + pos[n1:2 * n2]
    e3 = e3 + pos[2 * n1:]
    if masks is not None:
        e1, e2, e3 = apply_masks(e1, masks), apply_masks(e2, masks), apply_masks(e3, masks)
    t = torch.cat([e1, e2, e3], dim=1)
    nreg = 0
    if "register_tokens" in P:                                # models/VTT.py:305-312: register tokens lead the sequence
        nreg = P["register_tokens"].shape[1]
        t = torch.cat([P["register_tokens"].expand(t.shape[0], -1, -1), t], dim=1)
    pre = transformer(t, P, "transformer.", depth, heads, dim_head)
    xn = F.layer_norm(pre, (D,), P["norm.weight"], P["norm.bias"], 1e-6)
    return {"x_prenorm": pre, "x_norm_patchtokens": xn[:, nreg:], "x_norm_regtokens": xn[:, :nreg]}
