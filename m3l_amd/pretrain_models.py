"""Drop-in `VTT` / `VTMAE` (+ the `Transformer` they hold) with the reference's constructor kwargs, attribute names,
parameter names/shapes and forward signatures — reference: /root/reference/models/pretrain_models.py:59-786 and
vit-pytorch 1.6.4 `vit_pytorch.vit.Transformer` (third party).  Same construction ORDER as the reference, so
`torch.manual_seed(s)` followed by construction yields the reference's initial weights.

The nn.Modules here are parameter containers + orchestration; all arithmetic of the hot path runs in the HIP kernels of
m3l_amd/csrc through the C ABI (include/m3l_amd.h).  There is no eager/CPU fallback: inputs must be CUDA (ROCm) tensors.

Extensions over the reference signature (all optional, defaults keep reference behaviour):
  * `VTMAE.forward(..., mask_noise=[...])`: inject the uniform noise the reference would draw with torch.rand
    (one (B, n) tensor per modality in RNG order image, tactile1..k).  Needed for bit-exact mask parity.
  * `compute_dtype='fp32' | 'bf16'` (constructor kwarg / `set_compute_dtype`): bf16 = bf16 MFMA operands, fp32
    accumulation, fp32 master weights, fp32 residual stream.
"""
import math

import torch
import torch.optim as optim
from torch import nn

from . import _lib as L
from . import functional as Fn


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


class Rearrange(nn.Module):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (einops Rearrange of the reference, pretrain_models.py:768,775).
    Kept so that `encoder.image_to_patch_embedding[0]` exists and is callable; the MAE path itself patchifies
    inside the fused HIP gather kernel."""

    def __init__(self, p1, p2):
        super().__init__()
        self.p1, self.p2 = p1, p2

    def forward(self, x):
        b, c, H, W = x.shape
        h, w = H // self.p1, W // self.p2
        return x.reshape(b, c, h, self.p1, w, self.p2).permute(0, 2, 4, 3, 5, 1).reshape(b, h * w, self.p1 * self.p2 * c)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("m3l_amd attention kernels are built for dim_head = 64 (all reference configs)")
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout)) if project_out else nn.Identity()


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


class Transformer(nn.Module):
    """Parameter layout of vit_pytorch.vit.Transformer; forward runs the HIP transformer stack."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        if dropout != 0.:
            raise NotImplementedError("dropout > 0 is not implemented in the HIP path (the reference trains with dropout = 0)")
        self.dim, self.depth, self.heads, self.mlp_dim = dim, depth, heads, mlp_dim
        self.project_out = not (heads == 1 and dim_head == dim)
        self.compute_dtype = "fp32"
        self._sink = None            # (GradSync, bucket) when gradients go straight into a flat buffer (m3l_amd.parallel)
        self._tcache = None          # the parameter list of _tensors() (the Parameter objects never change; dropped by _apply)
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                                              FeedForward(dim, mlp_dim, dropout=dropout)]))

    def _tensors(self):
        """The kernels' tensor list (11 per layer + the final norm), cached: ~150 attribute look-ups per call otherwise.  The cache is
        checked against the modules' parameter dicts on every call (one dict look-up + identity test per entry), so a parameter OBJECT
        replaced after the first forward (`layer.to_qkv.weight = nn.Parameter(...)`, pruning, a swapped Linear's parameters) is picked up
        instead of the kernels reading the old tensor.  A swapped sub-MODULE (`attn.to_qkv = nn.Linear(...)`) is not seen by this check:
        set `transformer._tcache = None` (or call .to() / .float(), which drop it) after surgery of that kind."""
        if self._tcache is not None:
            for t, (d, k) in zip(self._tcache, self._tsrc):
                if d is not None and d.get(k) is not t:
                    self._tcache = None
                    break
        if self._tcache is None:
            t, src = [], []
            for attn, ff in self.layers:
                out = attn.to_out[0] if self.project_out else None
                mods = [(attn.norm, "weight"), (attn.norm, "bias"), (attn.to_qkv, "weight"), (out, "weight"), (out, "bias"),
                        (ff.net[0], "weight"), (ff.net[0], "bias"), (ff.net[1], "weight"), (ff.net[1], "bias"), (ff.net[4], "weight"),
                        (ff.net[4], "bias")]
                for m, k in mods:
                    t.append(getattr(m, k) if m is not None else None)
                    src.append((m._parameters if m is not None else None, k))
            t += [self.norm.weight, self.norm.bias]
            src += [(self.norm._parameters, "weight"), (self.norm._parameters, "bias")]
            self._tcache, self._tsrc = t, src
        return list(self._tcache)

    def _apply(self, fn, *args, **kwargs):
        self._tcache = None
        return super()._apply(fn, *args, **kwargs)

    def _cfg(self):
        return L.TfCfg(self.dim, self.depth, self.heads, self.mlp_dim, int(self.project_out), Fn.dtype_code(self.compute_dtype))

    def run(self, x):
        """-> (y in compute dtype, y in f32); both carry gradient."""
        return Fn.TransformerFn.apply(self._sink, self._cfg(), x, *self._tensors())

    def forward(self, x):
        return self.run(x)[1]


class EarlyCNN(nn.Module):
    """Parameter layout of the reference's conv stem (pretrain_models.py:37-56); the arithmetic runs as im2col + MFMA GEMM
    in the HIP library (m3l_earlycnn_fwd/bwd)."""

    def __init__(self, in_channels, encoder_dim, key='image'):
        super().__init__()
        self.key = key
        self.in_channels, self.encoder_dim = in_channels, encoder_dim
        self.conv1 = nn.Conv2d(in_channels, encoder_dim // 8, 4, stride=2, padding=1)
        self.conv2 = nn.Conv2d(encoder_dim // 8, encoder_dim // 4, 4, stride=2, padding=1)
        if key == 'image':
            self.conv3 = nn.Conv2d(encoder_dim // 4, encoder_dim // 2, 4, stride=2, padding=1)
        else:
            self.conv3 = nn.Conv2d(encoder_dim // 4, encoder_dim // 2, 3, stride=1, padding=1)
        self.conv4 = nn.Conv2d(encoder_dim // 2, encoder_dim, 1)

    def _tensors(self):
        return [self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, self.conv3.weight, self.conv3.bias,
                self.conv4.weight, self.conv4.bias]

    def run(self, xs, compute_dtype="fp32", sink=None):
        """xs: list of (B, C, H, W) inputs sharing this stem -> (len(xs) * B, h*w, D) tokens, input-major."""
        H, W = xs[0].shape[-2:]
        cfg = L.CnnCfg(self.in_channels, H, W, self.encoder_dim, int(self.key != 'image'), Fn.dtype_code(compute_dtype))
        return Fn.EarlyCnnFn.apply(sink, cfg, list(xs), *self._tensors())

    def forward(self, x):
        return self.run([x])


class VTT(nn.Module):
    """Encoder container of the reference (pretrain_models.py:717-786): two patch-embed Sequentials, learned
    pos_embedding (unused under sincos), Transformer.  Like the reference it has no forward."""

    def __init__(self, *, image_size, tactile_size, image_patch_size, tactile_patch_size, dim, depth, heads, mlp_dim,
                 image_channels=3, tactile_channels=3, dim_head=64, dropout=0., emb_dropout=0, num_tactiles=2, frame_stack=1):
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
        if image_patch_height != image_patch_width or tactile_patch_height != tactile_patch_width:
            raise NotImplementedError("m3l_amd supports square patches (all reference configs)")
        num_patches_image = (image_height // image_patch_height) * (image_width // image_patch_width)
        num_patches_tactile = (tactile_height // tactile_patch_height) * (tactile_width // tactile_patch_width) * num_tactiles
        num_patches = num_patches_image + num_patches_tactile
        image_patch_dim = image_channels * image_patch_height * image_patch_width
        tactile_patch_dim = tactile_channels * tactile_patch_height * tactile_patch_width
        self.image_to_patch_embedding = nn.Sequential(
            Rearrange(image_patch_height, image_patch_width), nn.LayerNorm(image_patch_dim),
            nn.Linear(image_patch_dim, dim), nn.LayerNorm(dim))
        self.tactile_to_patch_embedding = nn.Sequential(
            Rearrange(tactile_patch_height, tactile_patch_width), nn.LayerNorm(tactile_patch_dim),
            nn.Linear(tactile_patch_dim, dim), nn.LayerNorm(dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.to_latent = nn.Identity()


def _sincos_2d(channels_param, gh, gw, out_ch):
    """positional-encodings 6.0.1 `PositionalEncoding2D(channels_param)` on a (1, gh, gw, out_ch) tensor -> (1, gh*gw, out_ch)
    (reference call sites pretrain_models.py:120-140: ONE object parameterised by the encoder dim is reused for the
    decoder buffers, which therefore hold the first decoder_dim channels of the encoder-dim code)."""
    ch = int(math.ceil(channels_param / 4) * 2)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))

    def emb(n):
        s = torch.einsum("i,j->ij", torch.arange(n).float(), inv_freq)
        return torch.stack((s.sin(), s.cos()), dim=-1).flatten(-2, -1)
    e = torch.zeros(gh, gw, 2 * ch)
    e[:, :, :ch] = emb(gh).unsqueeze(1)
    e[:, :, ch:2 * ch] = emb(gw)
    return e[None, :, :, :out_ch].reshape(1, gh * gw, -1)


class VTMAE(nn.Module):
    def __init__(self, *, encoder, decoder_dim, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64,
                 num_tactiles=2, early_conv_masking=False, use_sincosmod_encodings=True, frame_stack=1, compute_dtype="fp32"):
        super().__init__()
        assert masking_ratio > 0 and masking_ratio < 1, 'masking ratio must be kept between 0 and 1'
        self.masking_ratio = masking_ratio
        self.num_tactiles = num_tactiles
        self.frame_stack = frame_stack
        self.encoder = encoder
        num_patches, encoder_dim = encoder.pos_embedding.shape[-2:]
        num_decoder_patches = num_patches - 1
        self.use_sincosmod_encodings = use_sincosmod_encodings
        self.early_conv_masking = early_conv_masking
        if self.early_conv_masking:
            self.early_conv_vision = EarlyCNN(self.encoder.image_channels, encoder_dim, key='image')
            self.early_conv_tactile = EarlyCNN(self.encoder.tactile_channels, encoder_dim, key='tactile')

        self.image_to_patch = encoder.image_to_patch_embedding[0]
        self.image_patch_to_emb = nn.Sequential(*encoder.image_to_patch_embedding[1:])
        pixel_values_per_patch = encoder.image_to_patch_embedding[2].weight.shape[-1]
        self.tactile_to_patch = encoder.tactile_to_patch_embedding[0]
        self.tactile_patch_to_emb = nn.Sequential(*encoder.tactile_to_patch_embedding[1:])
        tactile_values_per_patch = encoder.tactile_to_patch_embedding[2].weight.shape[-1]
        self.encoder_dim = encoder_dim

        self.decoder_dim = decoder_dim
        self.enc_to_dec = nn.Linear(encoder_dim, decoder_dim) if encoder_dim != decoder_dim else nn.Identity()
        self.mask_token = nn.Parameter(torch.randn(decoder_dim))
        self.decoder = Transformer(dim=decoder_dim, depth=decoder_depth, heads=decoder_heads, dim_head=decoder_dim_head,
                                   mlp_dim=decoder_dim * 4)
        self.decoder_pos_emb = nn.Embedding(num_decoder_patches, decoder_dim)
        self.to_pixels = nn.Linear(decoder_dim, pixel_values_per_patch)
        self.to_tactiles = nn.Linear(decoder_dim, tactile_values_per_patch)

        gh, gw = encoder.image_height // encoder.image_patch_height, encoder.image_width // encoder.image_patch_width
        th, tw = encoder.tactile_height // encoder.tactile_patch_height, encoder.tactile_width // encoder.tactile_patch_width
        if decoder_dim > 2 * int(math.ceil(encoder_dim / 4) * 2):
            raise ValueError("decoder_dim larger than the encoder-dim sincos code (the reference fails with a shape error here)")
        self.register_buffer('image_enc_pos_embedding', _sincos_2d(encoder_dim, gh, gw, encoder_dim))
        self.register_buffer('tactile_enc_pos_embedding', _sincos_2d(encoder_dim, th, tw, encoder_dim).repeat(1, num_tactiles, 1))
        self.register_buffer('image_dec_pos_embedding', _sincos_2d(encoder_dim, gh, gw, decoder_dim))
        self.register_buffer('tactile_dec_pos_embedding', _sincos_2d(encoder_dim, th, tw, decoder_dim).repeat(1, num_tactiles, 1))

        self.encoder_modality_embedding = nn.Embedding((1 + self.num_tactiles), encoder_dim)
        self.decoder_modality_embedding = nn.Embedding((1 + self.num_tactiles), decoder_dim)
        # use_sincosmod_encodings=False (:218-219,280-287): the kernels still add "modality + position" per token; the modality term is
        # this zero table and the position term a slice of the learned table (not part of the state dict)
        self.register_buffer('_zero_mod_enc', torch.zeros(1 + self.num_tactiles, encoder_dim), persistent=False)
        self.register_buffer('_zero_mod_dec', torch.zeros(1 + self.num_tactiles, decoder_dim), persistent=False)
        self.set_compute_dtype(compute_dtype)
        self._sinks = {}            # bucket name -> (GradSync, bucket id), filled by m3l_amd.parallel.GradSync
        self.last_mask = None       # (masked_indices, unmasked_indices) of the latest forward, int64 (B, *)

    # ------------------------------------------------------------------------------------------------------------
    def set_compute_dtype(self, compute_dtype):
        Fn.dtype_code(compute_dtype)
        self.compute_dtype = compute_dtype
        self.encoder.transformer.compute_dtype = compute_dtype
        self.decoder.compute_dtype = compute_dtype
        return self

    def _enc_positions(self, geom):
        """(modality table, image positions, tactile positions) the embed kernels add: the sincos buffers, or — learned mode — rows
        1.. of encoder.pos_embedding in the order of the tokens PRESENT in this call (pretrain_models.py:218-219)."""
        if self.use_sincosmod_encodings:
            return self.encoder_modality_embedding.weight, self.image_enc_pos_embedding[0], self.tactile_enc_pos_embedding[0]
        c = Fn.mask_counts(geom, self.masking_ratio)
        k = geom.num_tactiles if geom.use_tactile else 0
        n_img = c["n_img"]
        pe = self.encoder.pos_embedding[0]
        return self._zero_mod_enc, pe[1:1 + n_img], pe[1 + n_img:1 + n_img + k * c["n_tac"]]

    def _embed_tensors(self, geom):
        i, t = self.encoder.image_to_patch_embedding, self.encoder.tactile_to_patch_embedding
        return [i[1].weight, i[1].bias, i[2].weight, i[2].bias, i[3].weight, i[3].bias,
                t[1].weight, t[1].bias, t[2].weight, t[2].bias, t[3].weight, t[3].bias, *self._enc_positions(geom)]

    def _glue_tensors(self, geom):
        e2d = isinstance(self.enc_to_dec, nn.Linear)
        if self.use_sincosmod_encodings:
            pos = [self.decoder_modality_embedding.weight, self.image_dec_pos_embedding[0], self.tactile_dec_pos_embedding[0]]
        else:       # decoder_pos_emb(unmasked_indices) / (masked_indices): position j of the present tokens gets row j (:280-287)
            c = Fn.mask_counts(geom, self.masking_ratio)
            k = geom.num_tactiles if geom.use_tactile else 0
            w = self.decoder_pos_emb.weight
            pos = [self._zero_mod_dec, w[:c["n_img"]], w[c["n_img"]:c["n_img"] + k * c["n_tac"]]]
        return [self.enc_to_dec.weight if e2d else None, self.enc_to_dec.bias if e2d else None, self.mask_token, *pos]

    def _head_tensors(self):
        return [self.to_pixels.weight, self.to_pixels.bias, self.to_tactiles.weight, self.to_tactiles.bias]

    def _inputs(self, x, use_vision, use_tactile):
        if 'image' not in x.keys():
            use_vision = False
        use_tactile = bool(use_tactile) and self.num_tactiles > 0
        image = x['image'] if use_vision else None
        tactiles = [x['tactile' + str(i)] for i in range(1, self.num_tactiles + 1)] if use_tactile else []
        ref = image if image is not None else (tactiles[0] if tactiles else None)
        if ref is None:
            raise ValueError("neither image nor tactile inputs in use")
        geom = Fn.make_geom(self.encoder, self.num_tactiles, use_vision, use_tactile)
        return image, tactiles, geom, ref

    def _stem_tokens(self, geom, image, tactiles):
        """early_conv_masking=True: EarlyCNN stems over the whole frames, then + modality + sincos -> (B, N, D)."""
        part = (self._sinks["embed"][0], None) if "embed" in self._sinks else None     # three Functions share the embed bucket
        img_tok = self.early_conv_vision.run([image], self.compute_dtype, part) if image is not None else None
        tac_tok = self.early_conv_tactile.run(tactiles, self.compute_dtype, part) if tactiles else None
        return Fn.TokensAssembleFn.apply(part, geom, self.encoder_dim, img_tok, tac_tok, *self._enc_positions(geom))

    def _tokens(self, geom, image, tactiles, idx, cnt_img, L_tok):
        dt = Fn.dtype_code(self.compute_dtype)
        if self.early_conv_masking:
            tokens = self._stem_tokens(geom, image, tactiles)
            return tokens if idx is None else Fn.GatherTokensFn.apply(tokens, idx)
        return Fn.EmbedFn.apply(self._sinks.get("embed"), geom, self.encoder_dim, dt, idx, cnt_img, L_tok, image, tactiles,
                                *self._embed_tensors(geom))

    # ------------------------------------------------------------------------------------------------------------
    def _front_tensors(self, geom, has_img, has_tac):
        """(tensors, used) of the encoder's front end in the order the C step expects: patch embed (15) or EarlyCNN stems (8 + 8 + 3)."""
        learned = not self.use_sincosmod_encodings
        if self.early_conv_masking:
            t = self.early_conv_vision._tensors() + self.early_conv_tactile._tensors() + list(self._enc_positions(geom))
            return t, [has_img] * 8 + [has_tac] * 8 + [True, learned and has_img, learned and has_tac]
        return self._embed_tensors(geom), [has_img] * 6 + [has_tac] * 6 + [True, learned and has_img, learned and has_tac]

    def _step_fused(self, image, tactiles, geom, mask_noise, c, B, sync):
        """The whole step inside the library (csrc/mae_step.hip: m3l_mae_step_fwd / _bwd): one autograd node, two host calls."""
        plan = Fn.StepPlan()
        enc_tf, dec_tf = self.encoder.transformer, self.decoder
        learned = not self.use_sincosmod_encodings
        plan.cfg = L.MaeCfg(geom, enc_tf._cfg(), dec_tf._cfg(), float(self.masking_ratio), int(self.early_conv_masking), int(learned))
        has_img, has_tac = image is not None, len(tactiles) > 0
        emb, used_emb = self._front_tensors(geom, has_img, has_tac)
        glue = self._glue_tensors(geom)
        plan.tensors = emb + enc_tf._tensors() + glue + dec_tf._tensors() + self._head_tensors()
        # which tensors receive a gradient in this call (absent modalities and the fixed sincos tables do not; learned position tables do)
        plan.used = (used_emb + [True] * (11 * enc_tf.depth + 2) + [True, True, True, True, learned and has_img, learned and has_tac]
                     + [True] * (11 * dec_tf.depth + 2) + [has_img] * 2 + [has_tac] * 2)
        for i, t in enumerate(plan.tensors):
            if t is None:
                plan.used[i] = False          # (a frozen parameter still gets a scratch gradient: the kernels write every slot they own)
        plan.image = Fn._f32c(image)
        plan.tactiles = [Fn._f32c(t) for t in tactiles]
        plan.noises = [Fn._f32c(n) for n in mask_noise]
        plan.sync, plan.B, plan.nmask, plan.nvis = sync, B, c["num_masked"], c["num_unmasked"]
        anchor = next((t for t in plan.tensors if t is not None and t.requires_grad and t.is_leaf), None)
        if sync is not None and torch.is_grad_enabled() and anchor is not None:
            # anchor: the kernels write every gradient in place, autograd only has to call backward — and to carry the gradients of the
            # learned position slices (views of encoder.pos_embedding / decoder_pos_emb.weight) back to their parameters
            plan.extra = [i for i, t in enumerate(plan.tensors) if t is not None and plan.used[i] and t.requires_grad and not t.is_leaf]
            ins = (anchor,) + tuple(plan.tensors[i] for i in plan.extra)
        else:
            plan.extra = None
            ins = tuple(t for t in plan.tensors if t is not None)
        loss = Fn.MaeStepFn.apply(plan, *ins)
        self.last_mask = (plan.masked, plan.unmasked)
        return loss

    def _step(self, x, use_vision, use_tactile, mask_noise, dump, counts=None):
        image, tactiles, geom, ref = self._inputs(x, use_vision, use_tactile)
        Fn._require_cuda(ref, "MAE input")
        B, dev = ref.shape[0], ref.device
        dt = Fn.dtype_code(self.compute_dtype)
        c = Fn.mask_counts(geom, self.masking_ratio)
        sizes = ([c["n_img"]] if image is not None else []) + [c["n_tac"]] * len(tactiles)
        if mask_noise is None:
            mask_noise = [torch.rand(B, n, device=dev) for n in sizes]          # reference RNG order (:229,:237)
        assert [tuple(n.shape) for n in mask_noise] == [(B, n) for n in sizes], "mask_noise: one (B, n) tensor per modality"
        sync = self._sinks["heads"][0] if "heads" in self._sinks else None
        if (Fn.FUSED_STEP and dump is None and counts is None and Fn.BWD_CHUNK_LAYERS is None and (sync is None or not sync._comm or sync._direct)
                and (not self.early_conv_masking or (self.encoder.image_patch_height == 8 and self.encoder.tactile_patch_height == 4))):
            return self._step_fused(image, tactiles, geom, [n.to(dev) for n in mask_noise], c, B, sync)
        masked, unmasked, c = Fn.mask_sample(geom, self.masking_ratio, [n.to(dev) for n in mask_noise], counts)
        self.last_mask = (masked, unmasked)
        nvis_img = c["n_img"] - c["nm_img"]

        tokens = self._tokens(geom, image, tactiles, unmasked, nvis_img, c["num_unmasked"])
        enc_t, enc32 = self.encoder.transformer.run(tokens)
        dec_in = Fn.UnshuffleFn.apply(self._sinks.get("glue"), geom, self.encoder_dim, self.decoder_dim, dt, unmasked, masked,
                                      enc_t, enc32, *self._glue_tensors(geom))
        dec_t, _ = self.decoder.run(dec_in)
        if self.early_conv_masking:
            # pretrain_models.py:311-322: predict ALL patches, loss over all of them (same kernel, identity index list)
            N = c["num_masked"] + c["num_unmasked"]
            rows = torch.arange(N, device=dev, dtype=torch.int64).expand(B, N).contiguous()
            loss = Fn.HeadsLossFn.apply(self._sinks.get("heads"), geom, self.decoder_dim, dt, rows, c["n_img"], image, tactiles,
                                        dump, dec_t, *self._head_tensors())
        else:
            loss = Fn.HeadsLossFn.apply(self._sinks.get("heads"), geom, self.decoder_dim, dt, masked, c["nm_img"], image, tactiles,
                                        dump, dec_t, *self._head_tensors())
        if dump is not None:
            dump.update(masked_indices=masked, unmasked_indices=unmasked, encoder_in=tokens.detach(), encoder_out=enc32.detach(),
                        decoder_in=dec_in.detach(), decoder_out=dec_t.detach(), counts=c)
        return loss

    def forward(self, x, use_vision=True, use_tactile=True, mask_noise=None, dump=None):
        """Reconstruction loss (0-dim tensor with grad), pretrain_models.py:146-342."""
        return self._step(x, use_vision, use_tactile, mask_noise, dump)

    @torch.no_grad()
    def reconstruct(self, x, mask_ratio=None, use_vision=True, use_tactile=True, mask_noise=None):
        """Visual-logging companion of forward (pretrain_models.py:344-586): masks with its OWN count rule
        (int(r * n_img) image patches, int(r * n_tac_total / k) per sensor), runs the auto-encoder and returns
        {'image_rec', 'image_masked', 'recon_loss_image', 'tactile_rec', 'tactile_masked', 'recon_loss_tactile'}:
        frames with the masked patches replaced by the prediction / by 0.5 (image) or inf (tactile).  The arithmetic is
        the same HIP step as forward; torch only re-tiles patches into frames."""
        r = self.masking_ratio if mask_ratio is None else mask_ratio
        image, tactiles, geom, ref = self._inputs(x, use_vision, use_tactile)
        c0 = Fn.mask_counts(geom, self.masking_ratio)
        k = len(tactiles)
        nm_img = int(r * c0["n_img"]) if image is not None else 0
        nm_tac = int(r * (k * c0["n_tac"]) / k) if k else 0
        keep, self.masking_ratio = self.masking_ratio, r
        try:
            dump = {}
            self._step(x, use_vision, use_tactile, mask_noise, dump, counts=(nm_img, nm_tac))
        finally:
            self.masking_ratio = keep
        B = ref.shape[0]
        br = torch.arange(B, device=ref.device)[:, None]
        masked = dump["masked_indices"]
        n_img = c0["n_img"] if image is not None else 0
        mi_img, mi_tac = masked[:, :nm_img], masked[:, nm_img:] - n_img
        enc = self.encoder
        out = {}

        def frames(p, n, gh, gw, ph, pw):     # 'b (n h w) (p1 p2 c) -> b (n c) (h p1) (w p2)'
            b, _, pd = p.shape
            ch = pd // (ph * pw)
            return p.reshape(b, n, gh, gw, ph, pw, ch).permute(0, 1, 6, 2, 4, 3, 5).reshape(b, n * ch, gh * ph, gw * pw)
        if image is not None:
            gh, gw = enc.image_height // enc.image_patch_height, enc.image_width // enc.image_patch_width
            patches = self.image_to_patch(image.float())
            vis = patches.clone()
            vis[br, mi_img] = 0.5
            if self.early_conv_masking:
                rec = dump["pred_pixel"]
            else:
                rec = patches.clone()
                rec[br, mi_img] = dump["pred_pixel"]
            out["image_rec"] = frames(rec, 1, gh, gw, enc.image_patch_height, enc.image_patch_width)
            out["image_masked"] = frames(vis, 1, gh, gw, enc.image_patch_height, enc.image_patch_width)
            out["recon_loss_image"] = dump["loss_parts"][0]
        if k:
            gh, gw = enc.tactile_height // enc.tactile_patch_height, enc.tactile_width // enc.tactile_patch_width
            patches = torch.cat([self.tactile_to_patch(t.float()) for t in tactiles], dim=1)
            vis = patches.clone()
            vis[br, mi_tac] = float("inf")
            if self.early_conv_masking:
                rec = dump["pred_tactile"]
            else:
                rec = patches.clone()
                rec[br, mi_tac] = dump["pred_tactile"]
            out["tactile_rec"] = frames(rec, k, gh, gw, enc.tactile_patch_height, enc.tactile_patch_width)
            out["tactile_masked"] = frames(vis, k, gh, gw, enc.tactile_patch_height, enc.tactile_patch_width)
            out["recon_loss_tactile"] = dump["loss_parts"][1] / 10.0
        return out

    def get_embeddings(self, x, eval=True, use_vision=True, use_tactile=True):
        """Encoder over ALL tokens, no masking (pretrain_models.py:588-668) -> (B, N, D) f32 with grad."""
        if eval:
            self.eval()
        else:
            self.train()
        image, tactiles, geom, _ = self._inputs(x, use_vision, use_tactile)
        c = Fn.mask_counts(geom, self.masking_ratio)
        N = c["num_masked"] + c["num_unmasked"]
        tokens = self._tokens(geom, image, tactiles, None, c["n_img"], N)
        return self.encoder.transformer(tokens)

    def initialize_training(self, train_args):
        """pretrain_models.py:673-677.  train_args['fused_optimizer'] (not a reference key, default False): the AdamW update and the
        clip_grad_norm_(0.5) of train_iterations as m3l_amd.parallel.FlatAdamW over one flat parameter / gradient buffer (one
        reduction + one update launch per step) instead of torch.optim.AdamW + torch.nn.utils.clip_grad_norm_."""
        self._fused_opt = bool(train_args.get('fused_optimizer', False))
        if self._fused_opt:
            from .parallel import FlatAdamW, GradSync
            self._opt_sync = GradSync(self)
            self.optimizer = FlatAdamW(self._opt_sync, lr=train_args['lr'], max_grad_norm=0.5)
        else:
            self.optimizer = optim.AdamW(self.parameters(), lr=train_args['lr'])
        self.batch_size = train_args['batch_size']

    def train_iterations(self, iterations, replay_buffer, no_tactile=False):
        """Stand-alone MAE training from a replay buffer (pretrain_models.py:679-715): `iterations` steps of
        sample batch -> stack frames -> vt_load -> loss -> backward -> clip_grad_norm_(0.5) -> AdamW step; leaves the module in
        eval mode.  replay_buffer: list of {'image': (fs, H, W, 3), 'tactile': (fs, 6, h, w)} dicts."""
        import random
        import numpy as np
        from .pretrain_utils import vt_load
        if len(replay_buffer) < self.batch_size:
            print("Not enough samples in replay buffer")
            return
        self.train()
        dev = self.mask_token.device
        for _ in range(iterations):
            batch = random.choices(replay_buffer, k=self.batch_size)
            obs = {}
            for key in (['image'] if no_tactile else ['image', 'tactile']):
                obs[key] = np.stack([b[key] for b in batch])
            if 'image' in obs:                                  # (B, fs, H, W, 3) -> (B, H, W, 3*fs)
                im = obs['image'].transpose((0, 2, 3, 1, 4))
                obs['image'] = im.reshape((im.shape[0], im.shape[1], im.shape[2], -1))
            if 'tactile' in obs:                                # (B, fs, 6, h, w) -> (B, 6*fs, h, w)
                t = obs['tactile']
                obs['tactile'] = t.reshape((t.shape[0], -1, t.shape[3], t.shape[4]))
            xb = vt_load(obs, frame_stack=self.frame_stack, device=dev)
            self.optimizer.zero_grad()
            r_loss = self(xb, use_tactile=not no_tactile)
            r_loss.backward()
            if not getattr(self, "_fused_opt", False):
                torch.nn.utils.clip_grad_norm_(self.parameters(), 0.5)
            self.optimizer.step()                               # (fused: the clip is part of the update)
        self.eval()
