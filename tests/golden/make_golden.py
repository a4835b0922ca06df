#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference is mounted).

Executes the REFERENCE's own `VTT` + `VTMAE` (`/root/reference/models/pretrain_models.py:59-786`) and
`vt_load` (`/root/reference/utils/pretrain_utils.py:7-57`) on CPU fp32 and dumps small `.npz` fixtures
(weights by state-dict key, inputs, injected mask noise, indices, intermediates, loss, all grads) into
tests/golden/. The reference source is imported from where it lies; nothing of it is copied.

How the import is made possible (ordinary ModuleNotFoundError otherwise):
  * inert stubs for NON-arithmetic imports (gymnasium, stable_baselines3.*, cv2) that VTT/VTMAE never
    call on this path;
  * restated third-party arithmetic (vit-pytorch 1.6.4 Transformer, positional-encodings 6.0.1
    PositionalEncoding2D) from tests/golden/_thirdparty_restated.py  -> that part is "parity unpinned".

Mask noise: the reference draws `torch.rand(batch, n)` per modality (pretrain_models.py:229,237); we
replace `torch.rand` for the duration of the forward by a queue of pre-generated noise tensors so the
same noise can be handed to the oracle and to the HIP path.

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    sys.path.insert(0, HERE)
    import _thirdparty_restated as tp

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:  # inert base classes for the SB3 policy/extractor classes defined in the same file
        def __init__(self, *a, **k):
            pass

    class _BFE(torch.nn.Module):  # stable_baselines3 BaseFeaturesExtractor: an nn.Module that remembers features_dim
        def __init__(self, observation_space=None, features_dim=0):
            super().__init__()
            self._features_dim = features_dim

    vp = mod("vit_pytorch")
    vp.vit = mod("vit_pytorch.vit", pair=tp.pair, Transformer=tp.Transformer)
    pe = mod("positional_encodings")
    pe.torch_encodings = mod("positional_encodings.torch_encodings", PositionalEncoding2D=tp.PositionalEncoding2D)
    gym = mod("gymnasium", Space=_Dummy)
    gym.spaces = mod("gymnasium.spaces", Space=_Dummy, Dict=_Dummy, Box=_Dummy)
    sb3 = mod("stable_baselines3")
    sb3.common = mod("stable_baselines3.common")
    mod("stable_baselines3.common.torch_layers", BaseFeaturesExtractor=_BFE, FlattenExtractor=_Dummy)
    mod("stable_baselines3.common.type_aliases", Schedule=object)
    mod("stable_baselines3.common.policies", ActorCriticPolicy=torch.nn.Module)
    mod("stable_baselines3.common.logger", Video=_Dummy)
    mod("cv2")
    mod("omegaconf", DictConfig=_Dummy, OmegaConf=_Dummy)   # only reached through tactile_ssl/utils/logging.py:8
    lt = mod("lightning")                                    # tactile_ssl/utils/logging.py:16 (rank-zero log decorator)
    lt.fabric = mod("lightning.fabric")
    lt.fabric.utilities = mod("lightning.fabric.utilities", rank_zero_only=lambda f: f)


def _load_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_pretrain_models", os.path.join(REF, "models", "pretrain_models.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _RandQueue:
    """Context manager: torch.rand(...) pops pre-generated noise (shape-checked)."""

    def __init__(self, noises):
        self.noises = list(noises)
        self.real = torch.rand

    def __enter__(self):
        def fake(*shape, device=None, **kw):
            n = self.noises.pop(0)
            assert tuple(n.shape) == tuple(shape), (n.shape, shape)
            return n.clone()
        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self.real
        assert not self.noises, "unused noise tensors"


def _noise(gen, B, n, tie_row=None):
    x = torch.rand(B, n, generator=gen)
    if tie_row is not None and n >= 6:
        # crafted ties: float32-equal keys in one row (stable-ascending tie-break contract)
        x[tie_row, 1] = x[tie_row, 4]
        x[tie_row, 2] = x[tie_row, 4]
        x[tie_row, n - 1] = x[tie_row, 0]
    return x


def run_case(ref, name, *, image_hw, tactile_hw, ip, tp_, dim, depth, heads, mlp, C, num_tactiles,
             dec_dim, dec_depth, dec_heads, ratio, B, seed, with_embeddings=False, early_conv=False, sincos=True):
    torch.manual_seed(seed)
    enc = ref.VTT(image_size=image_hw, tactile_size=tactile_hw, image_patch_size=ip, tactile_patch_size=tp_,
                  dim=dim, depth=depth, heads=heads, mlp_dim=mlp, image_channels=C, tactile_channels=C,
                  num_tactiles=num_tactiles)
    mae = ref.VTMAE(encoder=enc, decoder_dim=dec_dim, masking_ratio=ratio, decoder_depth=dec_depth,
                    decoder_heads=dec_heads, num_tactiles=num_tactiles, early_conv_masking=early_conv,
                    use_sincosmod_encodings=sincos)
    # make LN affine / biases non-trivial so that every parameter is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, p in mae.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    mae.train()

    g = torch.Generator().manual_seed(seed + 2)
    x = {"image": torch.rand(B, C, image_hw, image_hw, generator=g)}
    for i in range(num_tactiles):
        x[f"tactile{i + 1}"] = torch.rand(B, C, tactile_hw, tactile_hw, generator=g)
    n_img = (image_hw // ip) ** 2
    n_tac = (tactile_hw // tp_) ** 2
    noises = [_noise(g, B, n_img, tie_row=0)] + [_noise(g, B, n_tac, tie_row=(B - 1 if i == 0 else None)) for i in range(num_tactiles)]

    cap = {}

    def hook(key):
        def f(mod, inp, out):
            cap[key + "_in"] = inp[0].detach().clone()
            cap[key + "_out"] = out.detach().clone()
        return f

    if early_conv:
        hs_extra = [mae.early_conv_vision.register_forward_hook(hook("early_conv_vision")),
                    mae.early_conv_tactile.register_forward_hook(hook("early_conv_tactile"))]
    else:
        hs_extra = []
    hs = hs_extra + [mae.encoder.transformer.register_forward_hook(hook("encoder")),
          mae.decoder.register_forward_hook(hook("decoder")),
          mae.to_pixels.register_forward_hook(hook("to_pixels")),
          mae.to_tactiles.register_forward_hook(hook("to_tactiles")),
          mae.image_patch_to_emb.register_forward_hook(hook("image_embed")),
          mae.tactile_patch_to_emb.register_forward_hook(hook("tactile_embed"))]
    with _RandQueue(noises):
        loss = mae({k: v.clone() for k, v in x.items()})
    for h in hs:
        h.remove()
    loss.backward()

    out = {}
    for k, v in mae.state_dict().items():
        out["param/" + k] = v.detach().numpy()
    for k, p in mae.named_parameters():
        if p.grad is not None:
            out["grad/" + k] = p.grad.detach().numpy()
    for k, v in x.items():
        out["input/" + k] = v.numpy()
    for i, nz in enumerate(noises):
        out[f"noise/{i}"] = nz.numpy()
        out[f"argsort/{i}"] = nz.argsort(dim=-1).numpy()      # the reference's own expression (:229,:237)
    for k, v in cap.items():
        out["cap/" + k] = v.numpy()
    out["loss"] = loss.detach().numpy()
    out["meta"] = np.array([image_hw, tactile_hw, ip, tp_, dim, depth, heads, mlp, C, num_tactiles,
                            dec_dim, dec_depth, dec_heads, B], dtype=np.int64)
    out["ratio"] = np.array(ratio, dtype=np.float64)
    out["unused_params"] = np.array([k for k, p in mae.named_parameters() if p.grad is None])
    out["early_conv"] = np.array(int(early_conv))
    out["sincos"] = np.array(int(sincos))

    if with_embeddings:
        with torch.no_grad():
            emb = mae.get_embeddings({k: v.clone() for k, v in x.items()}, eval=True)
        out["embeddings"] = emb.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(loss):.8f}  keys={len(out)}")


def run_reconstruct(ref, name, *, early_conv, ratio, mask_ratio, seed, use_tactile=True):
    """`VTMAE.reconstruct` (pretrain_models.py:344-586): its own mask-count rule and the logging frames."""
    torch.manual_seed(seed)
    B, C, nt = 2, 3, 2
    enc = ref.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2,
                  mlp_dim=128, image_channels=C, tactile_channels=C, num_tactiles=nt)
    mae = ref.VTMAE(encoder=enc, decoder_dim=64, masking_ratio=ratio, decoder_depth=1, decoder_heads=2, num_tactiles=nt,
                    early_conv_masking=early_conv, use_sincosmod_encodings=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, p in mae.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    mae.eval()
    x = {"image": torch.rand(B, C, 32, 32, generator=g)}
    for i in range(nt):
        x[f"tactile{i + 1}"] = torch.rand(B, C, 16, 16, generator=g)
    noises = [_noise(g, B, 16)] + ([_noise(g, B, 16) for _ in range(nt)] if use_tactile else [])
    with torch.no_grad(), _RandQueue(noises):
        r = mae.reconstruct({k: v.clone() for k, v in x.items()}, mask_ratio=mask_ratio, use_tactile=use_tactile)
    out = {"param/" + k: v.detach().numpy() for k, v in mae.state_dict().items()}
    out.update({"input/" + k: v.numpy() for k, v in x.items()})
    for i, nz in enumerate(noises):
        out[f"noise/{i}"] = nz.numpy()
    out.update({"out/" + k: v.detach().numpy() for k, v in r.items()})
    out["meta"] = np.array([32, 16, 8, 4, 64, 1, 2, 128, C, nt, 64, 1, 2, B], dtype=np.int64)
    out["ratio"] = np.array(ratio, dtype=np.float64)
    out["mask_ratio"] = np.array(-1.0 if mask_ratio is None else mask_ratio, dtype=np.float64)
    out["early_conv"] = np.array(int(early_conv))
    out["use_tactile"] = np.array(int(use_tactile))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: keys={sorted(r.keys())} losses={[float(v) for k, v in r.items() if k.startswith('recon_loss')]}")


def run_ppo_like(ref):
    """The MAE update of `PPO_MAE.train` (models/ppo_mae.py:232-266) with a separate optimizer: frame-stacked rollout observations
    -> permute / reshape -> vt_load -> mae(x) -> backward -> Adam(lr).step(), two mini-batches.  Records the losses and the weights
    after each step, so a replacement is pinned through the optimizer as well."""
    import utils.pretrain_utils as pu
    torch.manual_seed(61)
    fs, B, nb = 2, 3, 2
    enc = ref.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128,
                  image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = ref.VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2,
                    early_conv_masking=False, use_sincosmod_encodings=True, frame_stack=fs)
    mae.train()
    opt = torch.optim.Adam(mae.parameters(), lr=1e-3)                       # ppo_mae.py:182-183 (lr raised so two steps are visible)
    g = torch.Generator().manual_seed(62)
    obs = {"image": torch.rand(B * nb, fs, 32, 32, 3, generator=g), "tactile": torch.rand(B * nb, fs, 6, 16, 16, generator=g) * 2 - 1}
    out = {"param0/" + k: v.detach().clone().numpy() for k, v in mae.state_dict().items()}
    out["obs/image"], out["obs/tactile"] = obs["image"].numpy(), obs["tactile"].numpy()
    o = {k: v.clone() for k, v in obs.items()}
    o["image"] = o["image"].permute(0, 2, 3, 1, 4)
    o["image"] = o["image"].reshape((o["image"].shape[0], o["image"].shape[1], o["image"].shape[2], -1))
    o["tactile"] = o["tactile"].reshape((o["tactile"].shape[0], -1, o["tactile"].shape[3], o["tactile"].shape[4]))
    losses = []
    for i in range(nb):
        opt.zero_grad()
        x = pu.vt_load({k: v[i * B:(i + 1) * B].clone() for k, v in o.items()}, frame_stack=fs)
        noises = [_noise(g, B, 16) for _ in range(3)]
        for j, nz in enumerate(noises):
            out[f"noise/{i}/{j}"] = nz.numpy()
        with _RandQueue(noises):
            loss = mae(x)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        for k in ("to_pixels.weight", "encoder.transformer.layers.0.0.to_qkv.weight", "mask_token", "encoder.image_to_patch_embedding.2.bias"):
            out[f"step{i}/" + k] = dict(mae.named_parameters())[k].detach().clone().numpy()
    out["losses"] = np.array(losses, dtype=np.float64)
    out["meta"] = np.array([32, 16, 8, 4, 64, 1, 2, 128, 3 * fs, 2, 64, 1, 2, B], dtype=np.int64)
    out["frame_stack"] = np.array(fs)
    np.savez_compressed(os.path.join(HERE, "ppo_like_step.npz"), **out)
    print("ppo_like_step: losses", losses)


def run_train_iterations(ref):
    """`VTMAE.initialize_training` + `VTMAE.train_iterations` themselves (models/pretrain_models.py:673-715): AdamW(lr) +
    clip_grad_norm_(0.5) over batches that `random.choices` draws from a replay buffer, one iteration per call, twice.  Records the
    buffer, the drawn indices (the same `random` seed gives them to a replacement), the mask noise and the weights after each iteration,
    so the second optimizer form of the path is pinned to the reference's own code."""
    import random
    torch.manual_seed(71)
    fs, bs, nbuf = 2, 4, 6
    enc = ref.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128,
                  image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = ref.VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2,
                    early_conv_masking=False, use_sincosmod_encodings=True, frame_stack=fs)
    mae.initialize_training({"lr": 1e-3, "batch_size": bs})
    rng = np.random.default_rng(72)
    buf = [{"image": rng.random((fs, 32, 32, 3), dtype=np.float32) * 40.0,       # x 40: the gradient norm exceeds 0.5, the clip bites
            "tactile": (rng.random((fs, 6, 16, 16), dtype=np.float32) * 2 - 1)} for _ in range(nbuf)]
    out = {"param0/" + k: v.detach().clone().numpy() for k, v in mae.state_dict().items()}
    for i, b in enumerate(buf):
        out[f"buf/{i}/image"], out[f"buf/{i}/tactile"] = b["image"], b["tactile"]
    g = torch.Generator().manual_seed(73)
    random.seed(74)
    st = random.getstate()
    out["choices"] = np.array([random.choices(range(nbuf), k=bs) for _ in range(2)], dtype=np.int64)
    random.setstate(st)
    keys = ("to_pixels.weight", "to_tactiles.bias", "encoder.transformer.layers.0.0.to_qkv.weight", "mask_token",
            "encoder.image_to_patch_embedding.2.bias", "decoder.layers.0.1.net.1.weight", "encoder.transformer.norm.weight")
    for it in range(2):
        noises = [_noise(g, bs, 16) for _ in range(3)]
        for j, nz in enumerate(noises):
            out[f"noise/{it}/{j}"] = nz.numpy()
        with _RandQueue(noises):
            mae.train_iterations(1, buf)
        for k in keys:
            out[f"iter{it}/" + k] = dict(mae.named_parameters())[k].detach().clone().numpy()
    out["meta"] = np.array([32, 16, 8, 4, 64, 1, 2, 128, 3 * fs, 2, 64, 1, 2, bs], dtype=np.int64)
    out["frame_stack"] = np.array(fs)
    np.savez_compressed(os.path.join(HERE, "train_iterations.npz"), **out)
    print("train_iterations: |to_pixels.weight| after", [float(np.abs(out[f"iter{it}/to_pixels.weight"]).sum()) for it in range(2)])


def run_extractor(ref):
    """`MAEExtractor.forward` (models/pretrain_models.py:788-841), the policy-side consumer: frame-stacked observations -> vt_load ->
    get_embeddings(eval=False) -> 1-layer Transformer -> mean over tokens.  The SB3 base class is an inert nn.Module stub."""
    torch.manual_seed(71)
    fs, B, D = 2, 3, 64
    enc = ref.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=D, depth=2, heads=2, mlp_dim=128,
                  image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = ref.VTMAE(encoder=enc, decoder_dim=D, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2,
                    early_conv_masking=False, use_sincosmod_encodings=True, frame_stack=fs)
    g = torch.Generator().manual_seed(72)
    out = {}
    for vision_only in (False, True):
        ext = ref.MAEExtractor(None, mae, D, vision_only, fs)
        with torch.no_grad():
            for n_, p in ext.vit_layer.named_parameters():       # (the MAE is shared by both extractors: leave it alone)
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        obs = {"image": torch.rand(B, fs, 32, 32, 3, generator=g), "tactile": torch.rand(B, fs, 6, 16, 16, generator=g) * 2 - 1}
        tag = "vision_only" if vision_only else "vt"
        out[f"{tag}/obs/image"], out[f"{tag}/obs/tactile"] = obs["image"].numpy().copy(), obs["tactile"].numpy().copy()
        feat = ext({k: v.clone() for k, v in obs.items()})
        feat.square().mean().backward()
        out[f"{tag}/features"] = feat.detach().numpy()
        out[f"{tag}/grad/mae.encoder.transformer.layers.0.0.to_qkv.weight"] = mae.encoder.transformer.layers[0][0].to_qkv.weight.grad.clone().numpy()
        out[f"{tag}/grad/vit_layer.transformer.layers.0.1.net.1.weight"] = ext.vit_layer.transformer.layers[0][1].net[1].weight.grad.clone().numpy()
        out.update({f"{tag}/param/vit_layer." + k: v.detach().clone().numpy() for k, v in ext.vit_layer.state_dict().items() if k.startswith("transformer.")})
        mae.zero_grad()
    out.update({"param/mae." + k: v.detach().clone().numpy() for k, v in mae.state_dict().items()})
    out["meta"] = np.array([32, 16, 8, 4, D, 2, 2, 128, 3 * fs, 2, D, 1, 2, B], dtype=np.int64)
    out["frame_stack"] = np.array(fs)
    np.savez_compressed(os.path.join(HERE, "mae_extractor.npz"), **out)
    print("mae_extractor: features", out["vt/features"][0, :3], out["vision_only/features"][0, :3])


def run_dino_cat_extractor():
    """`MAEExtractor.forward` of the cfg-5 fusion head (models/pretrain_models_dino_cat_mae.py:793-904): get_embeddings -> 1-layer
    Transformer -> mean, concatenated with `dino_model(middle RGB frame)`, through the 3-Linear MLP (eval mode: Dropout off).  The
    frozen DINOv2 is replaced by a small deterministic stand-in (mean-pool + Linear): this fixture pins the GLUE — the frame-stack
    reshape, the middle-frame slice `image[:, 3*mid-3 : 3*mid]`, the concat order and the MLP — not the image encoder
    (tests/golden/dinov2_small.npz does that).  frame_stack = 4 (the script default), so the slice is channels 3..5."""
    spec = importlib.util.spec_from_file_location("ref_dino_cat", os.path.join(REF, "models", "pretrain_models_dino_cat_mae.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    torch.manual_seed(81)
    fs, B, D = 4, 2, 64
    enc = m.VTT(image_size=28, tactile_size=28, image_patch_size=7, tactile_patch_size=7, dim=D, depth=1, heads=2, mlp_dim=128,
                image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = m.VTMAE(encoder=enc, decoder_dim=D, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2,
                  early_conv_masking=False, use_sincosmod_encodings=True, frame_stack=fs)

    class StandInDino(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(3, D)

        def forward(self, x):
            assert x.shape[1] == 3, x.shape
            return self.proj(x.mean(dim=(2, 3)))

    dino = StandInDino()
    ext = m.MAEExtractor(None, dino, mae, D, False, fs).eval()
    g = torch.Generator().manual_seed(82)
    obs = {"image": torch.rand(B, fs, 28, 28, 3, generator=g), "tactile": torch.rand(B, fs, 6, 28, 28, generator=g) * 2 - 1}
    out = {"obs/image": obs["image"].numpy().copy(), "obs/tactile": obs["tactile"].numpy().copy()}
    with torch.no_grad():
        feat = ext({k: v.clone() for k, v in obs.items()})
    out["features"] = feat.numpy()
    out.update({"param/mae." + k: v.detach().clone().numpy() for k, v in mae.state_dict().items()})
    out.update({"param/ext." + k: v.detach().clone().numpy() for k, v in ext.state_dict().items()
                if k.startswith(("vit_layer.transformer.", "mlp.", "query", "key_projection"))})
    out["param/dino.proj.weight"], out["param/dino.proj.bias"] = dino.proj.weight.detach().numpy(), dino.proj.bias.detach().numpy()
    out["meta"] = np.array([28, 28, 7, 7, D, 1, 2, 128, 3 * fs, 2, D, 1, 2, B], dtype=np.int64)
    out["frame_stack"] = np.array(fs)
    np.savez_compressed(os.path.join(HERE, "dino_cat_extractor.npz"), **out)
    print("dino_cat_extractor: features", feat[0, :4].tolist())


def run_vt_load(ref):
    import utils.pretrain_utils as pu  # the reference's own file (cv2 / SB3 logger stubbed)
    g = np.random.default_rng(7)
    for fs in (1, 2):
        obs = {"image": g.random((2, 8, 8, 3 * fs), dtype=np.float32),
               "tactile": (g.random((2, 6 * fs, 4, 4), dtype=np.float32) * 2 - 1)}
        o = pu.vt_load({k: v.copy() for k, v in obs.items()}, frame_stack=fs)
        out = {"in/" + k: v for k, v in obs.items()}
        out.update({"out/" + k: v.numpy() for k, v in o.items()})
        np.savez_compressed(os.path.join(HERE, f"vt_load_fs{fs}.npz"), **out)
        print(f"vt_load_fs{fs}: keys={sorted(o.keys())}")
    # uint8 camera frames + non-default normalisation ranges (utils/pretrain_utils.py:28-30,47-49)
    obs = {"image": g.integers(0, 256, (2, 8, 8, 6), dtype=np.uint8), "tactile": (g.random((2, 12, 4, 4), dtype=np.float32) * 5 - 2)}
    o = pu.vt_load({k: v.copy() for k, v in obs.items()}, image_normalization=[0, 255], tactile_normalization=[-2, 3], frame_stack=2)
    out = {"in/" + k: v for k, v in obs.items()}
    out.update({"out/" + k: v.numpy() for k, v in o.items()})
    np.savez_compressed(os.path.join(HERE, "vt_load_u8.npz"), **out)
    print(f"vt_load_u8: keys={sorted(o.keys())} image dtype {o['image'].dtype}")
    # ranges that are not representable in binary: lo is rounded to fp32 by `tensor - lo`, the span (hi - lo) is formed in Python
    # double and rounded once by `tensor / span` (float(0.3) - float(0.1) != float(0.3 - 0.1) in fp32)
    obs = {"image": g.random((2, 8, 8, 3), dtype=np.float32), "tactile": (g.random((2, 6, 4, 4), dtype=np.float32) * 2 - 1)}
    o = pu.vt_load({k: v.copy() for k, v in obs.items()}, image_normalization=[0.1, 0.3], tactile_normalization=[-0.7, 0.9], frame_stack=1)
    out = {"in/" + k: v for k, v in obs.items()}
    out.update({"out/" + k: v.numpy() for k, v in o.items()})
    np.savez_compressed(os.path.join(HERE, "vt_load_frac.npz"), **out)
    print(f"vt_load_frac: keys={sorted(o.keys())}")


def run_vtt_dino(num_register_tokens=0, name="vtt_dino_small", seed=21):
    """DINO-style encoder `models/VTT.py:77-426` (forward / forward_features / prepare_tokens_with_masks),
    `SinusoidalEmbed` (tactile_ssl/model/layers/patch_embed.py:133-224) and `apply_masks`
    (tactile_ssl/utils/__init__.py:25-36).  `tactile_ssl/model/__init__.py` pulls torchvision through
    pretrained.py, so the `tactile_ssl.model` package object is created bare (its __path__ points at the real
    directory) and `tactile_ssl.model.layers` is imported from the reference's own files."""
    import tactile_ssl  # imports cleanly
    pkg = types.ModuleType("tactile_ssl.model")
    pkg.__path__ = [os.path.join(REF, "tactile_ssl", "model")]
    sys.modules["tactile_ssl.model"] = pkg
    spec = importlib.util.spec_from_file_location("ref_VTT", os.path.join(REF, "models", "VTT.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    torch.manual_seed(seed)
    B, D = 2, 64
    enc = m.VTT(image_size=32, tactile_size=32, image_patch_size=8, tactile_patch_size=8, dim=D, depth=2, heads=2,
                mlp_dim=128, num_tactiles=2, num_register_tokens=num_register_tokens)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, p in enc.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        if num_register_tokens:                 # the reference initialises them with std 1e-6: make them matter in the fixture
            enc.register_tokens.add_(0.5 * torch.randn(enc.register_tokens.shape, generator=g))
    x = {k: torch.rand(B, 3, 32, 32, generator=g) for k in ("image", "tactile1", "tactile2")}
    masks = [torch.stack([torch.randperm(16, generator=g)[:5] for _ in range(B)]),
             torch.stack([torch.randperm(16, generator=g)[:5] for _ in range(B)])]
    enc.eval()
    with torch.no_grad():
        full = enc.forward_features({k: v.clone() for k, v in x.items()})
        mk = enc({k: v.clone() for k, v in x.items()}, masks)
        pos = enc.pos_embed(torch.device("cpu"))
    out = {"param/" + k: v.numpy() for k, v in enc.state_dict().items()}
    out.update({"input/" + k: v.numpy() for k, v in x.items()})
    out["mask/0"], out["mask/1"] = masks[0].numpy(), masks[1].numpy()
    out["pos_embed"] = pos.numpy()
    out["full/x_norm_patchtokens"] = full["x_norm_patchtokens"].numpy()
    out["full/x_norm_regtokens"] = full["x_norm_regtokens"].numpy()
    out["full/x_prenorm"] = full["x_prenorm"].numpy()
    out["masked/x_norm_patchtokens"] = mk.numpy()
    emb = torch.rand(B, 16, 8, generator=g)
    out["apply_masks/in"] = emb.numpy()
    out["apply_masks/out"] = m.apply_masks(emb, masks).numpy()
    out["meta"] = np.array([32, 8, D, 2, 2, 128, B], dtype=np.int64)
    out["num_register_tokens"] = np.array(num_register_tokens, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name + ": patchtokens", tuple(mk.shape), "pos", tuple(pos.shape), "registers", num_register_tokens)


def _ref_layers():
    """`tactile_ssl.model.layers` of the reference, loaded from its own files: `tactile_ssl/model/__init__.py` pulls torchvision
    through pretrained.py, so the `tactile_ssl.model` package object is created bare (its __path__ points at the real directory)."""
    import tactile_ssl  # noqa: F401  imports cleanly
    if "tactile_ssl.model" not in sys.modules:
        pkg = types.ModuleType("tactile_ssl.model")
        pkg.__path__ = [os.path.join(REF, "tactile_ssl", "model")]
        sys.modules["tactile_ssl.model"] = pkg
    os.environ["XFORMERS_DISABLED"] = "1"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module("tactile_ssl.model.layers.block")


def run_block_stack():
    """The reference's OWN pre-norm transformer block — tactile_ssl/model/layers/block.py:43-114 (`Block.forward`: x + attn(norm1(x)),
    x + mlp(norm2(x))), attention.py:54-76 (bias-free fused qkv, q|k|v head-major, q scaled by head_dim^-0.5, softmax, proj) and
    mlp.py:34-40 (fc1, GELU, fc2) — as a 2-layer stack + nn.LayerNorm on random tokens at the MAE's two sequence lengths (48 visible /
    192 decoder tokens).  It is the same arithmetic as vit-pytorch's Transformer with the parameter names mapped
    (norm1 -> layers.i.0.norm, attn.qkv -> to_qkv, attn.proj -> to_out.0, norm2 -> net.0, mlp.fc1 -> net.1, mlp.fc2 -> net.4), so this
    fixture pins the attention / GELU-MLP / LayerNorm kernels and the oracle's `transformer` to reference-held code."""
    blk = _ref_layers()
    torch.manual_seed(41)
    D, heads, depth, ratio = 128, 2, 2, 2.0
    blocks = torch.nn.ModuleList([blk.Block(dim=D, num_heads=heads, mlp_ratio=ratio, qkv_bias=False) for _ in range(depth)])
    norm = torch.nn.LayerNorm(D)
    g = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for p in list(blocks.parameters()) + list(norm.parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    out = {"meta": np.array([D, depth, heads, int(D * ratio)], dtype=np.int64)}
    names = {}
    for i, b in enumerate(blocks):
        names.update({f"layers.{i}.0.norm.weight": b.norm1.weight, f"layers.{i}.0.norm.bias": b.norm1.bias,
                      f"layers.{i}.0.to_qkv.weight": b.attn.qkv.weight, f"layers.{i}.0.to_out.0.weight": b.attn.proj.weight,
                      f"layers.{i}.0.to_out.0.bias": b.attn.proj.bias, f"layers.{i}.1.net.0.weight": b.norm2.weight,
                      f"layers.{i}.1.net.0.bias": b.norm2.bias, f"layers.{i}.1.net.1.weight": b.mlp.fc1.weight,
                      f"layers.{i}.1.net.1.bias": b.mlp.fc1.bias, f"layers.{i}.1.net.4.weight": b.mlp.fc2.weight,
                      f"layers.{i}.1.net.4.bias": b.mlp.fc2.bias})
    names.update({"norm.weight": norm.weight, "norm.bias": norm.bias})
    assert len(names) == len(list(blocks.parameters())) + 2      # every parameter of the stack is mapped (no LayerScale: init_values=None)
    out.update({"param/" + k: v.detach().numpy().copy() for k, v in names.items()})
    for n, B in ((48, 3), (192, 2)):
        x = (torch.randn(B, n, D, generator=g) * 1.5).requires_grad_(True)
        cot = torch.randn(B, n, D, generator=g)
        for p in names.values():
            p.grad = None
        h = x
        mids = []
        for b in blocks:
            h = b(h)
            mids.append(h)
        y = norm(h)
        (y * cot).sum().backward()
        out[f"n{n}/x"], out[f"n{n}/cot"], out[f"n{n}/y"] = x.detach().numpy(), cot.numpy(), y.detach().numpy()
        out[f"n{n}/block0_out"] = mids[0].detach().numpy()
        out[f"n{n}/dx"] = x.grad.numpy()
        out.update({f"n{n}/grad/" + k: v.grad.numpy().copy() for k, v in names.items()})
        print(f"block_stack n={n}: y {tuple(y.shape)} |y| {float(y.abs().mean()):.4f}")
    np.savez_compressed(os.path.join(HERE, "block_stack.npz"), **out)


def main():
    ref = _load_reference()
    if "--ppo-only" in sys.argv:
        run_ppo_like(ref)
        return
    if "--trainiter-only" in sys.argv:
        run_train_iterations(ref)
        return
    if "--block-only" in sys.argv:
        run_block_stack()
        return
    if "--vtload-only" in sys.argv:
        run_vt_load(ref)
        return
    if "--dinoreg-only" in sys.argv:
        run_vtt_dino(num_register_tokens=4, name="vtt_dino_reg", seed=23)
        return
    if "--learnedpos-only" in sys.argv:
        run_case(ref, "vt_learnedpos", image_hw=32, tactile_hw=16, ip=8, tp_=4, dim=64, depth=1, heads=2, mlp=128, C=3,
                 num_tactiles=2, dec_dim=64, dec_depth=1, dec_heads=2, ratio=0.75, B=3, seed=16, with_embeddings=True, sincos=False)
        return
    if "--extractor-only" in sys.argv:
        run_extractor(ref)
        run_dino_cat_extractor()
        return
    if "--reconstruct-only" not in sys.argv:
        _main_cases(ref)
        run_ppo_like(ref)
        run_train_iterations(ref)
        run_extractor(ref)
        run_dino_cat_extractor()
    # F: reconstruct(): count rule int(r*n) (0.7*16 -> 11 image, 11 per sensor), both masking modes, default + vision-only
    run_reconstruct(ref, "recon_small", early_conv=False, ratio=0.75, mask_ratio=0.7, seed=31)
    run_reconstruct(ref, "recon_default_ratio", early_conv=False, ratio=0.8, mask_ratio=None, seed=32, use_tactile=False)
    run_reconstruct(ref, "recon_earlyconv", early_conv=True, ratio=0.75, mask_ratio=0.5, seed=33)


def _main_cases(ref):
    # A: vision + 2 tactile, the cfg-2 structure at reduced widths (D == dd -> enc_to_dec is Identity)
    run_case(ref, "vt_small", image_hw=32, tactile_hw=16, ip=8, tp_=4, dim=64, depth=2, heads=2, mlp=128, C=3,
             num_tactiles=2, dec_dim=64, dec_depth=1, dec_heads=2, ratio=0.75, B=3, seed=11, with_embeddings=True)
    # B: vision only (cfg-1 structure), num_tactiles = 0
    run_case(ref, "v_only_small", image_hw=32, tactile_hw=16, ip=8, tp_=4, dim=64, depth=2, heads=1, mlp=128, C=3,
             num_tactiles=0, dec_dim=64, dec_depth=1, dec_heads=1, ratio=0.75, B=2, seed=12)
    # C: dd != D (enc_to_dec Linear + truncated decoder sincos), frame_stack-like channels, reference mask ratio
    run_case(ref, "vt_decdim", image_hw=32, tactile_hw=16, ip=8, tp_=4, dim=128, depth=1, heads=2, mlp=256, C=6,
             num_tactiles=2, dec_dim=64, dec_depth=2, dec_heads=1, ratio=0.95, B=2, seed=13, with_embeddings=True)
    # D: cfg-2 token geometry exactly (64x64/P8 + 2x 32x32/P4 -> 192 tokens, 48 visible), narrow model
    run_case(ref, "vt_cfg2_geom", image_hw=64, tactile_hw=32, ip=8, tp_=4, dim=64, depth=1, heads=1, mlp=64, C=3,
             num_tactiles=2, dec_dim=64, dec_depth=1, dec_heads=1, ratio=0.75, B=2, seed=14)
    # E: the reference's DEFAULT flag early_conv_masking=True (EarlyCNN stem, loss over all patches), mask 0.95
    run_case(ref, "vt_earlyconv", image_hw=32, tactile_hw=16, ip=8, tp_=4, dim=64, depth=1, heads=2, mlp=128, C=3,
             num_tactiles=2, dec_dim=64, dec_depth=1, dec_heads=2, ratio=0.75, B=3, seed=15, with_embeddings=True, early_conv=True)
    # G: use_sincosmod_encodings=False (learned encoder positions, decoder_pos_emb at every decoder position; :218-219,280-287)
    run_case(ref, "vt_learnedpos", image_hw=32, tactile_hw=16, ip=8, tp_=4, dim=64, depth=1, heads=2, mlp=128, C=3,
             num_tactiles=2, dec_dim=64, dec_depth=1, dec_heads=2, ratio=0.75, B=3, seed=16, with_embeddings=True, sincos=False)
    run_vt_load(ref)
    run_vtt_dino()
    run_vtt_dino(num_register_tokens=4, name="vtt_dino_reg", seed=23)      # models/VTT.py:166-172,305-312
    run_block_stack()


if __name__ == "__main__":
    main()
