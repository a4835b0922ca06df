"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/m3l_amd.h declares, the
drop-in modules expose the reference's state-dict keys / shapes / init, and the host-only ABI calls (mask counts, error
convention) behave.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import m3l_amd
from m3l_amd import _lib as L
from m3l_amd import functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "m3l_amd.h")).read()
    declared = set(re.findall(r"\b(m3l_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"m3l_geom", "m3l_tf_cfg"}
    assert len(declared) >= 25
    lib = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/m3l_amd.h but not exported"
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    assert L.lib().m3l_version() >= 100


def test_error_convention_no_throw_across_abi():
    g = L.Geom(64, 64, 7, 3, 32, 32, 4, 3, 2, 1, 1)       # 64 % 7 != 0
    out = (C.c_int * 6)()
    assert L.lib().m3l_mask_counts(C.byref(g), 0.75, out) != 0
    assert "divisible by the patch size" in L.last_error()
    g = L.Geom(64, 64, 8, 3, 32, 32, 4, 3, 2, 1, 1)
    assert L.lib().m3l_mask_counts(C.byref(g), 1.5, out) != 0
    assert "masking ratio" in L.last_error()


@pytest.mark.parametrize("ratio,n_img_hw,ip,n_tac_hw,tp,k,expect", [
    (0.95, 64, 8, 32, 4, 2, (182, 10, 60, 61)),      # reference defaults: 60 image + 61 + 61
    (0.75, 64, 8, 32, 4, 2, (144, 48, 48, 48)),      # cfg 2
    (0.75, 64, 8, 32, 4, 0, (48, 16, 48, 0)),        # cfg 1 (vision only)
    (0.75, 224, 16, 64, 8, 4, (339, 113, 147, 48)),  # cfg 4
    (0.8, 70, 14, 70, 14, 2, (60, 15, 20, 20)),      # cfg 5
])
def test_mask_counts_match_reference_integer_rule(ratio, n_img_hw, ip, n_tac_hw, tp, k, expect):
    g = L.Geom(n_img_hw, n_img_hw, ip, 3, n_tac_hw, n_tac_hw, tp, 3, k, 1, 1)
    c = Fn.mask_counts(g, ratio)
    assert (c["num_masked"], c["num_unmasked"], c["nm_img"], c["nm_tac"]) == expect


def _build(z):
    image_hw, tactile_hw, ip, tp, dim, depth, heads, mlp, Cc, k, dd, ddepth, dheads, B = [int(v) for v in z["meta"]]
    enc = m3l_amd.VTT(image_size=image_hw, tactile_size=tactile_hw, image_patch_size=ip, tactile_patch_size=tp, dim=dim,
                      depth=depth, heads=heads, mlp_dim=mlp, image_channels=Cc, tactile_channels=Cc, num_tactiles=k)
    return m3l_amd.VTMAE(encoder=enc, decoder_dim=dd, masking_ratio=float(z["ratio"]), decoder_depth=ddepth,
                         decoder_heads=dheads, num_tactiles=k, early_conv_masking=bool(int(z["early_conv"])))


@pytest.mark.parametrize("name,seed", [("vt_small", 11), ("v_only_small", 12), ("vt_decdim", 13), ("vt_earlyconv", 15)])
def test_state_dict_keys_shapes_and_init_match_reference(golden_dir, name, seed):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    torch.manual_seed(seed)
    mae = _build(z)
    sd = mae.state_dict()
    ref_keys = {k[len("param/"):] for k in z.files if k.startswith("param/")}
    assert set(sd.keys()) == ref_keys, set(sd.keys()) ^ ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == z["param/" + k].shape, k
    # same construction order as the reference -> same RNG stream -> identical initial matrices (the fixture only
    # perturbed 1-D parameters after construction) and identical sincos buffers
    for k in ref_keys:
        if z["param/" + k].ndim >= 2 and not k.endswith("pos_embedding"):
            np.testing.assert_array_equal(sd[k].numpy(), z["param/" + k], err_msg=k)
        if k.endswith("_pos_embedding") and k != "encoder.pos_embedding":
            np.testing.assert_allclose(sd[k].numpy(), z["param/" + k], atol=1e-6, err_msg=k)
    mae.load_state_dict({k: torch.tensor(z["param/" + k]) for k in ref_keys}, strict=True)


def test_reference_attribute_surface():
    enc = m3l_amd.VTT(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=192, depth=2, heads=3, mlp_dim=768)
    assert enc.pos_embedding.shape == (1, 64 + 128 + 1, 192)
    assert enc.image_to_patch_embedding[2].weight.shape[-1] == 192 and enc.tactile_to_patch_embedding[2].weight.shape[-1] == 48
    for a in ("image_channels", "tactile_channels", "image_height", "image_width", "tactile_height", "tactile_width",
              "image_patch_height", "image_patch_width", "tactile_patch_height", "tactile_patch_width", "transformer"):
        assert hasattr(enc, a)
    patches = enc.image_to_patch_embedding[0](torch.arange(2 * 3 * 64 * 64, dtype=torch.float32).reshape(2, 3, 64, 64))
    assert patches.shape == (2, 64, 192)
    mae = m3l_amd.VTMAE(encoder=enc, decoder_dim=192, masking_ratio=0.75, decoder_depth=1, decoder_heads=3)
    assert isinstance(mae.enc_to_dec, torch.nn.Identity)
    opt = torch.optim.Adam(mae.parameters(), lr=1e-4)      # ppo_mae.py:182-183
    assert len(opt.param_groups[0]["params"]) > 20
    mae.initialize_training({"lr": 1e-4, "batch_size": 8})
    with pytest.raises(AssertionError):
        m3l_amd.VTMAE(encoder=enc, decoder_dim=192, masking_ratio=1.0)
    with pytest.raises(AssertionError):
        m3l_amd.VTT(image_size=64, tactile_size=32, image_patch_size=7, tactile_patch_size=4, dim=192, depth=1, heads=3, mlp_dim=8)


def test_no_cpu_fallback():
    enc = m3l_amd.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128)
    mae = m3l_amd.VTMAE(encoder=enc, decoder_dim=64, decoder_depth=1, decoder_heads=2)
    x = {"image": torch.rand(2, 3, 32, 32), "tactile1": torch.rand(2, 3, 16, 16), "tactile2": torch.rand(2, 3, 16, 16)}
    with pytest.raises(m3l_amd.M3LError):
        mae(x)


def test_dino_vtt_state_dict_and_pos_table(golden_dir):
    z = np.load(os.path.join(golden_dir, "vtt_dino_small.npz"))
    hw, p, D, depth, heads, mlp, B = [int(v) for v in z["meta"]]
    enc = m3l_amd.DinoVTT(image_size=hw, tactile_size=hw, image_patch_size=p, tactile_patch_size=p, dim=D, depth=depth, heads=heads,
                          mlp_dim=mlp, num_tactiles=2, num_register_tokens=0)
    ref_keys = {k[len("param/"):] for k in z.files if k.startswith("param/")}
    sd = enc.state_dict()
    assert set(sd.keys()) == ref_keys, set(sd.keys()) ^ ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == z["param/" + k].shape, k
    np.testing.assert_allclose(enc.pos_embed(torch.device("cpu")).numpy(), z["pos_embed"], atol=1e-6)
    # timm-style init: LayerNorm ones/zeros, Linear bias zeros, Linear weight ~ N(0, 0.02) (trunc at +-2 absolute)
    assert float(enc.norm.weight.min()) == 1.0 and float(enc.transformer.layers[0][1].net[1].bias.abs().max()) == 0.0
    assert abs(float(enc.image_to_patch_embedding[2].weight.std()) - 0.02) < 0.002
    enc.load_state_dict({k: torch.tensor(z["param/" + k]) for k in ref_keys}, strict=True)


def test_dinov2_frozen_surface(golden_dir):
    """DinoV2Frozen: hub-checkpoint state-dict names / shapes (the fixture stores them under those names), everything frozen,
    no CPU execution path."""
    import os
    import numpy as np
    import pytest
    import torch
    from m3l_amd import DinoV2Frozen, M3LError
    z = np.load(os.path.join(golden_dir, "dinov2_small.npz"))
    dim, depth, heads, patch, img, reg = [int(v) for v in z["meta"]]
    m = DinoV2Frozen(embed_dim=dim, depth=depth, num_heads=heads, patch_size=patch, img_size=img, num_register_tokens=reg)
    want = {k[len("param/"):]: tuple(z[k].shape) for k in z.files if k.startswith("param/")}
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    m.load_state_dict({k: torch.tensor(z["param/" + k]) for k in want}, strict=True)
    assert not any(p.requires_grad for p in m.parameters()) and not m.training
    with pytest.raises(M3LError):
        m(torch.zeros(1, 3, 70, 70))
    full = DinoV2Frozen()                                   # dinov2_vits14_reg sizes
    assert sum(p.numel() for p in full.parameters()) == 22_058_112   # == transformers' Dinov2WithRegistersModel at ViT-S/14, 4 registers


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/m3l_amd.h compiles as C99 and as C++11 on its own, and examples/c_host.c — a C program that binds the ABI without any
    Python — builds against the shared library and reproduces the oracle's mask counts (host-only entry points: no GPU needed)."""
    import os
    import subprocess
    from oracle import vtmae_oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "m3l_amd.h")
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], check=True)
    subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-std=c++11", "-Wall", "-Werror", hdr], check=True)
    import __graft_entry__ as ge
    ge.build()
    libdir = os.path.join(root, "m3l_amd", "lib")
    exe = str(tmp_path / "c_host")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_host.c"),
                    "-L" + libdir, "-lm3l_amd", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    nm, nmi, nmt = O.mask_counts(0.75, 64, 128, 2)
    assert f"cfg2 mask 0.75: masked {nm} unmasked {192 - nm} (image {nmi}, per sensor {nmt})" in out
    nm, nmi, nmt = O.mask_counts(0.95, 64, 128, 2)
    assert f"ref  mask 0.95: masked {nm} unmasked {192 - nm} (image {nmi}, per sensor {nmt})" in out
    assert "error convention:" in out and "must be a multiple" in out


def test_comm_entry_points_fail_loudly_before_init():
    """comm.hip (gradient all-reduce by RCCL on the library's side stream): no communicator before m3l_comm_init, and an all-reduce without
    one returns an error code with a message instead of touching anything."""
    from m3l_amd import _lib as L
    lib = L.lib()
    assert lib.m3l_comm_world() == 0
    rc = lib.m3l_comm_allreduce(None, 0, None)
    assert rc != 0 and "m3l_comm_init" in L.last_error()
    assert lib.m3l_comm_destroy() == 0


def test_transformer_tensor_cache_follows_replaced_parameters():
    """Transformer._tensors() caches the Parameter objects; a parameter replaced after the first call must be picked up."""
    import torch
    from m3l_amd.pretrain_models import Transformer
    tf = Transformer(dim=64, depth=2, heads=2, dim_head=64, mlp_dim=128)
    first = tf._tensors()
    assert tf._tensors()[2] is first[2] is tf.layers[0][0].to_qkv.weight
    new = torch.nn.Parameter(torch.zeros_like(tf.layers[1][0].to_qkv.weight))
    tf.layers[1][0].to_qkv.weight = new
    again = tf._tensors()
    assert again[11 + 2] is new and again[2] is first[2]
    tf.norm.bias = torch.nn.Parameter(torch.ones_like(tf.norm.bias))
    assert tf._tensors()[-1] is tf.norm.bias
