"""Pins the CPU oracle (oracle/vtmae_oracle.py) against the golden vectors produced by executing the reference's
own VTT / VTMAE / vt_load (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vtmae_oracle as O

CASES = ["vt_small", "v_only_small", "vt_decdim", "vt_cfg2_geom", "vt_earlyconv", "vt_learnedpos"]


def _sincos(z):
    return bool(int(z["sincos"])) if "sincos" in z.files else True


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _inputs(z):
    x = {k[len("input/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("input/")}
    noises = [torch.tensor(z[f"noise/{i}"]) for i in range(len([k for k in z.files if k.startswith("noise/")]))]
    return x, noises


@pytest.mark.parametrize("name", CASES)
def test_mask_indices_bit_exact(golden_dir, name):
    z = _load(golden_dir, name)
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    _, noises = _inputs(z)
    # On tie-free rows the stable argsort IS what torch.argsort produced inside the reference run.  On rows with
    # float32 ties the reference's order of the tied keys is unspecified (argsort without stable=True; torch's CPU
    # sort is observably unstable on 64-wide rows): there we require the same sorted key sequence and the stable
    # (ascending-index) tie-break of our contract == torch.argsort(stable=True).
    saw_tie = False
    for i, nz in enumerate(noises):
        a = nz.numpy()
        mine, ref = O.stable_argsort_rows(a), z[f"argsort/{i}"]
        assert np.array_equal(mine, nz.argsort(dim=-1, stable=True).numpy())
        for r in range(a.shape[0]):
            if len(np.unique(a[r])) == a.shape[1]:
                assert np.array_equal(mine[r], ref[r])
            else:
                saw_tie = True
                assert np.array_equal(a[r][mine[r]], a[r][ref[r]])
                assert np.array_equal(np.sort(ref[r]), np.arange(a.shape[1]))
    assert saw_tie, "fixtures are expected to contain crafted tied rows"
    masked, unmasked, nm_img, nm_tac = O.mask_indices([n.numpy() for n in noises], cfg.ratio, cfg.n_img, cfg.n_tac, cfg.num_tactiles)
    assert masked.dtype == np.int64 and unmasked.dtype == np.int64
    # counts: reference shapes (decoder gathers; with early_conv_masking the heads see ALL tokens instead)
    if not int(z["early_conv"]):
        assert masked.shape[1] == z["cap/to_pixels_in"].shape[1] + z["cap/to_tactiles_in"].shape[1]
    assert unmasked.shape[1] == z["cap/encoder_in"].shape[1]
    both = np.sort(np.concatenate([masked, unmasked], 1), 1)
    assert np.array_equal(both, np.broadcast_to(np.arange(cfg.n_total), both.shape))


def test_mask_counts_python_double_semantics():
    # r=.95, N=192 -> 182 masked, 60 image, 61 per tactile
    assert O.mask_counts(0.95, 64, 128, 2) == (182, 60, 61)
    assert O.mask_counts(0.75, 64, 128, 2) == (144, 48, 48)
    assert O.mask_counts(0.75, 64, 0, 0) == (48, 48, 0)
    assert O.mask_counts(0.75, 196, 256, 4) == (339, 147, 48)
    assert O.mask_counts(0.8, 25, 50, 2) == (60, 20, 20)


@pytest.mark.parametrize("name", CASES)
def test_forward_intermediates_and_loss(golden_dir, name):
    z = _load(golden_dir, name)
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    P = O.load_fixture_params(z)
    x, noises = _inputs(z)
    perms = [z[f"argsort/{i}"] for i in range(len(noises))]     # the permutations of THIS reference run
    with torch.no_grad():
        r = O.vtmae_forward(P, cfg, x, noises, perms=perms, sincos=_sincos(z))
        r_stable = O.vtmae_forward(P, cfg, x, noises, sincos=_sincos(z))
    # tie order only permutes rows inside the masked lists here -> the loss must not depend on it
    assert abs(float(r_stable["loss"]) - float(r["loss"])) <= 1e-6 * abs(float(r["loss"]))
    tol = dict(rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(r["encoder_in"].numpy(), z["cap/encoder_in"], **tol)
    np.testing.assert_allclose(r["encoder_out"].numpy(), z["cap/encoder_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r["decoder_in"].numpy(), z["cap/decoder_in"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r["decoder_out"].numpy(), z["cap/decoder_out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(r["pred_pixel"].numpy(), z["cap/to_pixels_out"], rtol=1e-4, atol=2e-5)
    if cfg.num_tactiles:
        np.testing.assert_allclose(r["pred_tactile"].numpy(), z["cap/to_tactiles_out"], rtol=1e-4, atol=2e-5)
    assert abs(float(r["loss"]) - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))


@pytest.mark.parametrize("name", CASES)
def test_backward_grads(golden_dir, name):
    z = _load(golden_dir, name)
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    P = O.load_fixture_params(z, requires_grad=True)
    x, noises = _inputs(z)
    O.vtmae_forward(P, cfg, x, noises, sincos=_sincos(z))["loss"].backward()
    unused = set(str(u) for u in z["unused_params"])
    checked = 0
    for k in z.files:
        if not k.startswith("grad/"):
            continue
        name_ = k[len("grad/"):]
        g = P[name_].grad
        assert g is not None, name_
        ref = z[k]
        scale = max(1e-6, float(np.abs(ref).max()))
        assert float(np.abs(g.numpy() - ref).max()) <= 2e-4 * scale + 1e-7, name_
        checked += 1
    assert checked > 20
    # parameters the reference leaves without grad (sincos mode) must also be unused in the restatement
    for u in unused:
        assert P[u].grad is None, u


@pytest.mark.parametrize("name", ["vt_small", "vt_decdim", "vt_earlyconv", "vt_learnedpos"])
def test_get_embeddings(golden_dir, name):
    z = _load(golden_dir, name)
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    P = O.load_fixture_params(z)
    x, _ = _inputs(z)
    with torch.no_grad():
        e = O.get_embeddings(P, cfg, x, sincos=_sincos(z))
    np.testing.assert_allclose(e.numpy(), z["embeddings"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("fs", [1, 2])
def test_vt_load(golden_dir, fs):
    z = _load(golden_dir, f"vt_load_fs{fs}")
    out = O.vt_load({"image": z["in/image"], "tactile": z["in/tactile"]}, frame_stack=fs)
    assert sorted(out.keys()) == sorted(k[4:] for k in z.files if k.startswith("out/"))
    for k, v in out.items():
        assert v.dtype == torch.float32
        np.testing.assert_array_equal(v.numpy(), z["out/" + k])


def test_vt_load_uint8_and_ranges(golden_dir):
    z = _load(golden_dir, "vt_load_u8")            # uint8 image, normalisation [0, 255] / [-2, 3], frame_stack 2
    out = O.vt_load({"image": z["in/image"], "tactile": z["in/tactile"]}, frame_stack=2, image_normalization=[0, 255], tactile_normalization=[-2, 3])
    for k, v in out.items():
        np.testing.assert_array_equal(v.numpy(), z["out/" + k])
    z = _load(golden_dir, "vt_load_frac")          # ranges binary floats cannot represent: [0.1, 0.3] / [-0.7, 0.9]
    out = O.vt_load({"image": z["in/image"], "tactile": z["in/tactile"]}, image_normalization=[0.1, 0.3], tactile_normalization=[-0.7, 0.9])
    for k, v in out.items():
        np.testing.assert_array_equal(v.numpy(), z["out/" + k])


def test_sincos_buffers_match_reference_buffers(golden_dir):
    z = _load(golden_dir, "vt_decdim")     # D=128, dd=64: decoder code = first dd channels of the D-parameterised code
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    g = cfg.image_hw // cfg.image_patch
    np.testing.assert_allclose(O.sincos_2d(cfg.dim, g, g, cfg.dim).numpy(), z["param/image_enc_pos_embedding"][0], atol=1e-6)
    np.testing.assert_allclose(O.sincos_2d(cfg.dim, g, g, cfg.dec_dim).numpy(), z["param/image_dec_pos_embedding"][0], atol=1e-6)
    gt = cfg.tactile_hw // cfg.tactile_patch
    tac = O.sincos_2d(cfg.dim, gt, gt, cfg.dim).repeat(cfg.num_tactiles, 1)
    np.testing.assert_allclose(tac.numpy(), z["param/tactile_enc_pos_embedding"][0], atol=1e-6)


@pytest.mark.parametrize("fixture", ["vtt_dino_small", "vtt_dino_reg"])
def test_dino_style_vtt(golden_dir, fixture):
    z = _load(golden_dir, fixture)
    hw, p, D, depth, heads, mlp, B = [int(v) for v in z["meta"]]
    P = O.load_fixture_params(z)
    x = {k[len("input/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("input/")}
    np.testing.assert_allclose(O.sinusoidal_embed((3 * hw // p, hw // p), D).numpy(), z["pos_embed"], atol=1e-6)
    masks = [torch.tensor(z["mask/0"]), torch.tensor(z["mask/1"])]
    np.testing.assert_array_equal(O.apply_masks(torch.tensor(z["apply_masks/in"]), masks).numpy(), z["apply_masks/out"])
    with torch.no_grad():
        full = O.vtt_dino_forward(P, image_patch=p, tactile_patch=p, depth=depth, heads=heads, x=x)
        mk = O.vtt_dino_forward(P, image_patch=p, tactile_patch=p, depth=depth, heads=heads, x=x, masks=masks)
    np.testing.assert_allclose(full["x_prenorm"].numpy(), z["full/x_prenorm"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(full["x_norm_patchtokens"].numpy(), z["full/x_norm_patchtokens"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(mk["x_norm_patchtokens"].numpy(), z["masked/x_norm_patchtokens"], rtol=1e-4, atol=2e-5)
    if fixture == "vtt_dino_reg":                              # num_register_tokens = 4 (models/VTT.py:166-172,305-312)
        assert full["x_norm_regtokens"].shape[1] == 4 == int(z["num_register_tokens"])
        np.testing.assert_allclose(full["x_norm_regtokens"].numpy(), z["full/x_norm_regtokens"], rtol=1e-4, atol=2e-5)


RECON = ["recon_small", "recon_default_ratio", "recon_earlyconv"]


@pytest.mark.parametrize("name", RECON)
def test_reconstruct(golden_dir, name):
    """VTMAE.reconstruct (pretrain_models.py:344-586): frames bit-for-position (0.5 / inf fills), predictions and MSEs."""
    z = _load(golden_dir, name)
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    P = O.load_fixture_params(z)
    x, noises = _inputs(z)
    mr = float(z["mask_ratio"])
    with torch.no_grad():
        r = O.reconstruct(P, cfg, x, noises, mask_ratio=None if mr < 0 else mr, use_tactile=bool(int(z["use_tactile"])))
    want = sorted(k[4:] for k in z.files if k.startswith("out/"))
    assert sorted(r.keys()) == want
    for k in want:
        a, b = r[k].numpy(), z["out/" + k]
        assert a.shape == b.shape, k
        if k.endswith("_masked"):
            np.testing.assert_array_equal(a, b)               # raw pixels + fill values: exact
        else:
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5, err_msg=k)


def test_dinov2_oracle_vs_independent_implementation(golden_dir):
    """oracle/dinov2_oracle.py (restated facebookresearch/dinov2 ViT with registers) against transformers' implementation of the
    same architecture (tests/golden/make_golden_dinov2.py): tokens incl. interpolated positions, all normed tokens, CLS feature."""
    from oracle import dinov2_oracle as D
    z = _load(golden_dir, "dinov2_small")
    dim, depth, heads, patch, _, _ = [int(v) for v in z["meta"]]
    P = {k[len("param/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    with torch.no_grad():
        r = D.dinov2_forward(P, torch.tensor(z["input/x"]), patch=patch, depth=depth, heads=heads)
    np.testing.assert_allclose(r["tokens_in"].numpy(), z["out/tokens_in"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["x_norm"].numpy(), z["out/x_norm"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r["cls"].numpy(), z["out/cls"], rtol=1e-4, atol=1e-5)


def test_ppo_like_update_with_adam(golden_dir):
    """The reference's MAE update as PPO_MAE.train runs it (models/ppo_mae.py:232-266: reshape frame-stacked observations -> vt_load ->
    loss -> backward -> Adam.step), two mini-batches: the oracle + torch.optim.Adam reproduce both losses and the stepped weights."""
    z = _load(golden_dir, "ppo_like_step")
    fs = int(z["frame_stack"])
    cfg = O.cfg_from_meta(z["meta"], 0.75)
    P = {k[len("param0/"):]: torch.tensor(z[k]).clone().requires_grad_(torch.tensor(z[k]).dtype.is_floating_point)
         for k in z.files if k.startswith("param0/")}
    trainable = [k for k, v in P.items() if v.requires_grad and not k.endswith("pos_embedding") and "pos_emb" not in k]
    opt = torch.optim.Adam([P[k] for k in trainable], lr=1e-3)
    img = z["obs/image"].transpose(0, 2, 3, 1, 4).reshape(6, 32, 32, -1)
    tac = z["obs/tactile"].reshape(6, -1, 16, 16)
    for i in range(2):
        opt.zero_grad()
        x = O.vt_load({"image": img[3 * i:3 * i + 3], "tactile": tac[3 * i:3 * i + 3]}, frame_stack=fs)
        noises = [torch.tensor(z[f"noise/{i}/{j}"]) for j in range(3)]
        loss = O.vtmae_forward(P, cfg, x, noises)["loss"]
        loss.backward()
        opt.step()
        assert abs(float(loss) - float(z["losses"][i])) <= 2e-5 * float(z["losses"][i])
        for k in [f[len(f"step{i}/"):] for f in z.files if f.startswith(f"step{i}/")]:
            np.testing.assert_allclose(P[k].detach().numpy(), z[f"step{i}/{k}"], rtol=2e-4, atol=2e-6, err_msg=k)


def test_train_iterations_adamw_clip(golden_dir):
    """The reference's VTMAE.initialize_training + train_iterations (models/pretrain_models.py:673-715: random.choices from a replay
    buffer -> stack -> vt_load -> loss -> backward -> clip_grad_norm_(0.5) -> AdamW.step), one iteration per call, twice: the oracle +
    torch's clip_grad_norm_ + torch.optim.AdamW reproduce the weights after each iteration (the clip is active: image values up to 40)."""
    z = _load(golden_dir, "train_iterations")
    fs = int(z["frame_stack"])
    cfg = O.cfg_from_meta(z["meta"], 0.75)
    P = {k[len("param0/"):]: torch.tensor(z[k]).clone().requires_grad_(torch.tensor(z[k]).dtype.is_floating_point)
         for k in z.files if k.startswith("param0/")}
    trainable = [k for k, v in P.items() if v.requires_grad and not k.endswith("pos_embedding") and "pos_emb" not in k]
    opt = torch.optim.AdamW([P[k] for k in trainable], lr=1e-3)
    nbuf = len([k for k in z.files if k.startswith("buf/") and k.endswith("/image")])
    for it in range(2):
        idx = z["choices"][it]
        img = np.stack([z[f"buf/{j}/image"] for j in idx]).transpose(0, 2, 3, 1, 4)
        img = img.reshape(img.shape[0], img.shape[1], img.shape[2], -1)
        tac = np.stack([z[f"buf/{j}/tactile"] for j in idx])
        tac = tac.reshape(tac.shape[0], -1, tac.shape[3], tac.shape[4])
        opt.zero_grad()
        x = O.vt_load({"image": img, "tactile": tac}, frame_stack=fs)
        noises = [torch.tensor(z[f"noise/{it}/{j}"]) for j in range(3)]
        O.vtmae_forward(P, cfg, x, noises)["loss"].backward()
        norm = torch.nn.utils.clip_grad_norm_([P[k] for k in trainable], 0.5)
        assert float(norm) > 0.5                                   # the clip bites
        opt.step()
        for k in [f[len(f"iter{it}/"):] for f in z.files if f.startswith(f"iter{it}/")]:
            np.testing.assert_allclose(P[k].detach().numpy(), z[f"iter{it}/{k}"], rtol=2e-4, atol=2e-6, err_msg=k)
    assert nbuf == 6


@pytest.mark.parametrize("tag", ["vt", "vision_only"])
def test_mae_extractor(golden_dir, tag):
    """The reference's MAEExtractor.forward (models/pretrain_models.py:788-841): vt_load -> get_embeddings -> 1-layer Transformer ->
    mean over tokens, with and without tactile control, composed from the oracle pieces."""
    z = _load(golden_dir, "mae_extractor")
    fs = int(z["frame_stack"])
    cfg = O.cfg_from_meta(z["meta"], 0.75)
    P = {k[len("param/mae."):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/mae.")}
    PL = {k[len(tag + "/param/vit_layer."):]: torch.tensor(z[k]) for k in z.files if k.startswith(tag + "/param/vit_layer.")}
    img = z[tag + "/obs/image"].transpose(0, 2, 3, 1, 4).reshape(3, 32, 32, -1)
    tac = z[tag + "/obs/tactile"].reshape(3, -1, 16, 16)
    x = O.vt_load({"image": img, "tactile": tac}, frame_stack=fs)
    with torch.no_grad():
        tok = O.get_embeddings(P, cfg, x, use_tactile=(tag == "vt"))
        feat = O.transformer(tok, PL, "transformer.", 1, 4, 64).mean(1)
    np.testing.assert_allclose(feat.numpy(), z[tag + "/features"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n", [48, 192])
def test_transformer_vs_reference_held_block(golden_dir, n):
    """oracle.transformer (the vit-pytorch restatement every other fixture leans on) against the reference's OWN pre-norm block,
    /root/reference/tactile_ssl/model/layers/block.py:89-114 + attention.py:54-76 + mlp.py:34-40 (tests/golden/block_stack.npz,
    written by make_golden.run_block_stack from the imported reference classes): outputs, input gradient and every parameter gradient."""
    z = _load(golden_dir, "block_stack")
    D, depth, heads, mlp = [int(v) for v in z["meta"]]
    P = {"t." + k[len("param/"):]: torch.tensor(z[k]).requires_grad_(True) for k in z.files if k.startswith("param/")}
    assert P["t.layers.0.1.net.1.weight"].shape == (mlp, D)
    x = torch.tensor(z[f"n{n}/x"]).requires_grad_(True)
    first = O.transformer(x, P, "t.", 1, heads, D // heads, final_norm=False)
    np.testing.assert_allclose(first.detach().numpy(), z[f"n{n}/block0_out"], rtol=1e-4, atol=1e-5)
    y = O.transformer(x, P, "t.", depth, heads, D // heads)
    np.testing.assert_allclose(y.detach().numpy(), z[f"n{n}/y"], rtol=1e-4, atol=1e-5)
    (y * torch.tensor(z[f"n{n}/cot"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), z[f"n{n}/dx"], rtol=1e-3, atol=1e-5)
    for k, p in P.items():
        ref = z[f"n{n}/grad/" + k[2:]]
        assert np.abs(p.grad.numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-6, k
