"""End-to-end parity of the HIP MAE step (through the C ABI, via the drop-in VTT / VTMAE modules) on a real MI355X:
  * against the golden vectors recorded from the reference's own VTMAE (tests/golden/*.npz),
  * against the CPU oracle on seeded inputs at other sizes,
  * size-independent properties at the BASELINE cfg-2 size.
Tolerances (BASELINE.json north_star): mask indices / gathers bit-exact; loss 1e-4 rel (fp32), 1e-2 rel (bf16)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from m3l_amd import VTMAE, VTT  # noqa: E402
from m3l_amd import _lib as L  # noqa: E402
from oracle import vtmae_oracle as O  # noqa: E402

CASES = ["vt_small", "v_only_small", "vt_decdim", "vt_cfg2_geom", "vt_earlyconv", "vt_learnedpos"]
DEV = "cuda:0"


def build_from_fixture(z, compute_dtype="fp32"):
    image_hw, tactile_hw, ip, tp, dim, depth, heads, mlp, C, k, dd, ddepth, dheads, B = [int(v) for v in z["meta"]]
    enc = VTT(image_size=image_hw, tactile_size=tactile_hw, image_patch_size=ip, tactile_patch_size=tp, dim=dim, depth=depth,
              heads=heads, mlp_dim=mlp, image_channels=C, tactile_channels=C, num_tactiles=k)
    mae = VTMAE(encoder=enc, decoder_dim=dd, masking_ratio=float(z["ratio"]), decoder_depth=ddepth, decoder_heads=dheads,
                num_tactiles=k, compute_dtype=compute_dtype, early_conv_masking=bool(int(z["early_conv"])),
                use_sincosmod_encodings=bool(int(z["sincos"])) if "sincos" in z.files else True)
    sd = {k_[len("param/"):]: torch.tensor(z[k_]) for k_ in z.files if k_.startswith("param/")}
    mae.load_state_dict(sd, strict=True)
    return mae.to(DEV)


def inputs_of(z):
    x = {k[len("input/"):]: torch.tensor(z[k]).to(DEV) for k in z.files if k.startswith("input/")}
    n = len([k for k in z.files if k.startswith("noise/")])
    return x, [torch.tensor(z[f"noise/{i}"]).to(DEV) for i in range(n)]


def ref_perm_rows_equal(z, masked, unmasked, cfg):
    """bit-exact index parity on tie-free rows; on tied rows (reference order unspecified) same sorted keys."""
    perms = [z[f"argsort/{i}"] for i in range(len([k for k in z.files if k.startswith("noise/")]))]
    noises = [z[f"noise/{i}"] for i in range(len(perms))]
    rm, ru, _, _ = O.mask_indices(noises, cfg.ratio, cfg.n_img, cfg.n_tac, cfg.num_tactiles, perms=perms)
    sm, su, _, _ = O.mask_indices(noises, cfg.ratio, cfg.n_img, cfg.n_tac, cfg.num_tactiles)
    assert np.array_equal(masked, sm) and np.array_equal(unmasked, su), "HIP mask indices != stable argsort contract"
    tie_free = np.ones(masked.shape[0], dtype=bool)
    for nz in noises:
        tie_free &= np.array([len(np.unique(r)) == len(r) for r in nz])
    assert np.array_equal(masked[tie_free], rm[tie_free]) and np.array_equal(unmasked[tie_free], ru[tie_free])
    return rm, ru


@pytest.mark.parametrize("name", CASES)
def test_golden_fp32(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = O.cfg_from_meta(z["meta"], z["ratio"])
    mae = build_from_fixture(z)
    x, noises = inputs_of(z)
    dump = {}
    loss = mae(x, mask_noise=noises, dump=dump)
    loss.backward()
    torch.cuda.synchronize()
    masked, unmasked = dump["masked_indices"].cpu().numpy(), dump["unmasked_indices"].cpu().numpy()
    assert masked.dtype == np.int64
    ref_perm_rows_equal(z, masked, unmasked, cfg)
    # loss: 1e-4 relative (north star, fp32)
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-4 * abs(float(z["loss"])), (float(loss.detach()), float(z["loss"]))
    # intermediates that do not depend on the (unspecified) order of tied keys inside the masked list
    np.testing.assert_allclose(dump["encoder_in"].cpu().numpy(), z["cap/encoder_in"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dump["encoder_out"].cpu().numpy(), z["cap/encoder_out"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dump["decoder_in"].cpu().numpy(), z["cap/decoder_in"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dump["decoder_out"].float().cpu().numpy(), z["cap/decoder_out"], rtol=1e-3, atol=2e-4)
    # gradients of every parameter the reference trains
    worst = 0.0
    for k in z.files:
        if not k.startswith("grad/"):
            continue
        p = dict(mae.named_parameters())[k[len("grad/"):]]
        assert p.grad is not None, k
        ref = z[k]
        err = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(1e-6, float(np.abs(ref).max()))
        worst = max(worst, err)
        assert err <= 2e-3, (k, err)
    for u in z["unused_params"]:
        assert dict(mae.named_parameters())[str(u)].grad is None, u


@pytest.mark.parametrize("name", ["vt_small", "vt_decdim", "vt_earlyconv", "vt_learnedpos"])
def test_golden_bf16_loss(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    mae = build_from_fixture(z, compute_dtype="bf16")
    x, noises = inputs_of(z)
    loss = mae(x, mask_noise=noises)
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-2 * abs(float(z["loss"])), (float(loss.detach()), float(z["loss"]))
    g = mae.to_pixels.weight.grad.cpu().numpy()
    ref = z["grad/to_pixels.weight"]
    assert np.abs(g - ref).max() <= 0.1 * np.abs(ref).max()
    cos = float((g * ref).sum() / (np.linalg.norm(g) * np.linalg.norm(ref)))
    assert cos > 0.99, cos


@pytest.mark.parametrize("name", ["vt_small", "vt_decdim", "vt_earlyconv", "vt_learnedpos"])
def test_golden_get_embeddings(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    mae = build_from_fixture(z)
    x, _ = inputs_of(z)
    with torch.no_grad():
        e = mae.get_embeddings(x, eval=True)
    np.testing.assert_allclose(e.cpu().numpy(), z["embeddings"], rtol=1e-3, atol=1e-4)


def _oracle_run(mae, cfg, x, noises):
    P = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in mae.state_dict().items()}
    r = O.vtmae_forward(P, cfg, {k: v.cpu() for k, v in x.items()}, [n.cpu() for n in noises])
    r["loss"].backward()
    return P, r


@pytest.mark.parametrize("dt,tol", [("fp32", 1e-4), ("bf16", 1e-2)])
def test_oracle_vit_tiny_shapes(dt, tol):
    """cfg-2 architecture (ViT-Tiny encoder 192/3 heads, decoder 192/3 heads, 64x64 + 2x32x32) at reduced depth and
    B=6 — the oracle finishes in seconds."""
    torch.manual_seed(42)
    enc = VTT(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=192, depth=3, heads=3, mlp_dim=768)
    mae = VTMAE(encoder=enc, decoder_dim=192, masking_ratio=0.75, decoder_depth=2, decoder_heads=3, compute_dtype=dt).to(DEV)
    B = 6
    g = torch.Generator(device="cpu").manual_seed(1)
    x = {"image": torch.rand(B, 3, 64, 64, generator=g).to(DEV), "tactile1": torch.rand(B, 3, 32, 32, generator=g).to(DEV),
         "tactile2": torch.rand(B, 3, 32, 32, generator=g).to(DEV)}
    noises = [torch.rand(B, 64, generator=g).to(DEV) for _ in range(3)]
    loss = mae(x, mask_noise=noises)
    loss.backward()
    cfg = O.OracleCfg(64, 32, 8, 4, 192, 3, 3, 768, 3, 2, 192, 2, 3, 0.75)
    P, r = _oracle_run(mae, cfg, x, noises)
    assert torch.equal(mae.last_mask[0].cpu(), r["masked_indices"]) and torch.equal(mae.last_mask[1].cpu(), r["unmasked_indices"])
    assert abs(float(loss.detach()) - float(r["loss"])) <= tol * abs(float(r["loss"])), (float(loss.detach()), float(r["loss"]))
    gtol = 1e-4 if dt == "fp32" else 0.15
    num = den = 0.0
    for name, p in mae.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert p.grad is None, name
            continue
        err = float((p.grad.cpu() - ref).abs().max()) / max(1e-6, float(ref.abs().max()))
        assert err <= gtol, (name, err)
        num += float((p.grad.cpu() - ref).double().square().sum())
        den += float(ref.double().square().sum())
    # the whole gradient as one vector: relative L2 distance to the fp32 oracle
    assert (num / den) ** 0.5 <= (1e-5 if dt == "fp32" else 8e-3), (num / den) ** 0.5


def test_vision_only_and_use_flags():
    """use_tactile=False on a vision+tactile model == the vision-only path of the reference (pretrain_models.py:163,208,226)."""
    torch.manual_seed(7)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2).to(DEV)
    B = 3
    x = {"image": torch.rand(B, 3, 32, 32, device=DEV), "tactile1": torch.rand(B, 3, 16, 16, device=DEV),
         "tactile2": torch.rand(B, 3, 16, 16, device=DEV)}
    noise = [torch.rand(B, 16, device=DEV)]
    loss = mae(x, use_tactile=False, mask_noise=noise)
    cfg = O.OracleCfg(32, 16, 8, 4, 64, 1, 2, 128, 3, 2, 64, 1, 2, 0.75)
    P = {k: v.detach().cpu() for k, v in mae.state_dict().items()}
    with torch.no_grad():
        r = O.vtmae_forward(P, cfg, {k: v.cpu() for k, v in x.items()}, [n.cpu() for n in noise], use_tactile=False)
    assert abs(float(loss.detach()) - float(r["loss"])) <= 1e-4 * abs(float(r["loss"]))


def test_cfg2_full_size_properties():
    """BASELINE cfg 2 exactly (ViT-Tiny 192/12/3, decoder 192/4/3, B=256, bf16): size-independent properties —
    index lists are permutations with the reference's per-modality counts, the gather is bit-exact w.r.t. the
    un-masked embedding, the loss is finite, deterministic, and invariant to permuting the batch."""
    torch.manual_seed(0)
    enc = VTT(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=192, depth=12, heads=3, mlp_dim=768)
    mae = VTMAE(encoder=enc, decoder_dim=192, masking_ratio=0.75, decoder_depth=4, decoder_heads=3, compute_dtype="bf16").to(DEV)
    B = 256
    x = {"image": torch.rand(B, 3, 64, 64, device=DEV), "tactile1": torch.rand(B, 3, 32, 32, device=DEV),
         "tactile2": torch.rand(B, 3, 32, 32, device=DEV)}
    noises = [torch.rand(B, 64, device=DEV) for _ in range(3)]
    dump = {}
    loss = mae(x, mask_noise=noises, dump=dump)
    loss.backward()
    masked, unmasked = dump["masked_indices"], dump["unmasked_indices"]
    assert masked.shape == (B, 144) and unmasked.shape == (B, 48)
    both = torch.cat([masked, unmasked], 1).sort(dim=1).values
    assert torch.equal(both, torch.arange(192, device=DEV).expand(B, -1))
    for lo, hi, a, b in [(0, 64, 0, 48), (64, 128, 48, 96), (128, 192, 96, 144)]:
        assert ((masked[:, a:b] >= lo) & (masked[:, a:b] < hi)).all()
    # sortedness of the keys along each modality's permutation (stable argsort contract)
    perm0 = torch.cat([masked[:, :48], unmasked[:, :16]], 1)
    keys = torch.gather(noises[0], 1, perm0)
    assert (keys[:, 1:] >= keys[:, :-1]).all()
    # bit-exact gather: visible tokens == rows of the full (un-masked) embedding at the unmasked indices
    with torch.no_grad():
        image, tactiles, geom, _ = mae._inputs(x, True, True)
        full = mae._tokens(geom, image, tactiles, None, 64, 192)
    vis = torch.gather(full, 1, unmasked[:, :, None].expand(-1, -1, 192))
    assert torch.equal(vis, dump["encoder_in"])
    assert torch.isfinite(loss)
    g1 = mae.to_pixels.weight.grad.clone()
    # determinism
    mae.zero_grad()
    loss2 = mae(x, mask_noise=noises)
    loss2.backward()
    assert float(loss2) == float(loss.detach()) and torch.equal(mae.to_pixels.weight.grad, g1)
    # batch-permutation invariance of the mean loss (samples are independent)
    perm = torch.randperm(B, device=DEV)
    loss3 = mae({k: v[perm] for k, v in x.items()}, mask_noise=[n[perm] for n in noises])
    assert abs(float(loss3) - float(loss.detach())) <= 1e-3 * abs(float(loss.detach()))
    assert all(torch.isfinite(p.grad).all() for p in mae.parameters() if p.grad is not None)


@pytest.mark.parametrize("sincos", [True, False])
def test_direct_grad_mode_matches_autograd_mode(sincos):
    """GradSync makes the kernels write parameter gradients straight into one flat buffer (no autograd accumulate):
    the values must be bit-identical to the gradients autograd receives in the default mode.  sincos=False: the learned position
    tables (pretrain_models.py:218-219,280-287) own the trailing span of the buffer and receive their gradient through autograd."""
    from m3l_amd.parallel import GradSync
    torch.manual_seed(3)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=128, depth=2, heads=2, mlp_dim=256)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, compute_dtype="bf16",
                use_sincosmod_encodings=sincos).to(DEV)
    B = 5
    x = {"image": torch.rand(B, 3, 32, 32, device=DEV), "tactile1": torch.rand(B, 3, 16, 16, device=DEV),
         "tactile2": torch.rand(B, 3, 16, 16, device=DEV)}
    noises = [torch.rand(B, 16, device=DEV) for _ in range(3)]
    mae(x, mask_noise=noises).backward()
    ref = {n: p.grad.clone() for n, p in mae.named_parameters() if p.grad is not None}
    mae.zero_grad(set_to_none=True)
    sync = GradSync(mae)
    if sincos:
        sync.flat.fill_(float("nan"))      # every slot must be overwritten by its producer
    else:
        sync.zero_grad()                   # (autograd ADDS the position-table gradients; the unused modality tables keep their zeros)
        assert "encoder.pos_embedding" in ref and "decoder_pos_emb.weight" in ref
    mae(x, mask_noise=noises).backward()
    # without communication the tails of the transformer backwards are still un-joined on the side stream (recorded from the autograd
    # worker thread) and finish(), called from this thread, joins them
    from m3l_amd import _lib as L_
    assert L_.lib().m3l_side_pending() > 0 and len(sync._keep) > 0
    sync.finish()
    assert L_.lib().m3l_side_pending() == 0 and not sync._keep
    assert torch.isfinite(sync.flat).all()
    for n, p in mae.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), n
        else:
            assert p.grad is None or (not sincos and not p.grad.any()), n


@pytest.mark.parametrize("with_sync", [False, True])
@pytest.mark.parametrize("arch", ["tiny_bf16", "decdim_bf16", "vision_only_fp32", "tiny_fp32", "earlyconv_bf16", "earlyconv_fp32", "learnedpos_fp32",
                                  "earlyconv_learnedpos_bf16"])
def test_fused_step_is_bit_identical_to_module_chain(arch, with_sync):
    """csrc/mae_step.hip: the whole step as two library calls (one autograd node) against the five per-module autograd Functions —
    same kernels in the same order, so loss, mask indices and every parameter gradient must be bit-identical; with a GradSync (the
    kernels write the flat buffer, autograd sees one anchor input) and without (autograd receives every gradient).  Round 4: the
    EarlyCNN front end with its all-patch loss (early_conv_masking=True, the reference's default, train.py:62) and learned position
    tables (use_sincosmod_encodings=False: their gradients reach the parameters through autograd in both modes)."""
    from m3l_amd import functional as Fn
    from m3l_amd.parallel import GradSync
    kw = dict(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=128, depth=3, heads=2, mlp_dim=256)
    mkw = dict(decoder_dim=128, masking_ratio=0.75, decoder_depth=2, decoder_heads=2)
    if arch == "decdim_bf16":
        mkw.update(decoder_dim=64, decoder_heads=1)          # enc_to_dec Linear + truncated decoder sincos
    if arch == "vision_only_fp32":
        kw.update(num_tactiles=0)
        mkw.update(num_tactiles=0)
    if "earlyconv" in arch:
        mkw.update(early_conv_masking=True)
    if "learnedpos" in arch:
        mkw.update(use_sincosmod_encodings=False)
    dt = "fp32" if arch.endswith("fp32") else "bf16"
    B = 6
    g = torch.Generator(device="cpu").manual_seed(11)
    x = {"image": torch.rand(B, 3, 32, 32, generator=g).to(DEV)}
    nt = kw.get("num_tactiles", 2)
    for i in range(nt):
        x[f"tactile{i + 1}"] = torch.rand(B, 3, 16, 16, generator=g).to(DEV)
    noises = [torch.rand(B, 16, generator=g).to(DEV) for _ in range(1 + nt)]

    def run(fused):
        torch.manual_seed(2)
        mae = VTMAE(encoder=VTT(**kw), compute_dtype=dt, **mkw).to(DEV)
        sync = GradSync(mae) if with_sync else None
        if sync is not None:
            sync.zero_grad()
        keep = Fn.FUSED_STEP
        Fn.FUSED_STEP = fused
        try:
            loss = mae(x, mask_noise=noises)
            assert (type(loss.grad_fn).__name__ == "MaeStepFnBackward") == fused
            (loss * 1.5).backward()
        finally:
            Fn.FUSED_STEP = keep
        if sync is not None:
            sync.finish()
        torch.cuda.synchronize()
        return loss.detach().clone(), mae.last_mask, {n: (None if p.grad is None else p.grad.clone()) for n, p in mae.named_parameters()}

    l0, m0, g0 = run(False)
    l1, m1, g1 = run(True)
    assert torch.equal(l0, l1) and torch.equal(m0[0], m1[0]) and torch.equal(m0[1], m1[1])
    pos_tables = ("encoder.pos_embedding", "decoder_pos_emb.weight")      # batch sums: torch.sum in the module chain, the library's fixed-order
    for n in g0:                                                          # reduction in the fused step -> equal to fp32 round-off, not bit for bit
        assert (g0[n] is None) == (g1[n] is None), n
        if g0[n] is not None and n in pos_tables:
            assert float((g0[n] - g1[n]).abs().max()) <= 1e-5 * float(g0[n].abs().max()) + 1e-9, n
        elif g0[n] is not None:
            assert torch.equal(g0[n], g1[n]), n
    if "learnedpos" in arch:
        assert g1["encoder.pos_embedding"] is not None and g1["decoder_pos_emb.weight"] is not None
        assert float(g1["encoder.pos_embedding"][0, 1:].abs().max()) > 0 and float(g1["decoder_pos_emb.weight"].abs().max()) > 0
    # use_vision / use_tactile flags go through the fused step too
    if nt:
        torch.manual_seed(2)
        mae = VTMAE(encoder=VTT(**kw), compute_dtype=dt, **mkw).to(DEV)
        res = []
        for fused in (False, True):
            Fn.FUSED_STEP = fused
            try:
                mae.zero_grad(set_to_none=True)
                loss = mae(x, use_vision=False, mask_noise=noises[1:])
                loss.backward()
                res.append((loss.detach().clone(), {n: (None if p.grad is None else p.grad.clone()) for n, p in mae.named_parameters()}))
            finally:
                Fn.FUSED_STEP = True
        assert torch.equal(res[0][0], res[1][0])
        for n in res[0][1]:
            a, b = res[0][1][n], res[1][1][n]
            assert (a is None) == (b is None), n
            if a is not None and n in pos_tables:
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, n
            else:
                assert a is None or torch.equal(a, b), n


@pytest.mark.parametrize("t192_mode", [1, 7])
def test_drop_h_is_bit_identical(t192_mode):
    """m3l_set_drop_h(1): the fused feed-forward kernels do not write h = GELU(u); the fc2 weight-gradient kernel stages u and applies the
    GELU of the kernel that produced it (erf form for the per-sample blocks of the encoder, the fitted form for the row tiles) in its LDS
    stage.  Loss and EVERY gradient bit-identical to the run that saved h — cfg-2 token geometry (encoder n = 48 on the block kernels,
    decoder n = 192 on 48-row tiles, or on 192-row tiles + per-sample attention when forced), also with a narrow MLP (h is then the
    NARROW operand of the weight-gradient tile)."""
    lib = L.lib()
    for mlp in (768, 128):
        torch.manual_seed(6)
        enc = VTT(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=192, depth=2, heads=3, mlp_dim=mlp)
        mae = VTMAE(encoder=enc, decoder_dim=192, masking_ratio=0.75, decoder_depth=2, decoder_heads=3, compute_dtype="bf16").to(DEV)
        if mlp == 128:
            for lyr in mae.decoder.layers:          # VTMAE builds the decoder with mlp = 4 * dim: the narrow case goes through the encoder only
                pass
        B = 5
        g = torch.Generator(device="cpu").manual_seed(13)
        x = {"image": torch.rand(B, 3, 64, 64, generator=g).to(DEV), "tactile1": torch.rand(B, 3, 32, 32, generator=g).to(DEV),
             "tactile2": torch.rand(B, 3, 32, 32, generator=g).to(DEV)}
        noises = [torch.rand(B, 64, generator=g).to(DEV) for _ in range(3)]
        res = []
        old_t = lib.m3l_set_t192(t192_mode)
        try:
            for drop in (0, 1):
                old = lib.m3l_set_drop_h(drop)
                try:
                    mae.zero_grad(set_to_none=True)
                    loss = mae(x, mask_noise=noises)
                    loss.backward()
                    torch.cuda.synchronize()
                finally:
                    lib.m3l_set_drop_h(old)
                res.append((loss.detach().clone(), {n: p.grad.clone() for n, p in mae.named_parameters() if p.grad is not None}))
        finally:
            lib.m3l_set_t192(old_t)
        assert torch.equal(res[0][0], res[1][0])
        assert res[0][1].keys() == res[1][1].keys()
        for n in res[0][1]:
            assert torch.equal(res[0][1][n], res[1][1][n]), (mlp, n)


def test_fused_step_with_frozen_parameters():
    """Parameters with requires_grad=False (a frozen decoder, a frozen patch projection) inside the fused step: the kernels still get a
    scratch slot for every gradient they write, the frozen parameters end up without .grad, the others match the per-module path."""
    from m3l_amd import functional as Fn
    torch.manual_seed(4)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=128, depth=2, heads=2, mlp_dim=256)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=2, decoder_heads=1, compute_dtype="bf16").to(DEV)
    frozen = [p for n, p in mae.named_parameters() if n.startswith("decoder.layers.0.") or n == "encoder.image_to_patch_embedding.2.weight"
              or n == "enc_to_dec.weight"]
    assert len(frozen) >= 10
    for p in frozen:
        p.requires_grad_(False)
    B = 4
    g = torch.Generator(device="cpu").manual_seed(12)
    x = {"image": torch.rand(B, 3, 32, 32, generator=g).to(DEV), "tactile1": torch.rand(B, 3, 16, 16, generator=g).to(DEV),
         "tactile2": torch.rand(B, 3, 16, 16, generator=g).to(DEV)}
    noises = [torch.rand(B, 16, generator=g).to(DEV) for _ in range(3)]
    res = []
    for fused in (False, True):
        Fn.FUSED_STEP = fused
        try:
            mae.zero_grad(set_to_none=True)
            loss = mae(x, mask_noise=noises)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), {n: (None if p.grad is None else p.grad.clone()) for n, p in mae.named_parameters()}))
        finally:
            Fn.FUSED_STEP = True
    assert torch.equal(res[0][0], res[1][0])
    for n, p in mae.named_parameters():
        a, b = res[0][1][n], res[1][1][n]
        if not p.requires_grad:
            assert a is None and b is None, n
        else:
            assert (a is None) == (b is None), n          # (the learned position tables are unused under sincos encodings: no gradient)
            assert a is None or torch.equal(a, b), n
    assert res[1][1]["decoder.layers.1.0.to_qkv.weight"] is not None and res[1][1]["encoder.image_to_patch_embedding.2.bias"] is not None


@pytest.mark.parametrize("fixture", ["vtt_dino_small", "vtt_dino_reg"])
def test_dino_style_vtt_matches_reference_fixture(golden_dir, fixture):
    """models/VTT.py forward / forward_features (with and without keep-index masks; without and with 4 register tokens, :166-172,
    305-312) vs the reference's own outputs."""
    from m3l_amd import DinoVTT
    z = np.load(os.path.join(golden_dir, fixture + ".npz"))
    hw, p, D, depth, heads, mlp, B = [int(v) for v in z["meta"]]
    nreg = int(z["num_register_tokens"]) if "num_register_tokens" in z.files else 0
    enc = DinoVTT(image_size=hw, tactile_size=hw, image_patch_size=p, tactile_patch_size=p, dim=D, depth=depth, heads=heads,
                  mlp_dim=mlp, num_tactiles=2, num_register_tokens=nreg)
    enc.load_state_dict({k[len("param/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    enc = enc.to(DEV).eval()
    x = {k[len("input/"):]: torch.tensor(z[k]).to(DEV) for k in z.files if k.startswith("input/")}
    masks = [torch.tensor(z["mask/0"]).to(DEV), torch.tensor(z["mask/1"]).to(DEV)]
    with torch.no_grad():
        full = enc.forward_features(x)
        mk = enc(x, masks)
    assert set(full.keys()) == {"x_norm_regtokens", "x_norm_patchtokens", "x_prenorm", "masks"}
    np.testing.assert_allclose(full["x_prenorm"].cpu().numpy(), z["full/x_prenorm"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(full["x_norm_patchtokens"].cpu().numpy(), z["full/x_norm_patchtokens"], rtol=1e-3, atol=1e-4)
    if nreg:
        assert full["x_norm_regtokens"].shape[1] == nreg
        np.testing.assert_allclose(full["x_norm_regtokens"].cpu().numpy(), z["full/x_norm_regtokens"], rtol=1e-3, atol=1e-4)
    assert mk.shape == z["masked/x_norm_patchtokens"].shape
    np.testing.assert_allclose(mk.cpu().numpy(), z["masked/x_norm_patchtokens"], rtol=1e-3, atol=1e-4)
    # gradients flow to every embed / transformer / norm parameter and match the oracle
    enc.train()
    out = enc(x, masks)
    out.square().mean().backward()
    P = O.load_fixture_params(z, requires_grad=True)
    r = O.vtt_dino_forward(P, image_patch=p, tactile_patch=p, depth=depth, heads=heads, x={k: v.cpu() for k, v in x.items()},
                           masks=[m.cpu() for m in masks])
    r["x_norm_patchtokens"].square().mean().backward()
    for name, prm in enc.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert prm.grad is None, name
            continue
        err = float((prm.grad.cpu() - ref).abs().max()) / max(1e-7, float(ref.abs().max()))
        assert err <= 5e-3, (name, err)


def test_extractor_style_consumer_with_grad():
    """MAEExtractor.forward (pretrain_models.py:819-841) = get_embeddings(eval=False) -> a 1-layer
    Transformer -> mean over tokens, with gradients into the encoder — vs the oracle + torch on CPU."""
    from m3l_amd import Transformer
    torch.manual_seed(5)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128)
    mae = VTMAE(encoder=enc, decoder_dim=64, decoder_depth=1, decoder_heads=2).to(DEV)
    head = Transformer(64, 1, 4, 64, 128).to(DEV)
    B = 3
    x = {"image": torch.rand(B, 3, 32, 32, device=DEV), "tactile1": torch.rand(B, 3, 16, 16, device=DEV),
         "tactile2": torch.rand(B, 3, 16, 16, device=DEV)}
    feat = head(mae.get_embeddings(x, eval=False)).mean(dim=1)
    assert feat.shape == (B, 64) and mae.training
    feat.square().sum().backward()
    cfg = O.OracleCfg(32, 16, 8, 4, 64, 2, 2, 128, 3, 2, 64, 1, 2, 0.75)
    P = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in mae.state_dict().items()}
    PH = {"t." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in head.state_dict().items()}
    emb = O.get_embeddings(P, cfg, {k: v.cpu() for k, v in x.items()})
    ref = O.transformer(emb, PH, "t.", 1, 4, 64).mean(dim=1)
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-3, atol=1e-4)
    ref.square().sum().backward()
    for name in ("encoder.transformer.layers.0.0.to_qkv.weight", "encoder.image_to_patch_embedding.2.weight", "encoder_modality_embedding.weight"):
        g, gr = dict(mae.named_parameters())[name].grad.cpu(), P[name].grad
        assert float((g - gr).abs().max()) <= 5e-3 * float(gr.abs().max()) + 1e-8, name
    gh = head.layers[0][1].net[1].weight.grad.cpu()
    assert float((gh - PH["t.layers.0.1.net.1.weight"].grad).abs().max()) <= 5e-3 * float(gh.abs().max()) + 1e-8


def _parity_vs_oracle(enc_kw, mae_kw, B, C, hw_img, hw_tac, k, cfg, tol=1e-4, gtol=1e-4, seed=0, l2tol=None):
    torch.manual_seed(seed)
    enc = VTT(**enc_kw)
    mae = VTMAE(encoder=enc, **mae_kw).to(DEV)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    x = {"image": torch.rand(B, C, hw_img, hw_img, generator=g).to(DEV)}
    for i in range(k):
        x[f"tactile{i + 1}"] = torch.rand(B, C, hw_tac, hw_tac, generator=g).to(DEV)
    noises = [torch.rand(B, cfg.n_img, generator=g).to(DEV)] + [torch.rand(B, cfg.n_tac, generator=g).to(DEV) for _ in range(k)]
    loss = mae(x, mask_noise=noises)
    loss.backward()
    P, r = _oracle_run(mae, cfg, x, noises)
    assert torch.equal(mae.last_mask[0].cpu(), r["masked_indices"]) and torch.equal(mae.last_mask[1].cpu(), r["unmasked_indices"])
    assert abs(float(loss.detach()) - float(r["loss"])) <= tol * abs(float(r["loss"])), (float(loss.detach()), float(r["loss"]))
    worst = ("", 0.0)
    num = den = 0.0
    for name, p in mae.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert p.grad is None, name
            continue
        err = float((p.grad.cpu() - ref).abs().max()) / max(1e-7, float(ref.abs().max()))
        if err > worst[1]:
            worst = (name, err)
        num += float((p.grad.cpu() - ref).double().square().sum())
        den += float(ref.double().square().sum())
    l2 = (num / max(den, 1e-300)) ** 0.5                    # the whole gradient as one vector: relative L2 distance to the fp32 oracle
    print(f"\n[parity] {mae_kw.get('compute_dtype', 'fp32')} dim {enc_kw['dim']} B {B}: loss rel "
          f"{abs(float(loss.detach()) - float(r['loss'])) / abs(float(r['loss'])):.2e}, worst grad {worst[0]} {worst[1]:.2e}, grad rel-L2 {l2:.2e}")
    assert worst[1] <= gtol, worst
    if l2tol is None:
        l2tol = 1e-5 if mae_kw.get("compute_dtype", "fp32") == "fp32" else 8e-3        # measured 2e-7 / 2.5e-3
    assert l2 <= l2tol, l2
    return mae


# per-parameter bound in bf16: 0.15 at B = 3 (the smallest gradients — a stem's conv3 weight — carry 9e-2 of relative noise at three samples),
# 0.05 at B = 40 (measured 1.7e-2); the whole-gradient rel-L2 is held to 8e-3 at both (measured 2.5e-3)
@pytest.mark.parametrize("dt,tol,gtol,B", [("fp32", 1e-4, 1e-4, 3), ("bf16", 1e-2, 0.15, 3), ("bf16", 1e-2, 0.05, 40)])
def test_reference_default_architecture(dt, tol, gtol, B):
    """M3L's own defaults at their real depth (train.py:58-67,128-153): dim 256 / depth 4 / heads 4 / mlp 512, decoder 256 / depth 3 /
    4 heads / mlp 1024, mask 0.95, early_conv_masking=True, frame_stack 4 -> 12 channels (SURVEY section 8 'ref' row); bf16 also takes
    the 4 x 4 im2col kernel of the EarlyCNN stem, and at B = 40 the decoder is M = 7680 rows (40 full row tiles)."""
    cfg = O.OracleCfg(64, 32, 8, 4, 256, 4, 4, 512, 12, 2, 256, 3, 4, 0.95)
    _parity_vs_oracle(dict(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=256, depth=4, heads=4, mlp_dim=512,
                           image_channels=12, tactile_channels=12, num_tactiles=2, frame_stack=4),
                      dict(decoder_dim=256, masking_ratio=0.95, decoder_depth=3, decoder_heads=4, num_tactiles=2, early_conv_masking=True,
                           frame_stack=4, compute_dtype=dt), B=B, C=12, hw_img=64, hw_tac=32, k=2, cfg=cfg, tol=tol, gtol=gtol)


def test_cfg4_shapes():
    """BASELINE cfg 4 geometry: 224x224 RGB (P16, 196 patches, pd 768) + 4 tactile 64x64 (P8), ViT-Small encoder 384 / 6 heads /
    mlp 1536, decoder 192 / 3 heads (enc_to_dec projection, truncated decoder sincos), 452 tokens -> 113 visible; reduced depth."""
    cfg = O.OracleCfg(224, 64, 16, 8, 384, 2, 6, 1536, 3, 4, 192, 1, 3, 0.75)
    mae = _parity_vs_oracle(dict(image_size=224, tactile_size=64, image_patch_size=16, tactile_patch_size=8, dim=384, depth=2, heads=6,
                                 mlp_dim=1536, num_tactiles=4),
                            dict(decoder_dim=192, masking_ratio=0.75, decoder_depth=1, decoder_heads=3, num_tactiles=4),
                            B=2, C=3, hw_img=224, hw_tac=64, k=4, cfg=cfg)
    assert mae.last_mask[0].shape == (2, 339) and mae.last_mask[1].shape == (2, 113)


def test_cfg5_shapes_patch14():
    """BASELINE cfg 5 MAE geometry (train_dino_cat_mae.py:76,78,140-147): 70x70 frames, P = 14 (patch dim 588, not a multiple of
    8 -> zero-padded K), 25 patches per modality, mask 0.8 -> 15 visible, 384 / 4 heads / mlp 768, decoder 384 / 4 heads."""
    cfg = O.OracleCfg(70, 70, 14, 14, 384, 2, 4, 768, 3, 2, 384, 1, 4, 0.8)
    mae = _parity_vs_oracle(dict(image_size=70, tactile_size=70, image_patch_size=14, tactile_patch_size=14, dim=384, depth=2, heads=4,
                                 mlp_dim=768, num_tactiles=2),
                            dict(decoder_dim=384, masking_ratio=0.8, decoder_depth=1, decoder_heads=4, num_tactiles=2),
                            B=3, C=3, hw_img=70, hw_tac=70, k=2, cfg=cfg)
    assert mae.last_mask[1].shape == (3, 15)


def test_cfg5_frame_stack4_patch_dim_2352():
    """cfg 5 at the reference's default --frame_stack 4 (train_dino_cat_mae.py:69,149-151): 12-channel 14x14 patches, patch dim 2352
    (> 1024: the wide patch-LayerNorm kernels), loss and every gradient against the oracle."""
    cfg = O.OracleCfg(70, 70, 14, 14, 128, 1, 2, 256, 12, 2, 128, 1, 2, 0.8)
    _parity_vs_oracle(dict(image_size=70, tactile_size=70, image_patch_size=14, tactile_patch_size=14, dim=128, depth=1, heads=2,
                           mlp_dim=256, image_channels=12, tactile_channels=12, num_tactiles=2, frame_stack=4),
                      dict(decoder_dim=128, masking_ratio=0.8, decoder_depth=1, decoder_heads=2, num_tactiles=2, frame_stack=4),
                      B=2, C=12, hw_img=70, hw_tac=70, k=2, cfg=cfg)


def test_odd_patch_dim_takes_the_element_wise_walk():
    """7x7 RGB patches (patch dim 147, not a multiple of 4; tactile 5x5 = 75): the patch kernels' 16-byte walk of the patch vector does
    not apply and the per-element form runs — same bounds against the oracle (pretrain_models.py:196-205,327-340)."""
    cfg = O.OracleCfg(56, 30, 7, 5, 128, 1, 2, 256, 3, 2, 128, 1, 2, 0.75)
    _parity_vs_oracle(dict(image_size=56, tactile_size=30, image_patch_size=7, tactile_patch_size=5, dim=128, depth=1, heads=2,
                           mlp_dim=256, num_tactiles=2),
                      dict(decoder_dim=128, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2),
                      B=3, C=3, hw_img=56, hw_tac=30, k=2, cfg=cfg)


def test_flat_adam_matches_torch_adam():
    """FlatAdam (one HIP launch over the flat parameter buffer) == torch.optim.Adam step for step (reference optimizer,
    ppo_mae.py:182-183), including weight decay."""
    from m3l_amd.parallel import FlatAdam, GradSync
    torch.manual_seed(9)

    def make():
        torch.manual_seed(9)
        enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128)
        return VTMAE(encoder=enc, decoder_dim=64, decoder_depth=1, decoder_heads=2).to(DEV)
    a, b = make(), make()
    sync = GradSync(a)
    oa = FlatAdam(sync, lr=3e-3, weight_decay=0.01)
    names = [n for n, p in b.named_parameters() if n not in GradSync.SKIP]
    ob = torch.optim.Adam([dict(b.named_parameters())[n] for n in names], lr=3e-3, weight_decay=0.01)
    g = torch.Generator(device="cpu").manual_seed(1)
    for _ in range(4):
        grads = {n: torch.randn(dict(b.named_parameters())[n].shape, generator=g).to(DEV) for n in names}
        for n in names:
            dict(a.named_parameters())[n].grad.copy_(grads[n])
            dict(b.named_parameters())[n].grad = grads[n].clone()
        oa.step()
        ob.step()
    for n in names:
        pa, pb = dict(a.named_parameters())[n], dict(b.named_parameters())[n]
        assert float((pa - pb).abs().max()) <= 2e-6 * max(1.0, float(pb.abs().max())), n


@pytest.mark.parametrize("chunk", [1, 2])
def test_chunked_transformer_backward_is_bit_identical(chunk):
    """m3l_transformer_bwd_range: splitting the backward into layer chunks (used to overlap the RCCL all-reduce of finished layers
    with the remaining backward) must not change a single bit of any gradient."""
    from m3l_amd import functional as Fn
    torch.manual_seed(21)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=3, heads=2, mlp_dim=128)
    mae = VTMAE(encoder=enc, decoder_dim=64, decoder_depth=2, decoder_heads=2, compute_dtype="bf16").to(DEV)
    B = 4
    x = {"image": torch.rand(B, 3, 32, 32, device=DEV), "tactile1": torch.rand(B, 3, 16, 16, device=DEV),
         "tactile2": torch.rand(B, 3, 16, 16, device=DEV)}
    noises = [torch.rand(B, 16, device=DEV) for _ in range(3)]
    mae(x, mask_noise=noises).backward()
    ref = {n: p.grad.clone() for n, p in mae.named_parameters() if p.grad is not None}
    mae.zero_grad(set_to_none=True)
    try:
        Fn.BWD_CHUNK_LAYERS = chunk
        mae(x, mask_noise=noises).backward()
    finally:
        Fn.BWD_CHUNK_LAYERS = None
    for n, p in mae.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), n


def test_chunked_backward_remainder_groups_fit_the_workspace():
    """A weight-gradient group may hold fewer layers than the stack's layers-per-launch (a chunked backward's range ends, the last group of a
    stack).  Fewer tiles mean more splits per tile, and at cfg 5's decoder shape (384 / 3 / 4 / 1536, M = 128 x 75) a 1-layer group needs MORE
    slab workspace than the multi-layer group the workspace used to be sized for ("wgrad: workspace too small").  Chunks of one layer against
    the unchunked backward: no error, same gradients (different split plans: compared to 2e-3 of the largest entry)."""
    from m3l_amd import functional as Fn
    from m3l_amd.pretrain_models import Transformer
    torch.manual_seed(23)
    tf = Transformer(dim=384, depth=3, heads=4, dim_head=64, mlp_dim=1536).to(DEV)
    tf.compute_dtype = "bf16"
    x = torch.randn(128, 75, 384, device=DEV)
    dy = torch.randn(128, 75, 384, device=DEV)
    (tf(x) * dy).sum().backward()
    ref = {n: p.grad.clone() for n, p in tf.named_parameters()}
    tf.zero_grad(set_to_none=True)
    try:
        Fn.BWD_CHUNK_LAYERS = 1
        (tf(x) * dy).sum().backward()
    finally:
        Fn.BWD_CHUNK_LAYERS = None
    for n, p in tf.named_parameters():
        scale = float(ref[n].abs().max()) + 1e-12
        assert float((p.grad - ref[n]).abs().max()) <= 2e-3 * scale, n


def test_fused_gemm_layernorm_path_matches_separate_kernels():
    """gemm_rowln.hip (LayerNorm forward / backward inside the GEMM epilogue; experimental, off by default) must reproduce the
    separate-kernel path: same loss to 1e-5 and every gradient to 2e-3 (fp32), and the bf16 loss within 1e-2 of the oracle."""
    from m3l_amd import _lib as L
    torch.manual_seed(31)
    enc = VTT(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=192, depth=2, heads=3, mlp_dim=768)
    mae = VTMAE(encoder=enc, decoder_dim=192, decoder_depth=2, decoder_heads=3).to(DEV)
    B = 5
    x = {"image": torch.rand(B, 3, 64, 64, device=DEV), "tactile1": torch.rand(B, 3, 32, 32, device=DEV),
         "tactile2": torch.rand(B, 3, 32, 32, device=DEV)}
    noises = [torch.rand(B, 64, device=DEV) for _ in range(3)]
    loss0 = mae(x, mask_noise=noises)
    loss0.backward()
    ref = {n: p.grad.clone() for n, p in mae.named_parameters() if p.grad is not None}
    mae.zero_grad(set_to_none=True)
    old = L.lib().m3l_set_rowln(1)
    try:
        loss1 = mae(x, mask_noise=noises)
        loss1.backward()
        assert abs(float(loss1.detach()) - float(loss0.detach())) <= 1e-5 * abs(float(loss0.detach()))
        for n, p in mae.named_parameters():
            if n in ref:
                err = float((p.grad - ref[n]).abs().max()) / max(1e-7, float(ref[n].abs().max()))
                assert err <= 2e-3, (n, err)
        mae.set_compute_dtype("bf16")
        lb = mae(x, mask_noise=noises)
        assert abs(float(lb.detach()) - float(loss0.detach())) <= 1e-2 * abs(float(loss0.detach()))
    finally:
        L.lib().m3l_set_rowln(old)


@pytest.mark.parametrize("name", ["recon_small", "recon_default_ratio", "recon_earlyconv"])
def test_reconstruct_golden(golden_dir, name):
    """VTMAE.reconstruct (pretrain_models.py:344-586) against the reference's own outputs: masked frames exact,
    reconstructed frames / MSEs to fp32 tolerance, and the module's masking_ratio untouched afterwards."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    mae = build_from_fixture(z).eval()
    x, noises = inputs_of(z)
    mr = float(z["mask_ratio"])
    keep = mae.masking_ratio
    r = mae.reconstruct(x, mask_ratio=None if mr < 0 else mr, use_tactile=bool(int(z["use_tactile"])), mask_noise=noises)
    assert mae.masking_ratio == keep
    want = sorted(k[4:] for k in z.files if k.startswith("out/"))
    assert sorted(r.keys()) == want
    for k in want:
        a, b = r[k].float().cpu().numpy(), z["out/" + k]
        assert a.shape == b.shape, k
        if k.endswith("_masked"):
            np.testing.assert_array_equal(a, b)
        elif k.startswith("recon_loss"):
            assert abs(float(a) - float(b)) <= 1e-4 * abs(float(b)), k
        else:
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4, err_msg=k)


def test_train_iterations_replay_buffer():
    """VTMAE.train_iterations (pretrain_models.py:679-715): too-small buffer is a no-op; otherwise `iterations` clipped
    AdamW steps that change the weights, lower the loss on the buffer, and leave the module in eval mode."""
    import random
    torch.manual_seed(0)
    random.seed(0)
    rng = np.random.default_rng(0)
    fs = 2
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2,
                frame_stack=fs).to(DEV)
    mae.initialize_training({"lr": 1e-3, "batch_size": 8})
    buf = [{"image": rng.random((fs, 32, 32, 3), dtype=np.float32), "tactile": rng.random((fs, 6, 16, 16), dtype=np.float32) * 2 - 1}
           for _ in range(16)]
    before = {k: v.clone() for k, v in mae.state_dict().items()}
    mae.train_iterations(3, buf[:4])                           # fewer samples than batch_size -> returns untouched
    assert all(torch.equal(v, before[k]) for k, v in mae.state_dict().items())

    from m3l_amd import vt_load
    obs = {"image": np.stack([b["image"] for b in buf]).transpose(0, 2, 3, 1, 4).reshape(16, 32, 32, -1),
           "tactile": np.stack([b["tactile"] for b in buf]).reshape(16, -1, 16, 16)}
    xb = vt_load(obs, frame_stack=fs, device=DEV)
    noises = [torch.rand(16, 16, device=DEV) for _ in range(3)]
    with torch.no_grad():
        l0 = float(mae(xb, mask_noise=noises))
    mae.train_iterations(30, buf)
    assert not mae.training
    assert any(not torch.equal(v, before[k]) for k, v in mae.state_dict().items())
    with torch.no_grad():
        l1 = float(mae(xb, mask_noise=noises))
    assert l1 < l0


@pytest.mark.parametrize("max_norm", [0.5, 1e6, None])
def test_fused_adamw_clip_matches_torch(max_norm):
    """FlatAdamW (m3l_adamw_step: one gradient-norm reduction + one update launch over the flat buffers) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW — the update of VTMAE.train_iterations (pretrain_models.py:675,710) — on the same
    model, inputs and mask noise over 4 steps: clipping active (0.5), inactive (1e6) and absent; the clipped gradients and the norm as
    clip_grad_norm_ leaves / returns them."""
    import copy
    from m3l_amd.parallel import FlatAdamW, GradSync
    torch.manual_seed(3)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128)
    ref = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2).to(DEV)
    fus = copy.deepcopy(ref)
    x = {"image": torch.rand(8, 3, 32, 32, device=DEV), "tactile1": torch.rand(8, 3, 16, 16, device=DEV), "tactile2": torch.rand(8, 3, 16, 16, device=DEV)}
    opt_r = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    sync = GradSync(fus)
    opt_f = FlatAdamW(sync, lr=1e-3, max_grad_norm=max_norm)
    for it in range(4):
        noise = [torch.rand(8, 16, device=DEV) for _ in range(3)]
        opt_r.zero_grad()
        (ref(x, mask_noise=noise) * 50.0).backward()              # x 50: the norm exceeds 0.5, the clip bites
        norm_r = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm) if max_norm is not None else None
        opt_r.step()
        opt_f.zero_grad()
        (fus(x, mask_noise=noise) * 50.0).backward()
        opt_f.step()
        if max_norm is not None:
            assert abs(float(opt_f.last_grad_norm) - float(norm_r)) <= 1e-5 * float(norm_r), it
        gr = dict(ref.named_parameters())
        for k, p_f in fus.named_parameters():
            if gr[k].grad is None:
                continue
            scale = float(gr[k].detach().abs().max()) + 1e-12
            assert float((p_f.detach() - gr[k].detach()).abs().max()) <= 2e-5 * scale + 2e-6, (it, k)
            if max_norm is not None and it == 0:
                gs = float(gr[k].grad.abs().max()) + 1e-12
                assert float((p_f.grad - gr[k].grad).abs().max()) <= 1e-4 * gs, ("grad", k)


def test_fused_adamw_leaves_gradless_parameters_alone():
    """torch.optim.AdamW skips parameters whose .grad is None (pretrain_models.py:675 builds it over self.parameters(); under
    train_iterations(no_tactile=True) the tactile patch embed and to_tactiles never receive a gradient): FlatAdamW must neither decay them
    nor give them moments, although its one launch runs over the whole flat buffer.  3 steps of the vision-only loss against
    clip_grad_norm_ + torch AdamW: every parameter equal, the gradient-less ones bit-identical to their initial values."""
    import copy
    from m3l_amd.parallel import FlatAdamW, GradSync
    torch.manual_seed(5)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128)
    ref = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2).to(DEV)
    fus = copy.deepcopy(ref)
    init = {k: v.detach().clone() for k, v in fus.named_parameters()}
    x = {"image": torch.rand(8, 3, 32, 32, device=DEV), "tactile1": torch.rand(8, 3, 16, 16, device=DEV), "tactile2": torch.rand(8, 3, 16, 16, device=DEV)}
    opt_r = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.1)
    sync = GradSync(fus)
    opt_f = FlatAdamW(sync, lr=1e-3, weight_decay=0.1, max_grad_norm=0.5)
    for it in range(3):
        noise = [torch.rand(8, 16, device=DEV)]
        opt_r.zero_grad()
        (ref(x, use_tactile=False, mask_noise=noise) * 50.0).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt_r.step()
        opt_f.zero_grad()
        (fus(x, use_tactile=False, mask_noise=noise) * 50.0).backward()
        opt_f.step()
    gr = dict(ref.named_parameters())
    idle = 0
    for k, p_f in fus.named_parameters():
        if id(p_f) not in sync._span:
            continue
        if gr[k].grad is None:
            idle += 1
            assert torch.equal(p_f.detach(), init[k]), k            # untouched: no decay, no update
            assert torch.equal(gr[k].detach(), init[k]), k
            a, b = sync._span[id(p_f)]
            assert float(opt_f.exp_avg[a:b].abs().max()) == 0.0 and float(opt_f.exp_avg_sq[a:b].abs().max()) == 0.0, k
        else:
            scale = float(gr[k].detach().abs().max()) + 1e-12
            assert float((p_f.detach() - gr[k].detach()).abs().max()) <= 2e-5 * scale + 2e-6, k
    assert idle >= 6, idle                                          # tactile patch embed (LN, Linear, LN) + to_tactiles at least


def test_train_iterations_fused_optimizer():
    """train_args['fused_optimizer']: train_iterations with FlatAdamW (clip fused into the update) lowers the loss like the torch path."""
    import random
    torch.manual_seed(0)
    random.seed(0)
    rng = np.random.default_rng(0)
    fs = 2
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2, frame_stack=fs).to(DEV)
    mae.initialize_training({"lr": 1e-3, "batch_size": 8, "fused_optimizer": True})
    buf = [{"image": rng.random((fs, 32, 32, 3), dtype=np.float32), "tactile": rng.random((fs, 6, 16, 16), dtype=np.float32) * 2 - 1}
           for _ in range(16)]
    from m3l_amd import vt_load
    obs = {"image": np.stack([b["image"] for b in buf]).transpose(0, 2, 3, 1, 4).reshape(16, 32, 32, -1),
           "tactile": np.stack([b["tactile"] for b in buf]).reshape(16, -1, 16, 16)}
    xb = vt_load(obs, frame_stack=fs, device=DEV)
    noises = [torch.rand(16, 16, device=DEV) for _ in range(3)]
    with torch.no_grad():
        l0 = float(mae(xb, mask_noise=noises))
    mae.train_iterations(30, buf)
    assert not mae.training
    with torch.no_grad():
        l1 = float(mae(xb, mask_noise=noises))
    assert l1 < l0


@pytest.mark.parametrize("comm", ["rccl", "c10d"])
def test_rccl_path_world1_matches_local(comm, monkeypatch):
    """The multi-GPU step (GradSync buckets + chunked transformer backward + RCCL all-reduce on its own stream + FlatAdam)
    rehearsed on ONE GPU: backend "nccl" (= RCCL) at world size 1 with force_comm, so every collective is really issued and
    fenced; parameters after 3 steps must equal the no-communication run bit for bit (SUM over one rank, x 1/1)."""
    import torch.distributed as dist
    from m3l_amd.parallel import FlatAdam, GradSync

    def run(force):
        torch.manual_seed(0)
        enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=6, heads=2, mlp_dim=128,
                  num_tactiles=2)
        mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=2, decoder_heads=2, num_tactiles=2,
                    compute_dtype="bf16").to(DEV)
        sync = GradSync(mae, force_comm=force)
        assert sync._comm == force
        if force:      # "rccl": the library's own RCCL communicator on its side stream; "c10d": torch.distributed's stream via the helper
            assert sync._direct == (comm == "rccl") and (sync._helper is None) == (comm == "rccl")
        opt = FlatAdam(sync, lr=1e-3)
        g = torch.Generator(device=DEV).manual_seed(5)
        x = {"image": torch.rand(8, 3, 32, 32, device=DEV, generator=g), "tactile1": torch.rand(8, 3, 16, 16, device=DEV, generator=g),
             "tactile2": torch.rand(8, 3, 16, 16, device=DEV, generator=g)}
        for _ in range(3):
            noises = [torch.rand(8, 16, device=DEV, generator=g) for _ in range(3)]
            sync.zero_grad()
            mae(x, mask_noise=noises).backward()
            sync.finish()
            opt.step()
        torch.cuda.synchronize()
        return sync.flat_params.clone(), sync.flat.clone()

    monkeypatch.setenv("M3L_COMM", comm)
    p0, g0 = run(False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        p1, g1 = run(True)
    finally:
        dist.destroy_process_group()
    assert torch.equal(g0, g1) and torch.equal(p0, p1)


# ---- cfg-5 fusion head: frozen DINOv2 encoder + concat extractor -------------------------------------------------
def _dino_from_fixture(z, compute_dtype):
    from m3l_amd import DinoV2Frozen
    dim, depth, heads, patch, img, reg = [int(v) for v in z["meta"]]
    m = DinoV2Frozen(embed_dim=dim, depth=depth, num_heads=heads, patch_size=patch, img_size=img, num_register_tokens=reg,
                     compute_dtype=compute_dtype)
    m.load_state_dict({k[len("param/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("compute_dtype,tol", [("fp32", 2e-4), ("bf16", 4e-2)])
def test_dinov2_frozen_golden(golden_dir, compute_dtype, tol):
    """m3l_frozen_vit_fwd against the independent implementation's outputs (tests/golden/dinov2_small.npz): interpolated
    positions + registers (exact data movement in fp32), every normed token, the CLS feature."""
    z = np.load(os.path.join(golden_dir, "dinov2_small.npz"))
    m = _dino_from_fixture(z, compute_dtype)
    x = torch.tensor(z["input/x"]).to(DEV)
    f = m.forward_features(x)
    cls = m(x)
    assert torch.equal(cls, f["x_norm_clstoken"]) and cls.shape == (2, 64) and not cls.requires_grad
    scale = float(np.abs(z["out/x_norm"]).max())
    np.testing.assert_allclose(f["tokens_in"].cpu().numpy(), z["out/tokens_in"], rtol=0, atol=(2e-2 if compute_dtype == "bf16" else 2e-5))
    y = torch.cat((f["x_norm_clstoken"][:, None], f["x_norm_regtokens"], f["x_norm_patchtokens"]), 1).cpu().numpy()
    assert np.abs(y - z["out/x_norm"]).max() <= tol * scale
    assert np.abs(cls.cpu().numpy() - z["out/cls"]).max() <= tol * scale


def test_dinov2_frozen_vits14_reg_vs_transformers_live():
    """Full dinov2_vits14_reg architecture (384 / 12 layers / 6 heads / 4 registers, position table for 518x518 interpolated to the
    5x5 grid of a 70x70 frame), random frozen weights: our HIP forward against transformers' CPU forward of the same weights."""
    transformers = pytest.importorskip("transformers")
    from m3l_amd import DinoV2Frozen
    from oracle.dinov2_oracle import hf_to_hub_state_dict
    torch.manual_seed(7)
    hf = transformers.Dinov2WithRegistersModel(transformers.Dinov2WithRegistersConfig(
        hidden_size=384, num_hidden_layers=12, num_attention_heads=6, patch_size=14, image_size=518, num_register_tokens=4)).eval()
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if "lambda" in n:
                p.copy_(0.5 + torch.rand(p.shape, generator=g))           # non-trivial LayerScale
            elif p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.rand(4, 3, 70, 70, generator=g)
    with torch.no_grad():
        ref = hf(x).pooler_output
    m = DinoV2Frozen()
    m.load_state_dict(hf_to_hub_state_dict(hf.state_dict()), strict=True)
    m = m.to(DEV)
    scale = ref.abs().max().item()
    for cd, tol in (("fp32", 5e-4), ("bf16", 5e-2)):
        m.set_compute_dtype(cd)
        out = m(x.to(DEV)).cpu()
        assert out.shape == (4, 384)
        assert (out - ref).abs().max().item() <= tol * scale, (cd, (out - ref).abs().max().item(), scale)
    # frozen-weight cache follows the parameters: an in-place change must be picked up
    with torch.no_grad():
        m.norm.bias.add_(1.0)
    assert (m(x.to(DEV)).cpu() - (ref + 0)).abs().max().item() > 0.5 * 1.0 - 5e-2 * scale


@pytest.mark.parametrize("D,depth", [(128, 2), (384, 4)])
def test_dino_cat_extractor_cfg5(D, depth):
    """cfg-5 geometry (70x70, P = 14, 2 tactile 70x70, frame_stack 4 = the reference default; at dim 384 / depth 4 / 4 heads / mlp 768 =
    train_dino_cat_mae.py:144-147 exactly, and once at dim 128 for the narrow kernels): DinoCatMAEExtractor against
    the same pipeline composed from the CPU oracles (get_embeddings -> 1-layer Transformer -> mean | DINOv2 CLS -> cat -> MLP), and
    gradients reach the MAE encoder, the extra transformer layer and the MLP but not the frozen DINOv2."""
    from m3l_amd import DinoCatMAEExtractor, DinoV2Frozen
    from oracle import dinov2_oracle as DO
    torch.manual_seed(3)
    fs, dh = 4, D // 64
    enc = VTT(image_size=70, tactile_size=70, image_patch_size=14, tactile_patch_size=14, dim=D, depth=depth, heads=4, mlp_dim=2 * D,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=D, masking_ratio=0.8, decoder_depth=1, decoder_heads=4, num_tactiles=2, frame_stack=fs).to(DEV)
    dino = DinoV2Frozen(embed_dim=D, depth=2, num_heads=dh, img_size=98, compute_dtype="fp32")
    with torch.no_grad():
        for p in dino.parameters():
            p.copy_(0.2 * torch.randn(p.shape))
        dino.norm.weight.fill_(1.0)
    dino = dino.to(DEV)
    ext = DinoCatMAEExtractor(dino, mae, D, vision_only_control=False, frame_stack=fs).to(DEV).eval()      # eval: Dropout off
    assert ext.features_dim == D
    rng = np.random.default_rng(0)
    B = 3
    obs = {"image": torch.tensor(rng.random((B, fs, 70, 70, 3), dtype=np.float32)).to(DEV),
           "tactile": torch.tensor(rng.random((B, fs, 6, 70, 70), dtype=np.float32) * 2 - 1).to(DEV)}
    out = ext(obs)
    assert out.shape == (B, D)
    out.square().mean().backward()
    assert mae.encoder.transformer.layers[0][0].to_qkv.weight.grad is not None
    assert ext.vit_layer.transformer.layers[0][1].net[1].weight.grad is not None and ext.mlp[0].weight.grad is not None
    assert all(p.grad is None for p in dino.parameters())

    # the same numbers from the CPU oracles
    x = O.vt_load({"image": obs["image"].cpu().permute(0, 2, 3, 1, 4).reshape(B, 70, 70, -1).numpy(),
                   "tactile": obs["tactile"].cpu().reshape(B, -1, 70, 70).numpy()}, frame_stack=fs)
    P = {k: v.detach().cpu() for k, v in mae.state_dict().items()}
    cfg = O.OracleCfg(70, 70, 14, 14, D, depth, 4, 2 * D, 3 * fs, 2, D, 1, 4, 0.8)
    with torch.no_grad():
        tok = O.get_embeddings(P, cfg, x)
        PL = {k: v.detach().cpu() for k, v in ext.vit_layer.state_dict().items()}
        pooled = O.transformer(tok, PL, "transformer.", 1, 4, 64).mean(1)
        PD = {k: v.detach().cpu() for k, v in dino.state_dict().items()}
        cls = DO.dinov2_forward(PD, x["image"][:, 3:6], patch=14, depth=2, heads=dh)["cls"]   # fs = 4: mid = 2 -> channels 3..5
        ref = ext.mlp.cpu()(torch.cat((pooled, cls), -1))
    ext.mlp.to(DEV)
    assert (out.detach().cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-5


_RANDOM_GEOMS = [
    # (image, ip, tactile, tp, C, k, D, depth, heads, mlp, dd, ddepth, dheads, ratio, B)
    (32, 8, 16, 4, 3, 1, 64, 1, 1, 72, 64, 1, 1, 0.5, 1),          # one sensor, heads = 1 (identity to_out), mlp not a power of two, B = 1
    (48, 16, 24, 8, 3, 3, 128, 2, 2, 200, 64, 1, 2, 0.9, 2),       # three sensors, dd < D (truncated decoder sincos), 9 + 3*9 tokens
    (40, 8, 32, 4, 6, 2, 192, 1, 3, 384, 128, 2, 1, 0.75, 5),      # 25 image + 2*64 tactile tokens, 6 channels, odd batch
    (28, 14, 28, 14, 3, 2, 64, 2, 2, 128, 64, 1, 2, 0.6, 3),       # 14x14 patches (patch dim 588, K padded), 4 + 2*4 tokens
    (64, 8, 16, 4, 3, 0, 256, 1, 4, 512, 256, 1, 4, 0.75, 2),      # vision only, D = 256
    (32, 8, 32, 4, 3, 4, 64, 1, 2, 64, 64, 2, 2, 0.8, 2),          # four sensors (M3L_MAX_SENSORS)
]


@pytest.mark.parametrize("geom", _RANDOM_GEOMS, ids=[f"g{i}" for i in range(len(_RANDOM_GEOMS))])
def test_assorted_geometries_vs_oracle(geom):
    """Loss, mask indices and every gradient against the CPU oracle on geometries away from the named configs: B = 1, odd batches,
    0..4 sensors, token counts that are not multiples of the 16-row MFMA tile, identity to_out, decoder narrower than the encoder."""
    img, ip, tac, tp, C, k, D, depth, heads, mlp, dd, ddepth, dheads, ratio, B = geom
    cfg = O.OracleCfg(img, tac, ip, tp, D, depth, heads, mlp, C, k, dd, ddepth, dheads, ratio)
    _parity_vs_oracle(dict(image_size=img, tactile_size=tac, image_patch_size=ip, tactile_patch_size=tp, dim=D, depth=depth, heads=heads,
                           mlp_dim=mlp, image_channels=C, tactile_channels=C, num_tactiles=k),
                      dict(decoder_dim=dd, masking_ratio=ratio, decoder_depth=ddepth, decoder_heads=dheads, num_tactiles=k),
                      B=B, C=C, hw_img=img, hw_tac=tac, k=k, cfg=cfg, seed=sum(geom[:6]))


def test_checkpoint_resume_is_bit_identical():
    """state_dict() / load_state_dict() of the module and of FlatAdam after the parameters were re-homed into GradSync's flat buffer:
    2 steps + save + fresh objects + load + 2 steps == 4 uninterrupted steps, bit for bit (SB3 zip checkpoints, utils/callbacks.py:126-133)."""
    import copy
    from m3l_amd.parallel import FlatAdam, GradSync

    def make():
        torch.manual_seed(0)
        enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128)
        mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, compute_dtype="bf16").to(DEV)
        sync = GradSync(mae)
        return mae, sync, FlatAdam(sync, lr=1e-3, weight_decay=0.01)

    def batch(i):
        g = torch.Generator().manual_seed(50 + i)
        x = {"image": torch.rand(4, 3, 32, 32, generator=g).to(DEV), "tactile1": torch.rand(4, 3, 16, 16, generator=g).to(DEV),
             "tactile2": torch.rand(4, 3, 16, 16, generator=g).to(DEV)}
        return x, [torch.rand(4, 16, generator=g).to(DEV) for _ in range(3)]

    def steps(mae, sync, opt, idx):
        for i in idx:
            x, noises = batch(i)
            sync.zero_grad()
            mae(x, mask_noise=noises).backward()
            sync.finish()
            opt.step()

    a = make()
    steps(*a, range(4))
    b = make()
    steps(*b, range(2))
    ck_model = copy.deepcopy({k: v.cpu() for k, v in b[0].state_dict().items()})
    ck_opt = {k: (v.cpu().clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in b[2].state_dict().items() if k != "param_groups"}
    c = make()
    c[0].load_state_dict(ck_model, strict=True)
    assert c[0].mask_token.data_ptr() == dict(c[0].named_parameters())["mask_token"].data_ptr()
    c[2].load_state_dict({"step": ck_opt["step"], "exp_avg": ck_opt["exp_avg"].to(DEV), "exp_avg_sq": ck_opt["exp_avg_sq"].to(DEV)})
    steps(*c, range(2, 4))
    torch.cuda.synchronize()
    assert torch.equal(a[1].flat_params, c[1].flat_params)
    for (k, v), (_, w) in zip(a[0].state_dict().items(), c[0].state_dict().items()):
        assert torch.equal(v, w), k


@pytest.mark.parametrize("optimizer", ["torch_adam", "flat_adam"])
def test_ppo_like_update_golden(golden_dir, optimizer):
    """The whole MAE update of PPO_MAE.train (models/ppo_mae.py:232-266) against the reference's own run, THROUGH THE OPTIMIZER:
    frame-stacked observations -> m3l_amd.vt_load -> VTMAE -> backward -> Adam step, two mini-batches; losses and stepped weights."""
    from m3l_amd import vt_load
    from m3l_amd.parallel import FlatAdam, GradSync
    z = np.load(os.path.join(golden_dir, "ppo_like_step.npz"))
    fs = int(z["frame_stack"])
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2, frame_stack=fs)
    mae.load_state_dict({k[len("param0/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param0/")}, strict=True)
    mae = mae.to(DEV).train()
    if optimizer == "flat_adam":
        sync = GradSync(mae)
        opt = FlatAdam(sync, lr=1e-3)
    else:
        sync = None
        opt = torch.optim.Adam(mae.parameters(), lr=1e-3)
    obs = {"image": torch.tensor(z["obs/image"]), "tactile": torch.tensor(z["obs/tactile"])}
    o = {"image": obs["image"].permute(0, 2, 3, 1, 4).reshape(6, 32, 32, -1), "tactile": obs["tactile"].reshape(6, -1, 16, 16)}
    for i in range(2):
        opt.zero_grad()
        x = vt_load({k: v[3 * i:3 * i + 3].numpy() for k, v in o.items()}, frame_stack=fs, device=DEV)
        noises = [torch.tensor(z[f"noise/{i}/{j}"]).to(DEV) for j in range(3)]
        loss = mae(x, mask_noise=noises)
        loss.backward()
        if sync is not None:
            sync.finish()
        opt.step()
        assert abs(float(loss.detach()) - float(z["losses"][i])) <= 1e-4 * float(z["losses"][i])
        named = dict(mae.named_parameters())
        for k in [f[len(f"step{i}/"):] for f in z.files if f.startswith(f"step{i}/")]:
            np.testing.assert_allclose(named[k].detach().cpu().numpy(), z[f"step{i}/{k}"], rtol=1e-3, atol=1e-5, err_msg=k)


class _RandQueue:
    """torch.rand(...) pops pre-generated noise (as tests/golden/make_golden.py injects it into the reference)."""

    def __init__(self, noises):
        self.noises, self.real = list(noises), torch.rand

    def __enter__(self):
        def fake(*shape, device=None, **kw):
            n = self.noises.pop(0)
            assert tuple(n.shape) == tuple(shape), (n.shape, shape)
            return n.clone().to(device) if device is not None else n.clone()
        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self.real


@pytest.mark.parametrize("fused", [False, True])
def test_train_iterations_golden(golden_dir, fused):
    """VTMAE.initialize_training + train_iterations against the reference's OWN run of them (models/pretrain_models.py:673-715,
    tests/golden/train_iterations.npz): the same `random` seed draws the same batches from the same replay buffer, the same mask noise is
    injected, and the weights after each of two iterations must agree — with torch.optim.AdamW + clip_grad_norm_ (the reference's calls)
    and with the fused FlatAdamW (train_args['fused_optimizer']: one gradient-norm reduction + one update launch)."""
    import random
    z = np.load(os.path.join(golden_dir, "train_iterations.npz"))
    fs = int(z["frame_stack"])
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2, frame_stack=fs)
    mae.load_state_dict({k[len("param0/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param0/")}, strict=True)
    mae = mae.to(DEV)
    mae.initialize_training({"lr": 1e-3, "batch_size": 4, "fused_optimizer": fused})
    buf = [{"image": z[f"buf/{i}/image"], "tactile": z[f"buf/{i}/tactile"]} for i in range(6)]
    random.seed(74)
    for it in range(2):
        noises = [torch.tensor(z[f"noise/{it}/{j}"]) for j in range(3)]
        with _RandQueue(noises):
            mae.train_iterations(1, buf)
        named = dict(mae.named_parameters())
        for k in [f[len(f"iter{it}/"):] for f in z.files if f.startswith(f"iter{it}/")]:
            np.testing.assert_allclose(named[k].detach().cpu().numpy(), z[f"iter{it}/{k}"], rtol=2e-3, atol=2e-5, err_msg=f"{it} {k}")


@pytest.mark.parametrize("tag", ["vt", "vision_only"])
def test_mae_extractor_golden(golden_dir, tag):
    """m3l_amd.MAEExtractor against the reference's own MAEExtractor.forward (models/pretrain_models.py:788-841): features and the
    gradients that reach the MAE encoder and the extra transformer layer, with and without tactile control."""
    from m3l_amd import MAEExtractor
    z = np.load(os.path.join(golden_dir, "mae_extractor.npz"))
    fs = int(z["frame_stack"])
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2, frame_stack=fs)
    mae.load_state_dict({k[len("param/mae."):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/mae.")}, strict=True)
    ext = MAEExtractor(mae, 64, vision_only_control=(tag == "vision_only"), frame_stack=fs)
    sd = {k[len(tag + "/param/vit_layer."):]: torch.tensor(z[k]) for k in z.files if k.startswith(tag + "/param/vit_layer.")}
    missing, unexpected = ext.vit_layer.load_state_dict(sd, strict=False)
    assert not unexpected and all(not m.startswith("transformer.") for m in missing)
    ext = ext.to(DEV)
    obs = {"image": torch.tensor(z[tag + "/obs/image"]).to(DEV), "tactile": torch.tensor(z[tag + "/obs/tactile"]).to(DEV)}
    feat = ext(obs)
    assert feat.shape == (3, 64) and ext.features_dim == 64
    np.testing.assert_allclose(feat.detach().cpu().numpy(), z[tag + "/features"], rtol=1e-3, atol=1e-4)
    feat.square().mean().backward()
    for name, p in (("mae.encoder.transformer.layers.0.0.to_qkv.weight", mae.encoder.transformer.layers[0][0].to_qkv.weight),
                    ("vit_layer.transformer.layers.0.1.net.1.weight", ext.vit_layer.transformer.layers[0][1].net[1].weight)):
        ref = z[f"{tag}/grad/{name}"]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-3 * np.abs(ref).max() + 1e-7, name


@pytest.mark.parametrize("arch", ["patch_fp32", "patch_bf16", "earlyconv_bf16", "vision_only_control_fp32"])
def test_fused_extractor_is_bit_identical_to_module_chain(arch):
    """m3l_extractor_fwd / m3l_extractor_bwd (MAEExtractor.forward, models/pretrain_models.py:819-841, as one autograd node) against
    get_embeddings -> Transformer -> torch.mean through the per-module Functions: the same kernels up to the token mean (the library sums
    the tokens in order, torch.mean in its own order), so features and gradients agree to fp32 round-off; also under no_grad (the rollout
    path: identical to the grad-mode forward)."""
    from m3l_amd import MAEExtractor
    from m3l_amd import functional as Fn
    dt = "fp32" if arch.endswith("fp32") else "bf16"
    D, fs = 128, 2
    kw = dict(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=D, depth=2, heads=2, mlp_dim=256,
              image_channels=3 * fs, tactile_channels=3 * fs, frame_stack=fs)
    mkw = dict(decoder_dim=D, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, frame_stack=fs, early_conv_masking="earlyconv" in arch)
    B = 5
    g = torch.Generator(device="cpu").manual_seed(5)
    obs = {"image": torch.rand(B, fs, 32, 32, 3, generator=g).to(DEV), "tactile": (torch.rand(B, fs, 6, 16, 16, generator=g) * 2 - 1).to(DEV)}

    def run(fused, extra_mae_loss=False):
        torch.manual_seed(4)
        mae = VTMAE(encoder=VTT(**kw), compute_dtype=dt, **mkw).to(DEV)
        ext = MAEExtractor(mae, D, arch.startswith("vision_only_control"), fs).to(DEV)
        assert ext.vit_layer.transformer.compute_dtype == dt          # the extractor's own layer computes in the MAE's type
        keep = Fn.FUSED_EXTRACTOR
        Fn.FUSED_EXTRACTOR = fused
        try:
            with torch.no_grad():
                f0 = ext(obs).clone()
            f = ext(obs)
            f.square().mean().backward()
        finally:
            Fn.FUSED_EXTRACTOR = keep
        torch.cuda.synchronize()
        return f0, f.detach().clone(), {n: (None if p.grad is None else p.grad.clone()) for n, p in ext.named_parameters()}

    a0, a, ga = run(False)
    b0, b, gb = run(True)
    tol = 1e-5 if dt == "fp32" else 2e-2      # bf16: a last-bit difference of the pooled feature moves bf16-rounded gradients downstream
    assert torch.equal(a0, a) and torch.equal(b0, b)
    assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-7
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n
        if ga[n] is not None:
            scale = float(ga[n].abs().max()) + 1e-12
            assert float((ga[n] - gb[n]).abs().max()) <= tol * scale, (n, float((ga[n] - gb[n]).abs().max()), scale)



def test_dino_cat_extractor_glue_golden(golden_dir):
    """m3l_amd.DinoCatMAEExtractor against the reference's own class (models/pretrain_models_dino_cat_mae.py:793-904, eval mode) with a
    stand-in `dino_model`: the frame-stack reshape, the middle-frame slice (frame_stack 4 -> channels 3..5), the concat order and the
    3-Linear MLP are the reference's; the image encoder itself is pinned by the DINOv2 tests."""
    from m3l_amd import DinoCatMAEExtractor
    z = np.load(os.path.join(golden_dir, "dino_cat_extractor.npz"))
    fs, D = int(z["frame_stack"]), 64

    class StandInDino(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(3, D)

        def forward(self, x):
            assert x.shape[1] == 3
            return self.proj(x.mean(dim=(2, 3)))

    enc = VTT(image_size=28, tactile_size=28, image_patch_size=7, tactile_patch_size=7, dim=D, depth=1, heads=2, mlp_dim=128,
              image_channels=3 * fs, tactile_channels=3 * fs, num_tactiles=2, frame_stack=fs)
    mae = VTMAE(encoder=enc, decoder_dim=D, masking_ratio=0.75, decoder_depth=1, decoder_heads=2, num_tactiles=2, frame_stack=fs)
    mae.load_state_dict({k[len("param/mae."):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/mae.")}, strict=True)
    dino = StandInDino()
    dino.load_state_dict({"proj.weight": torch.tensor(z["param/dino.proj.weight"]), "proj.bias": torch.tensor(z["param/dino.proj.bias"])})
    ext = DinoCatMAEExtractor(dino, mae, D, vision_only_control=False, frame_stack=fs)
    sd = {k[len("param/ext."):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/ext.")}
    missing, unexpected = ext.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith(("mae_model.", "dino_model.", "vit_layer.")) and not m.startswith("vit_layer.transformer.")
                                  for m in missing), (missing, unexpected)
    ext = ext.to(DEV).eval()
    obs = {"image": torch.tensor(z["obs/image"]).to(DEV), "tactile": torch.tensor(z["obs/tactile"]).to(DEV)}
    with torch.no_grad():
        feat = ext(obs)
    np.testing.assert_allclose(feat.cpu().numpy(), z["features"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("D,heads,n_img_hw,B", [(192, 3, 64, 5), (128, 2, 32, 3)])
def test_fused_attention_block_matches_unfused(D, heads, n_img_hw, B):
    """Short-sequence encoder layers run LN1 + QKV + attention + out-proj + residual + LN2 as ONE launch (attn_block.hip).  The whole
    MAE step with it against the same step on the five separate kernels: loss and every gradient (the backward reads the
    activations the fused kernel saved)."""
    from m3l_amd import _lib as L
    torch.manual_seed(0)
    enc = VTT(image_size=n_img_hw, tactile_size=n_img_hw // 2, image_patch_size=8, tactile_patch_size=4, dim=D, depth=3, heads=heads,
              mlp_dim=2 * D, num_tactiles=2)
    mae = VTMAE(encoder=enc, decoder_dim=D, masking_ratio=0.75, decoder_depth=1, decoder_heads=heads, num_tactiles=2,
                compute_dtype="bf16").to(DEV)
    with torch.no_grad():
        for p in mae.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator(device=DEV).manual_seed(1)
    n = (n_img_hw // 8) ** 2
    x = {"image": torch.rand(B, 3, n_img_hw, n_img_hw, device=DEV, generator=g),
         "tactile1": torch.rand(B, 3, n_img_hw // 2, n_img_hw // 2, device=DEV, generator=g),
         "tactile2": torch.rand(B, 3, n_img_hw // 2, n_img_hw // 2, device=DEV, generator=g)}
    noises = [torch.rand(B, n, device=DEV, generator=g) for _ in range(3)]
    res = {}
    old = L.lib().m3l_set_attn_block(1)
    try:
        for mode in (0, 1, 3):        # unfused; forward blocks + MLP backward block (default); + attention backward block
            L.lib().m3l_set_attn_block(mode)
            mae.zero_grad(set_to_none=True)
            loss = mae(x, mask_noise=noises)
            loss.backward()
            emb = mae.get_embeddings(x, eval=False).detach().clone()          # all tokens: decoder-length sequences stay unfused
            res[mode] = (float(loss.detach()), emb, {k: p.grad.clone() for k, p in mae.named_parameters() if p.grad is not None})
    finally:
        L.lib().m3l_set_attn_block(old)
    for mode in (1, 3):
        assert abs(res[mode][0] - res[0][0]) <= 2e-3 * abs(res[0][0]), (mode, res[mode][0], res[0][0])
        assert (res[mode][1] - res[0][1]).abs().max().item() <= 3e-2 * res[0][1].abs().max().item()
        num = den = 0.0
        for k, g1 in res[mode][2].items():
            g0 = res[0][2][k]
            num += float((g1 - g0).double().square().sum())
            den += float(g0.double().square().sum())
        assert (num / den) ** 0.5 <= 2e-2, (mode, (num / den) ** 0.5)


def test_gradsync_second_backward_accumulates():
    """with a GradSync installed the kernels write gradients straight into the flat buffer (overwrite).  A second backward
    before zero_grad() — the reference's `separate_optimizer=False` update runs mae_loss.backward() and then the policy loss through
    MAEExtractor -> get_embeddings (ppo_mae.py:249-266) — must ADD to the first, as autograd does without a GradSync."""
    from m3l_amd.parallel import GradSync
    torch.manual_seed(5)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2).to(DEV)
    B = 4
    x = {"image": torch.rand(B, 3, 32, 32, device=DEV), "tactile1": torch.rand(B, 3, 16, 16, device=DEV),
         "tactile2": torch.rand(B, 3, 16, 16, device=DEV)}
    noises = [torch.rand(B, 16, device=DEV) for _ in range(3)]

    def two_backwards():
        mae(x, mask_noise=noises).backward()
        mae.get_embeddings(x, eval=False).square().mean().backward()

    two_backwards()                                               # plain autograd accumulation = the reference semantics
    ref = {n: p.grad.clone() for n, p in mae.named_parameters() if p.grad is not None}
    mae.zero_grad(set_to_none=True)
    sync = GradSync(mae)
    sync.zero_grad()
    mae(x, mask_noise=noises).backward()
    # the first backward left weight gradients un-joined on the library's side stream (deferred join) ...
    assert L.lib().m3l_side_pending() > 0
    mae.get_embeddings(x, eval=False).square().mean().backward()
    # ... and the second one, which accumulates into the same flat views through autograd, joined them first (no race
    # between AccumulateGrad on the compute stream and the first backward's weight-gradient kernels on the side stream)
    assert L.lib().m3l_side_pending() == 0
    sync.finish()
    sync.finish()                                                 # idempotent
    for n, p in mae.named_parameters():
        if n in ref:
            assert torch.allclose(p.grad, ref[n], rtol=1e-5, atol=1e-7), n
    # and a parameter update between forward and backward is refused, as autograd refuses it
    loss = mae(x, mask_noise=noises)
    with torch.no_grad():
        mae.to_pixels.weight.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()
    # ... and so is an observation batch refilled in place (the backward re-reads the raw frames)
    with torch.no_grad():
        mae.to_pixels.weight.sub_(1.0)
    sync.zero_grad()
    loss = mae(x, mask_noise=noises)
    x["image"].mul_(0.5)
    with pytest.raises(RuntimeError, match="input 0 was modified by an inplace operation"):
        loss.backward()


def test_gradsync_two_fused_steps_accumulate():
    """micro-batch accumulation under the fused step with a non-communicating GradSync — two `mae(x).backward()` calls before
    zero_grad().  The fused node has ONE autograd input (the anchor parameter); on the second backward the kernels may not overwrite
    the flat views, so the fresh gradients are added into them inside MaeStepFn.backward.  Compared with plain autograd accumulation
    (no GradSync) on the same two micro-batches."""
    from m3l_amd.parallel import GradSync
    torch.manual_seed(6)
    enc = VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=2, heads=2, mlp_dim=128)
    mae = VTMAE(encoder=enc, decoder_dim=64, masking_ratio=0.75, decoder_depth=1, decoder_heads=2).to(DEV)
    B = 4
    xs = [{"image": torch.rand(B, 3, 32, 32, device=DEV), "tactile1": torch.rand(B, 3, 16, 16, device=DEV),
           "tactile2": torch.rand(B, 3, 16, 16, device=DEV)} for _ in range(2)]
    noises = [[torch.rand(B, 16, device=DEV) for _ in range(3)] for _ in range(2)]
    for x, n in zip(xs, noises):
        mae(x, mask_noise=n).backward()
    ref = {n: p.grad.clone() for n, p in mae.named_parameters() if p.grad is not None}
    mae.zero_grad(set_to_none=True)
    sync = GradSync(mae)
    sync.zero_grad()
    for x, n in zip(xs, noises):
        loss = mae(x, mask_noise=n)
        assert type(loss.grad_fn).__name__ == "MaeStepFnBackward"
        loss.backward()
    sync.finish()
    torch.cuda.synchronize()
    for n, p in mae.named_parameters():
        if n in ref:
            assert torch.allclose(p.grad, ref[n], rtol=1e-5, atol=1e-7), n


def test_two_rank_data_parallel_step_on_one_gpu():
    """N > 1 control flow of the real HIP step (chunked transformer backward, GradSync prefix buckets, FlatAdam): two processes share
    cuda:0 and exchange gradients over gloo (RCCL refuses two ranks per device).  tools/dp_two_rank_check.py asserts that both ranks
    hold bit-identical parameters after 3 optimizer steps and that they equal a single-process run on the concatenated batch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(root, "tools", "dp_two_rank_check.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ranks identical: True" in r.stdout, r.stdout[-2000:]


def test_wgrad_inline_switch_is_bit_identical():
    """m3l_set_wgrad_inline (bench.py's stand-alone roofline pass, the PMC passes): the same kernels on the caller's stream instead of the
    side stream — every gradient bit-identical, nothing left pending."""
    from m3l_amd import _lib as L
    from m3l_amd.parallel import GradSync

    def run(inline):
        torch.manual_seed(0)
        enc = VTT(image_size=64, tactile_size=32, image_patch_size=8, tactile_patch_size=4, dim=192, depth=3, heads=3, mlp_dim=768, num_tactiles=2)
        mae = VTMAE(encoder=enc, decoder_dim=192, masking_ratio=0.75, decoder_depth=2, decoder_heads=3, num_tactiles=2, compute_dtype="bf16").to(DEV)
        sync = GradSync(mae)
        g = torch.Generator(device=DEV).manual_seed(5)
        x = {"image": torch.rand(6, 3, 64, 64, device=DEV, generator=g), "tactile1": torch.rand(6, 3, 32, 32, device=DEV, generator=g),
             "tactile2": torch.rand(6, 3, 32, 32, device=DEV, generator=g)}
        noises = [torch.rand(6, 64, device=DEV, generator=g) for _ in range(3)]
        old = L.lib().m3l_set_wgrad_inline(1 if inline else 0)
        try:
            sync.zero_grad()
            mae(x, mask_noise=noises).backward()
            if inline:
                assert L.lib().m3l_side_pending() == 0
            sync.finish()
            torch.cuda.synchronize()
        finally:
            L.lib().m3l_set_wgrad_inline(old)
        return sync.flat.clone()

    assert torch.equal(run(False), run(True))


def test_attention_block_phase_stamps():
    """Profiling hook of the forward attention block (m3l_set_attn_phase_buffer, tools/attn_phase_probe.py): six increasing shader-clock
    stamps per sample while the buffer is set, nothing written once it is cleared, results unchanged."""
    from m3l_amd import Transformer
    from m3l_amd import _lib as L
    torch.manual_seed(0)
    tf = Transformer(192, 2, 3, 64, 768)
    tf.compute_dtype = "bf16"
    tf = tf.to(DEV)
    B = 9
    x = torch.randn(B, 48, 192, device=DEV)
    y0 = tf(x).clone()
    ts = torch.zeros(B, 8, dtype=torch.int64, device=DEV)
    L.lib().m3l_set_attn_phase_buffer(ts.data_ptr())
    try:
        y1 = tf(x).clone()
        torch.cuda.synchronize()
    finally:
        L.lib().m3l_set_attn_phase_buffer(None)
    t = ts.cpu()
    assert bool((t[:, 0] > 0).all()) and bool((t[:, 1:6] > t[:, 0:5]).all()), t
    ts.zero_()
    y2 = tf(x)
    torch.cuda.synchronize()
    assert int(ts.abs().sum()) == 0
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
