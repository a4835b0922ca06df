"""Per-kernel numerics on a real MI355X: every hand-written HIP kernel, through the C ABI, against a plain PyTorch fp32
reference of the same op (fp32 compute: tight tolerance; bf16 compute: bf16-rounding tolerance)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from m3l_amd import _lib as L  # noqa: E402


def _s():
    return torch.cuda.current_stream().cuda_stream


def _t(code):
    return torch.bfloat16 if code else torch.float32


def _tol(code):
    return dict(rtol=2e-2, atol=2e-2) if code else dict(rtol=2e-4, atol=2e-4)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    L.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 192, 192), (77, 48, 48), (1000, 576, 192), (129, 200, 776), (64, 8, 8)])
def test_gemm_nt_plain(dev, code, M, N, K):
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev).to(_t(code))
    W = torch.randn(N, K, device=dev).to(_t(code))
    bias = torch.randn(N, device=dev)
    out32 = torch.full((M, N), float("nan"), device=dev)
    out_t = torch.zeros(M, N, device=dev, dtype=_t(code))
    L.check(L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, L.ptr(bias), None, L.ptr(out32), L.ptr(out_t),
                                   None, None, 0, N, _s()), "gemm_nt")
    ref = A.float() @ W.float().t() + bias
    scale = ref.abs().max().item()
    assert (out32 - ref).abs().max().item() <= (3e-3 if code else 2e-5) * scale + 1e-5
    torch.testing.assert_close(out_t.float(), ref, rtol=2e-2 if code else 1e-4, atol=(2e-2 if code else 1e-4) * scale)


@pytest.mark.parametrize("code", [0, 1])
def test_gemm_nt_asymmetric_layout(dev, code):
    """A = I against an ASYMMETRIC W catches a swapped row/col accumulator map (guide: 'Always A=I-check with asymmetric B')."""
    M = N = K = 128
    A = torch.eye(M, device=dev).to(_t(code))
    W = (torch.arange(N, device=dev)[:, None] * 2 + torch.arange(K, device=dev)[None, :] % 7).float().to(_t(code))
    out32 = torch.zeros(M, N, device=dev)
    L.check(L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, L.ptr(out32), None, None, None, 0, N, _s()), "gemm_nt")
    torch.testing.assert_close(out32, W.float().t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("code", [0, 1])
def test_gemm_nt_epilogues(dev, code):
    torch.manual_seed(5)
    M, N, K = 200, 256, 64
    A = (0.3 * torch.randn(M, K, device=dev)).to(_t(code))
    W = (0.3 * torch.randn(N, K, device=dev)).to(_t(code))
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    # GELU + pre-activation copy (fc1)
    u = torch.zeros(M, N, device=dev, dtype=_t(code))
    h = torch.zeros(M, N, device=dev, dtype=_t(code))
    L.check(L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, L.ptr(bias), None, None, L.ptr(h), L.ptr(u), None, 1, N, _s()), "fc1")
    ref_u = A.float() @ W.float().t() + bias
    torch.testing.assert_close(u.float(), ref_u, **_tol(code))
    torch.testing.assert_close(h.float(), torch.nn.functional.gelu(u.float()), **_tol(code))
    # bias + residual into f32 (out-proj / fc2)
    out = torch.zeros(M, N, device=dev)
    L.check(L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, L.ptr(bias), L.ptr(res), L.ptr(out), None, None, None, 0, N, _s()), "res")
    torch.testing.assert_close(out, ref_u + res, **_tol(code))
    # dgrad through GELU: acc * gelu'(u)
    du = torch.zeros(M, N, device=dev, dtype=_t(code))
    L.check(L.lib().m3l_op_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, M, N, K, None, None, None, L.ptr(du), None, L.ptr(u), 0, N, _s()), "dgelu")
    uu = u.float().clone().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    torch.testing.assert_close(du.float(), (A.float() @ W.float().t()) * uu.grad, **_tol(code))


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("M,N,K", [(512, 128, 128), (1000, 192, 576), (333, 48, 192), (4096, 768, 192), (70, 8, 200), (9000, 576, 192),
                                   (2500, 1536, 384), (8300, 192, 768), (300, 264, 136)])
def test_gemm_tn(dev, code, M, N, K):
    torch.manual_seed(M + N)
    Y = torch.randn(M, N, device=dev).to(_t(code))
    X = torch.randn(M, K, device=dev).to(_t(code))
    nb = L.lib().m3l_op_gemm_tn_ws_bytes(M, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = torch.full((N, K), float("nan"), device=dev)
    L.check(L.lib().m3l_op_gemm_tn(code, L.ptr(Y), N, L.ptr(X), K, M, N, K, L.ptr(ws), nb, L.ptr(out), K, _s()), "gemm_tn")
    ref = Y.float().t() @ X.float()
    assert (out - ref).abs().max().item() <= (2e-3 if code else 2e-5) * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize("M,shapes", [
    (777, [(192, 768), (768, 192), (576, 192), (192, 192)] * 2),                 # two ViT-Tiny layers (the 384 x 192 tiles: 2 + 2 + 2 + 1 per layer, half-empty tiles), ragged M
    (5000, [(256, 512), (512, 256), (768, 256), (256, 256)]),                    # M3L's default width (256 x 256 tiles)
    (1030, [(384, 1536), (1536, 384), (1152, 384), (384, 384)]),                 # ViT-Small layer
    (2100, [(192, 768), (64, 32), (8, 200), (264, 136), (192, 48)]),             # mixed: transformer weights beside small / odd ones
    (12288, [(192, 768), (768, 192), (576, 192), (192, 192)] * 4),               # the encoder's four-layer group of cfg 2
])
def test_gemm_tn_grouped(dev, M, shapes):
    """Grouped weight gradients dW_i = Y_i^T X_i through ONE launch (wgrad.hip via m3l_op_gemm_tn_grouped) against torch, bf16 operands:
    every nn.Linear.weight.grad of a transformer layer group (vit_pytorch Attention / FeedForward via loss.backward(), ppo_mae.py:263)."""
    import ctypes as C
    torch.manual_seed(M)
    cnt = len(shapes)
    Ys = [(0.5 * torch.randn(M, n, device=dev)).to(torch.bfloat16) for n, _ in shapes]
    Xs = [(0.5 * torch.randn(M, k, device=dev)).to(torch.bfloat16) for _, k in shapes]
    outs = [torch.full((n, k), float("nan"), device=dev) for n, k in shapes]
    Ns, Ks = (C.c_int * cnt)(*[n for n, _ in shapes]), (C.c_int * cnt)(*[k for _, k in shapes])
    nb = L.lib().m3l_op_gemm_tn_grouped_ws_bytes(1, cnt, M, Ns, Ks)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for rep in range(2):            # twice: bit-reproducible (fixed-order slab reduce)
        L.check(L.lib().m3l_op_gemm_tn_grouped(1, cnt, M, L.ptr_array(Ys), Ns, L.ptr_array(Xs), Ks, Ns, Ks, L.ptr_array(outs), L.ptr(ws), nb, _s()),
                "gemm_tn_grouped")
        if rep == 0:
            first = [o.clone() for o in outs]
    for i in range(cnt):
        ref = Ys[i].float().t() @ Xs[i].float()
        assert (outs[i] - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-4, (i, shapes[i])
        assert torch.equal(outs[i], first[i]), i


def _attn_ref(qkv, B, n, H):
    q, k, v = [t.reshape(B, n, H, 64).transpose(1, 2) for t in qkv.float().reshape(B, n, 3 * H * 64).chunk(3, dim=-1)]
    dots = (q @ k.transpose(-1, -2)) * 0.125
    o = (dots.softmax(-1) @ v).transpose(1, 2).reshape(B * n, H * 64)
    return o, torch.logsumexp(dots, -1)     # (B*n, H*64), (B, H, n)


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("B,n,H", [(2, 48, 3), (3, 16, 1), (2, 10, 4), (2, 75, 2), (1, 192, 3), (1, 113, 6), (1, 452, 1)])
def test_attention_fwd_bwd(dev, code, B, n, H):
    torch.manual_seed(n * 7 + H)
    qkv = torch.randn(B * n, 3 * H * 64, device=dev).to(_t(code))
    dO = torch.randn(B * n, H * 64, device=dev).to(_t(code))
    o = torch.zeros(B * n, H * 64, device=dev, dtype=_t(code))
    lse = torch.zeros(B, H, n, device=dev)
    L.check(L.lib().m3l_op_attn_fwd(code, L.ptr(qkv), L.ptr(o), L.ptr(lse), B, n, H, _s()), "attn_fwd")
    ref_in = qkv.float().clone().requires_grad_(True)
    ref_o, ref_lse = _attn_ref(ref_in, B, n, H)
    torch.testing.assert_close(o.float(), ref_o, **_tol(code))
    torch.testing.assert_close(lse, ref_lse, rtol=1e-4, atol=2e-2 if code else 1e-4)
    dsum = torch.zeros(B, H, n, device=dev)
    dqkv = torch.full_like(qkv, float("nan"))
    L.check(L.lib().m3l_op_attn_bwd(code, L.ptr(qkv), L.ptr(o), L.ptr(dO), L.ptr(lse), L.ptr(dsum), L.ptr(dqkv), B, n, H, _s()), "attn_bwd")
    (ref_o * dO.float()).sum().backward()
    scale = ref_in.grad.abs().max().item()
    assert (dqkv.float() - ref_in.grad).abs().max().item() <= (4e-2 if code else 2e-4) * scale + 1e-5


def test_attention_softmax_spike(dev):
    """One key dominating from a late tile forces the online-softmax rescale branch (guide rule 26)."""
    B, n, H = 1, 96, 1
    torch.manual_seed(3)
    qkv = 0.1 * torch.randn(B * n, 192, device=dev)
    qkv[5, 0:64] = 4.0        # query 5
    qkv[80, 64:128] = 4.0     # key 80 (third 32-key tile) lines up with it -> score 4*4*64/8 = 128 >> others
    o = torch.zeros(B * n, 64, device=dev)
    lse = torch.zeros(B, H, n, device=dev)
    L.check(L.lib().m3l_op_attn_fwd(0, L.ptr(qkv), L.ptr(o), L.ptr(lse), B, n, H, _s()), "attn_fwd")
    ref_o, ref_lse = _attn_ref(qkv, B, n, H)
    torch.testing.assert_close(o, ref_o, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(lse, ref_lse, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("D", [48, 64, 192, 384, 592, 768, 1024])
def test_layernorm_fwd_bwd(dev, D):
    torch.manual_seed(D)
    M = 1031
    x = torch.randn(M, D, device=dev) * 2 + 0.5
    g = torch.randn(D, device=dev)
    b = torch.randn(D, device=dev)
    y = torch.zeros(M, D, device=dev)
    L.check(L.lib().m3l_layernorm_fwd(0, L.ptr(x), M, D, L.ptr(g), L.ptr(b), 1e-5, L.ptr(y), None, _s()), "ln_fwd")
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    dy = torch.randn(M, D, device=dev)
    res = torch.randn(M, D, device=dev)
    dx = torch.zeros(M, D, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ws = torch.empty(L.lib().m3l_layernorm_ws_bytes(D), dtype=torch.uint8, device=dev)
    L.check(L.lib().m3l_layernorm_bwd(0, L.ptr(dy), L.ptr(x), M, D, L.ptr(g), 1e-5, L.ptr(res), L.ptr(dx), L.ptr(ws), L.ptr(dg), L.ptr(db), _s()), "ln_bwd")
    ref.backward(dy)
    torch.testing.assert_close(dx, xr.grad + res, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dg, gr.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db, br.grad, rtol=1e-3, atol=1e-3)


def test_vt_load_matches_golden(dev, golden_dir):
    import os
    import numpy as np
    from m3l_amd import vt_load
    for fs in (1, 2):
        z = np.load(os.path.join(golden_dir, f"vt_load_fs{fs}.npz"))
        out = vt_load({"image": z["in/image"], "tactile": z["in/tactile"]}, frame_stack=fs)
        assert sorted(out.keys()) == sorted(k[4:] for k in z.files if k.startswith("out/"))
        for k, v in out.items():
            assert v.dtype == torch.float32 and v.is_cuda
            np.testing.assert_array_equal(v.cpu().numpy(), z["out/" + k])
    # uint8 frames go to the device as bytes; non-default normalisation ranges: bit-exact with the reference's fp32 arithmetic
    z = np.load(os.path.join(golden_dir, "vt_load_u8.npz"))
    assert z["in/image"].dtype == np.uint8
    out = vt_load({"image": z["in/image"], "tactile": z["in/tactile"]}, image_normalization=[0, 255], tactile_normalization=[-2, 3], frame_stack=2)
    for k, v in out.items():
        np.testing.assert_array_equal(v.cpu().numpy(), z["out/" + k])
    # ranges that binary floats cannot represent ([0.1, 0.3], [-0.7, 0.9]): the span is formed in double and rounded once, as the
    # reference's `tensor / (hi - lo)` does — fp32(0.3) - fp32(0.1) would be one ulp off
    zf = np.load(os.path.join(golden_dir, "vt_load_frac.npz"))
    outf = vt_load({"image": zf["in/image"], "tactile": zf["in/tactile"]}, image_normalization=[0.1, 0.3], tactile_normalization=[-0.7, 0.9])
    for k, v in outf.items():
        np.testing.assert_array_equal(v.cpu().numpy(), zf["out/" + k])
    # a path argument: .npz of arrays is accepted, the reference's pickled .npy is refused with an explanation
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "obs.npz"), image=z["in/image"], tactile=z["in/tactile"])
        out2 = vt_load(os.path.join(d, "obs.npz"), image_normalization=[0, 255], tactile_normalization=[-2, 3], frame_stack=2)
        assert all(torch.equal(out2[k], out[k]) for k in out)
        with pytest.raises(NotImplementedError, match="unpickling is refused"):
            vt_load(os.path.join(d, "obs.npy"))


@pytest.mark.parametrize("M,K,N", [(3, 128, 128), (128, 768, 768), (37, 256, 64)])
def test_fusion_mlp_linear_act(dev, M, K, N):
    """functional.LinearActFn / Concat2Fn (the cfg-5 extractor's trainable fusion MLP, models/pretrain_models_dino_cat_mae.py:828-836,
    899-903) against torch: Linear + ReLU + Dropout with a GIVEN keep-mask (scale 1 / 0.9), forward and every gradient; f32 MFMA compute."""
    from m3l_amd import functional as Fn
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True)
    mask = (torch.rand(M, N, generator=g) >= 0.1).to(torch.uint8).to(dev)
    cot = torch.randn(M, N, generator=g).to(dev)
    for relu, mk in ((True, mask), (True, None), (False, None)):
        scale = 1.0 / 0.9 if mk is not None else 1.0
        y = Fn.LinearActFn.apply(x, w, b, relu, mk, scale)
        gx, gw, gb = torch.autograd.grad((y * cot).sum(), (x, w, b))
        xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
        yr = xr @ wr.t() + br
        if relu:
            yr = torch.relu(yr)
        if mk is not None:
            yr = yr * mk.double() * scale
        rx, rw, rb = torch.autograd.grad((yr * cot.double()).sum(), (xr, wr, br))
        for a, r in ((y, yr), (gx, rx), (gw, rw), (gb, rb)):
            assert float((a.double() - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-6, (relu, mk is not None)
    a = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    c = torch.randn(M, N, generator=g).to(dev)
    cat = Fn.Concat2Fn.apply(a, c)
    assert torch.equal(cat, torch.cat((a, c), -1))
    (ga,) = torch.autograd.grad((cat * cat).sum(), (a,))
    assert torch.equal(ga, 2 * a.detach())


@pytest.mark.parametrize("dim,C,hw,tactile,B,nsrc", [(64, 3, 32, False, 3, 1), (64, 3, 16, True, 2, 2), (128, 6, 64, False, 2, 1), (256, 12, 64, False, 3, 1),
                                                     (256, 12, 32, True, 2, 2), (128, 3, 40, False, 2, 1), (256, 3, 24, True, 1, 3)])
def test_direct_conv_stem_vs_im2col_and_torch(dev, dim, C, hw, tactile, B, nsrc):
    """conv.hip (direct / implicit-GEMM convolutions of the EarlyCNN stem, models/pretrain_models.py:37-56) against the im2col + GEMM path
    (m3l_set_direct_conv(0)) and against torch's Conv2d in fp64: stem outputs, every weight / bias gradient; output sizes that are not
    multiples of the 8 x 8 tile (40 -> 20 / 10 / 5, 24 -> 12 / 6 / 6), 1-3 sources sharing the stem, 3 / 6 / 12 input channels."""
    from m3l_amd.pretrain_models import EarlyCNN
    torch.manual_seed(dim + hw)
    stem = EarlyCNN(C, dim, key="tactile" if tactile else "image").to(dev)
    g = torch.Generator().manual_seed(hw)
    xs = [torch.rand(B, C, hw, hw, generator=g).to(dev) for _ in range(nsrc)]
    res = {}
    for direct in (0, 1):
        old = L.lib().m3l_set_direct_conv(direct)
        try:
            stem.zero_grad(set_to_none=True)
            out = stem.run(xs, "bf16")
            cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
            (out * cot).sum().backward()
            torch.cuda.synchronize()
        finally:
            L.lib().m3l_set_direct_conv(old)
        res[direct] = (out.detach().clone(), {n: p.grad.clone() for n, p in stem.named_parameters()})
    # fp64 torch reference
    ref = EarlyCNN(C, dim, key="tactile" if tactile else "image").double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in stem.state_dict().items()})
    xr = torch.cat([x.cpu().double() for x in xs], 0)
    h = torch.relu(ref.conv1(xr)); h = torch.relu(ref.conv2(h)); h = torch.relu(ref.conv3(h)); y = ref.conv4(h)
    yr = y.flatten(2).transpose(1, 2)                       # (nsrc B, h w, dim): tokens, source-major
    (yr * cot.cpu().double()).sum().backward()
    rel = lambda a, b: float((a.double().cpu() - b).abs().max()) / max(1e-9, float(b.abs().max()))      # noqa: E731
    worst = {}
    for direct in (0, 1):
        out, gr = res[direct]
        worst[direct] = max([("out", rel(out, yr.detach()))] + [(n, rel(gr[n], p.grad)) for n, p in ref.named_parameters()], key=lambda t: t[1])
    # direct vs im2col: the same bf16 operands and roundings between layers, different accumulation order only
    dvi = max([("out", rel(res[1][0], res[0][0].double().cpu()))] + [(n, rel(res[1][1][n], res[0][1][n].double().cpu())) for n in res[0][1]],
              key=lambda t: t[1])
    print(f"\n[conv] dim={dim} C={C} hw={hw} tactile={tactile}: worst vs fp64 torch im2col {worst[0]}, direct {worst[1]}; direct vs im2col {dvi}")
    # bf16 through four layers against fp64: the direct path must be as close to the exact result as the im2col path is (x 1.5 + slack)
    assert worst[1][1] <= max(0.1, 1.5 * worst[0][1]), worst
    assert dvi[1] <= 0.08, dvi
