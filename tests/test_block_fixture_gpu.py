"""The transformer kernels (LayerNorm, QKV / out-proj / MLP GEMMs, attention, GELU; per-op and fused block kernels) against

  * tests/golden/block_stack.npz — outputs and gradients of the reference's OWN pre-norm block
    (/root/reference/tactile_ssl/model/layers/block.py:89-114, attention.py:54-76, mlp.py:34-40; written by
    tests/golden/make_golden.py::run_block_stack from the imported reference classes), and
  * the CPU oracle at the sequence lengths / widths where the block kernels change code path (partial last tile, one or two key
    tiles, the short MLP ring), per parameter, for every block mode.

bf16 bounds are relative to the largest element of the reference tensor."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from m3l_amd import Transformer  # noqa: E402
from m3l_amd import _lib as L  # noqa: E402
from oracle import vtmae_oracle as O  # noqa: E402

DEV = "cuda:0"


def _run(tf, x, cot, mode):
    old = L.lib().m3l_set_attn_block(mode)
    try:
        tf.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = tf(xg)
        (y * cot).sum().backward()
        torch.cuda.synchronize()
    finally:
        L.lib().m3l_set_attn_block(old)
    return y.detach().cpu(), xg.grad.cpu(), {k: p.grad.cpu().clone() for k, p in tf.named_parameters()}


class _Residual:
    """m3l_set_residual_bf16 for the duration of a test (1 = the default since round 4: the layer inputs / outputs and the residual gradient
    of a fully fused stack travel as bf16; 0 = fp32, the mode the kernel-family equivalence bounds below were set in)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = L.lib().m3l_set_residual_bf16(self.on)

    def __exit__(self, *a):
        L.lib().m3l_set_residual_bf16(self.old)


# bounds (output, input gradient, worst parameter gradient; relative to the tensor's max) of a 2-layer bf16 stack against the fp32 oracle:
# fp32 residual stream / bf16 residual stream (every half layer's result rounded once more: 2^-9 relative per element)
_BOUNDS = {0: (5e-3, 1e-2, 2e-2), 1: (1.5e-2, 2.5e-2, 3e-2)}


def _relmax(a, ref):
    return float((a - ref).abs().max()) / max(1e-7, float(ref.abs().max()))


@pytest.mark.parametrize("n", [48, 192])
@pytest.mark.parametrize("dt,mode", [("fp32", 1), ("bf16", 0), ("bf16", 1), ("bf16", 3)])
def test_transformer_vs_reference_held_block(golden_dir, n, dt, mode):
    z = np.load(os.path.join(golden_dir, "block_stack.npz"))
    D, depth, heads, mlp = [int(v) for v in z["meta"]]
    tf = Transformer(D, depth, heads, 64, mlp)
    tf.load_state_dict({k[len("param/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    tf.compute_dtype = dt
    tf = tf.to(DEV)
    x, cot = torch.tensor(z[f"n{n}/x"]).to(DEV), torch.tensor(z[f"n{n}/cot"]).to(DEV)
    y, dx, grads = _run(tf, x, cot, mode)
    ytol, gtol = (1e-5, 1e-5) if dt == "fp32" else (1e-2, 2e-2)
    ey = _relmax(y, torch.tensor(z[f"n{n}/y"]))
    edx = _relmax(dx, torch.tensor(z[f"n{n}/dx"]))
    worst = max(((k, _relmax(g, torch.tensor(z[f"n{n}/grad/{k}"]))) for k, g in grads.items()), key=lambda t: t[1])
    print(f"\n[block_stack] n={n} {dt} mode {mode}: y {ey:.2e} dx {edx:.2e} worst grad {worst[0]} {worst[1]:.2e}")
    assert ey <= ytol and edx <= gtol and worst[1] <= gtol, (ey, edx, worst)


@pytest.mark.parametrize("D,heads,mlp,n,B", [(192, 3, 768, 20, 5), (192, 3, 768, 40, 3), (192, 3, 64, 48, 4), (128, 2, 64, 33, 3),
                                             (128, 2, 256, 17, 6), (192, 3, 384, 1, 7), (128, 2, 128, 47, 2)])
@pytest.mark.parametrize("rb", [0, 1])
def test_block_kernels_edge_lengths(D, heads, mlp, n, B, rb):
    """Sequence lengths that take the block kernels' other paths — a partial last 16-row tile, n in 17..32 (one key tile), n in
    33..47 (second key tile half present), n = 1 — and mlp = 64 (two-block MLP ring): modes 1 and 3 per parameter against the fp32
    oracle AND against the unfused bf16 kernels (mode 0)."""
    torch.manual_seed(D + n)
    tf = Transformer(D, 2, heads, 64, mlp)
    g = torch.Generator().manual_seed(n)
    with torch.no_grad():
        for p in tf.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    tf.compute_dtype = "bf16"
    tf = tf.to(DEV)
    x = (torch.randn(B, n, D, generator=g) * 1.5).to(DEV)
    cot = torch.randn(B, n, D, generator=g).to(DEV)
    P = {"t." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in tf.state_dict().items()}
    xo = x.cpu().clone().requires_grad_(True)
    yo = O.transformer(xo, P, "t.", 2, heads, 64)
    (yo * cot.cpu()).sum().backward()
    with _Residual(rb):          # (rb = 1 engages where both halves of every layer are block kernels: mode 3)
        res = {m: _run(tf, x, cot, m) for m in (0, 1, 3)}
    by, bdx, bg = _BOUNDS[rb]
    for m in (0, 1, 3):
        y, dx, grads = res[m]
        ey, edx = _relmax(y, yo.detach()), _relmax(dx, xo.grad)
        worst = max(((k, _relmax(gr, P["t." + k].grad)) for k, gr in grads.items()), key=lambda t: t[1])
        print(f"\n[edge] D={D} mlp={mlp} n={n} mode {m} rb {rb} vs oracle: y {ey:.2e} dx {edx:.2e} worst {worst[0]} {worst[1]:.2e}")
        assert ey <= by and edx <= bdx and worst[1] <= bg, (m, ey, edx, worst)
    for m in (1, 3):
        y, dx, grads = res[m]
        y0, dx0, g0 = res[0]
        ey, edx = _relmax(y, y0), _relmax(dx, dx0)
        worst = max(((k, _relmax(gr, g0[k])) for k, gr in grads.items()), key=lambda t: t[1])
        print(f"\n[edge] D={D} mlp={mlp} n={n} mode {m} rb {rb} vs unfused: y {ey:.2e} dx {edx:.2e} worst {worst[0]} {worst[1]:.2e}")
        assert ey <= by and edx <= bdx and worst[1] <= bg, (m, ey, edx, worst)


T192_CASES = [(192, 3, m, n, b) for m, n, b in [(768, 192, 3), (768, 100, 3), (96, 64, 5), (384, 200, 2), (768, 48, 70), (768, 48, 130), (256, 20, 40),
                                                (768, 192, 210), (768, 452, 64), (768, 100, 250), (768, 192, 129)]]
# widths 256 (M3L's default architecture: 256 / 4 heads, train.py:128-153) and 384 (ViT-Small: cfg 4's encoder 384 / 6 / 1536 at n = 113; cfg 5:
# 384 / 4 heads -> inner width 256 != D, so no out-proj prologue): 128- / 96-row tiles, ragged last tiles, per-sample attention at D = 256
T192_CASES += [(256, 4, 1024, 192, 3), (256, 4, 512, 10, 41), (256, 4, 1024, 192, 130), (256, 4, 1024, 100, 9), (384, 6, 1536, 113, 5),
               (384, 4, 768, 75, 20), (384, 6, 1536, 113, 64), (384, 4, 1536, 15, 30)]


@pytest.mark.parametrize("rb", [0, 1])
@pytest.mark.parametrize("D,heads,mlp,n,B", T192_CASES)
def test_t192_row_tiled_kernels(D, heads, mlp, n, B, rb):
    """Long sequences (n > 48: the MAE decoder) run their half layers as row tiles (t192.hip), whatever the sample boundaries: tiles
    that straddle samples, a ragged last tile (M = B n not a multiple of the tile height) and a narrow MLP — against the fp32 oracle and
    against the per-op bf16 kernels (M3L_T192 off), per parameter.  Modes: 3 = the library's own tile choice (D = 192: <3,2> 48-row tiles
    below 128 tiles of 192 rows, <12,1> from there up: (768, 452, 64) is BASELINE cfg 4's decoder at B = 64, M = 28 928 = 150.67 tiles;
    (768, 100, 250) is 130.2 tiles; both end in a ragged <12,1> tile.  D = 256 / 384: 128- / 96-row tiles from 1024 rows up), 7 = every
    row-tiled kernel (MLP tiles, out-proj prologue, dxn1 + LN1 backward, per-sample attention) forced at any M, so the small ragged
    shapes cover the tail handling too."""
    torch.manual_seed(n + mlp)
    tf = Transformer(D, 2, heads, 64, mlp)
    g = torch.Generator().manual_seed(n)
    with torch.no_grad():
        for p in tf.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    tf.compute_dtype = "bf16"
    tf = tf.to(DEV)
    x = (torch.randn(B, n, D, generator=g) * 1.5).to(DEV)
    cot = torch.randn(B, n, D, generator=g).to(DEV)
    P = {"t." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in tf.state_dict().items()}
    xo = x.cpu().clone().requires_grad_(True)
    yo = O.transformer(xo, P, "t.", 2, heads, 64)
    (yo * cot.cpu()).sum().backward()
    res = {}
    modes = (0, 3, 7) + ((13, 17) if D == 192 else ())         # 13 / 17 = modes 3 / 7 with 96-row tall tiles (two workgroups per CU)
    for on in modes:
        old = L.lib().m3l_set_t192(on % 10)
        old_tt = L.lib().m3l_set_t192_tt(6 if on >= 10 else 12)
        try:
            with _Residual(rb):   # (rb = 1 engages where every half layer runs row-tiled: per-sample attention + tiles, mode 7 at 48 < n <= 192)
                res[on] = _run(tf, x, cot, 1)
        finally:
            L.lib().m3l_set_t192(old)
            L.lib().m3l_set_t192_tt(old_tt)
    by, bdx, bg = _BOUNDS[rb]
    for on in modes:
        y, dx, grads = res[on]
        ey, edx = _relmax(y, yo.detach()), _relmax(dx, xo.grad)
        worst = max(((k, _relmax(gr, P["t." + k].grad)) for k, gr in grads.items()), key=lambda t: t[1])
        print(f"\n[t192] D={D} mlp={mlp} n={n} B={B} t192={on} rb {rb} vs oracle: y {ey:.2e} dx {edx:.2e} worst {worst[0]} {worst[1]:.2e}")
        assert ey <= by and edx <= bdx and worst[1] <= bg, (on, ey, edx, worst)
    y0, dx0, g0 = res[0]
    for on in modes[1:]:
        y, dx, grads = res[on]
        worst = max(((k, _relmax(gr, g0[k])) for k, gr in grads.items()), key=lambda t: t[1])
        print(f"\n[t192] D={D} mlp={mlp} n={n} B={B} t192={on} rb {rb} vs off: y {_relmax(y, y0):.2e} dx {_relmax(dx, dx0):.2e} worst {worst[0]} {worst[1]:.2e}")
        assert _relmax(y, y0) <= by and _relmax(dx, dx0) <= bdx and worst[1] <= bg, on


@pytest.mark.parametrize("D,heads,mlp,n,B,depth", [(192, 3, 768, 48, 5, 4), (128, 2, 256, 33, 3, 3), (192, 3, 384, 17, 2, 16), (192, 3, 768, 48, 7, 6)])
@pytest.mark.parametrize("rb", [0, 1])
def test_whole_stack_forward_launch_is_bit_identical(D, heads, mlp, n, B, depth, rb):
    """enc_mega.hip: the forward of a short-sequence stack as ONE launch and its backward as one launch per weight-gradient group of layers
    (the block kernels' bodies back to back; depth 6 and 3 leave a shorter last group) against one launch per half layer
    (m3l_set_enc_mega bits 1 / 2): outputs, input gradient and every parameter gradient bit-identical."""
    torch.manual_seed(D + n)
    tf = Transformer(D, depth, heads, 64, mlp)
    tf.compute_dtype = "bf16"
    tf = tf.to(DEV)
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(B, n, D, generator=g) * 1.5).to(DEV)
    cot = torch.randn(B, n, D, generator=g).to(DEV)
    res = {}
    megas = (0, 1, 2, 3) if rb == 0 else (0, 1)      # (the grouped backward launch, bit 2, runs the fp32 residual stream only)
    for mega in megas:
        old = L.lib().m3l_set_enc_mega(mega)
        try:
            with _Residual(rb):
                res[mega] = _run(tf, x, cot, 3)
        finally:
            L.lib().m3l_set_enc_mega(old)
    y0, dx0, g0 = res[0]
    for mega in megas[1:]:
        y1, dx1, g1 = res[mega]
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1), mega
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (mega, k)
