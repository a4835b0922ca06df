"""torch.autograd glue over the C ABI (include/m3l_amd.h).  PyTorch is plumbing here: it owns device memory, the
current HIP stream and the autograd tape; every number is produced by the HIP kernels in m3l_amd/csrc.

Each Function is one module of the reference's hot path (models/pretrain_models.py:146-342):
  EmbedFn        patchify + LayerNorm/Linear/LayerNorm + modality + sincos (+ visible gather)   :157-216,255-256
  TransformerFn  vit_pytorch.vit.Transformer.forward                                             :266,309
  UnshuffleFn    enc_to_dec + mask tokens + decoder modality/sincos                              :270-307
  HeadsLossFn    masked gather + to_pixels/to_tactiles + weighted MSE                            :260-262,327-340
"""
import contextlib
import ctypes as C

import torch

from . import _lib as L

DT_F32, DT_BF16 = 0, 1

# Layers per backward chunk of a transformer stack (None: one call).  With a GradSync on more than one rank the chunks let the
# all-reduce of finished layers travel while the remaining layers compute; tests set it to exercise the range entry point.
BWD_CHUNK_LAYERS = None


def dtype_code(compute_dtype) -> int:
    if compute_dtype in ("bf16", torch.bfloat16, 1):
        return DT_BF16
    if compute_dtype in ("fp32", "f32", torch.float32, 0):
        return DT_F32
    raise ValueError(f"unsupported compute dtype {compute_dtype!r} (use 'fp32' or 'bf16')")


def tdtype(code: int):
    return torch.bfloat16 if code == DT_BF16 else torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise L.M3LError(f"{what} must live on the GPU: m3l_amd runs only on its HIP kernels (no CPU fallback)")


def _f32c(t):
    return t.detach().contiguous().float() if t is not None else None


def _ws(nbytes: int, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def make_geom(enc, num_tactiles, use_vision=True, use_tactile=True) -> L.Geom:
    return L.Geom(enc.image_height, enc.image_width, enc.image_patch_height, enc.image_channels,
                  enc.tactile_height, enc.tactile_width, enc.tactile_patch_height, enc.tactile_channels,
                  int(num_tactiles), int(bool(use_vision)), int(bool(use_tactile)))


def mask_counts(geom: L.Geom, ratio: float):
    out = (C.c_int * 6)()
    L.check(L.lib().m3l_mask_counts(C.byref(geom), float(ratio), out), "m3l_mask_counts")
    return dict(num_masked=out[0], num_unmasked=out[1], nm_img=out[2], nm_tac=out[3], n_img=out[4], n_tac=out[5])


def mask_sample(geom: L.Geom, ratio: float, noises, counts=None):
    """noises: list of (B, n_i) f32 CUDA tensors in the reference's RNG order.  Returns int64 (masked, unmasked).
    counts = (nm_img, nm_tac) overrides the forward's integer rule (VTMAE.reconstruct has its own)."""
    c = mask_counts(geom, ratio)
    if counts is not None:
        k = geom.num_tactiles if geom.use_tactile else 0
        c = dict(c, nm_img=counts[0], nm_tac=counts[1], num_masked=counts[0] + k * counts[1])
        c["num_unmasked"] = c["n_img"] + k * c["n_tac"] - c["num_masked"]
    B = noises[0].shape[0]
    dev = noises[0].device
    noises = [_f32c(n) for n in noises]
    for n in noises:
        _require_cuda(n, "mask noise")
    masked = torch.empty(B, c["num_masked"], dtype=torch.int64, device=dev)
    unmasked = torch.empty(B, c["num_unmasked"], dtype=torch.int64, device=dev)
    L.check(L.lib().m3l_mask_sample_counts(C.byref(geom), c["nm_img"], c["nm_tac"], B, L.ptr_array(noises), L.ptr(masked),
                                           L.ptr(unmasked), _stream()), "m3l_mask_sample")
    return masked, unmasked, c


def _grad_targets(sink, params, used=None):
    """Where the native backward writes parameter gradients -> (targets, direct).
    sink None  -> fresh zero tensors, returned to autograd (which accumulates into .grad: reference semantics); direct = False.
    sink given -> (GradSync, bucket): the parameters' .grad are views into the sync's flat buffer; the kernels write them in place
                  (overwrite) and autograd gets None for them (no per-tensor accumulate / copy kernels); direct = True.
    A SECOND backward through the same parameters before GradSync.zero_grad() (the reference's `separate_optimizer=False` update,
    ppo_mae.py:249-266, runs mae_loss.backward() and then the policy loss through MAEExtractor -> get_embeddings; micro-batch
    accumulation does the same) must ADD: it takes the autograd route (fresh tensors, accumulated into the flat views by autograd).
    When the sync communicates, the first gradient has already been all-reduced, so that is an error instead of a silent divergence."""
    direct = sink is not None
    if direct:
        sync = sink[0]
        ids = [id(p) for i, p in enumerate(params)
               if p is not None and (used is None or used[i]) and p.is_leaf and p.grad is not None and p.requires_grad]
        if any(i in sync._written for i in ids):
            if sync._comm:
                raise RuntimeError("GradSync: second backward through the same parameters before zero_grad() while gradients are being "
                                   "all-reduced (the first ones have already travelled); call finish() + step() + zero_grad() per backward, "
                                   "or build the GradSync without communication for gradient accumulation")
            direct = False
            # the first backward's weight gradients may still be writing these very views on the library's side stream (deferred join):
            # autograd's accumulate into them, on this stream, must come after
            if sync._defer:
                L.check(L.lib().m3l_side_join(_stream()), "m3l_side_join")
        else:
            sync._written.update(ids)
    out = []
    for i, p in enumerate(params):
        ok = p is not None and (used is None or used[i])
        if not ok:
            out.append(None)
        elif direct and p.is_leaf and p.grad is not None and p.requires_grad:
            out.append(p.grad)
        else:
            out.append(torch.zeros_like(p))
    return out, direct


def _versions(tensors):
    return [None if t is None else t._version for t in tensors]


def _check_versions(ctx, what):
    """The Functions keep references to the live parameters and compute-type copies made in forward (not save_for_backward): an
    in-place update between forward and backward (optimizer.step(), load_state_dict) would silently mix old and new weights, so it is
    refused the way autograd refuses it."""
    for i, (t, v) in enumerate(zip(ctx.params, ctx.versions)):
        if t is not None and t._version != v:
            raise RuntimeError(f"{what}: parameter {i} was modified by an inplace operation between forward and backward "
                               f"(version {t._version}, expected {v})")


def _check_inputs(what, tensors, versions):
    """The backward re-reads the raw observation frames (patches for the patch-LayerNorm backward, the stems' input frames for the direct
    convolutions' weight gradient): a staging buffer refilled in place between forward and backward would silently give a wrong gradient,
    so it is refused like a modified parameter."""
    for i, (t, v) in enumerate(zip(tensors, versions)):
        if t is not None and t._version != v:
            raise RuntimeError(f"{what}: input {i} was modified by an inplace operation between forward and backward "
                               f"(version {t._version}, expected {v}); keep the observation batch unchanged until backward() has run")


def _returned(sink, grads):
    return tuple(None for _ in grads) if sink is not None else tuple(grads)


def _done(sink):
    if sink is not None and sink[1] is not None:      # bucket None: this Function only contributes part of a bucket
        sink[0].bucket_done(sink[1])


def _pos_grads(pos_img, pos_tac, dtokens, idx):
    """Gradient of the LEARNED position rows (use_sincosmod_encodings=False, pretrain_models.py:218-219,280-287): position j receives
    the sum over the batch of the token gradients that sit at position j.  (None, None) when the positions are fixed sincos buffers.
    dtokens (B, L, D) f32; idx (B, L) int64 positions of the L tokens, or None when all positions are present in order."""
    want_i = pos_img is not None and pos_img.requires_grad
    want_t = pos_tac is not None and pos_tac.requires_grad
    if not (want_i or want_t):
        return None, None
    n_img = pos_img.shape[0] if pos_img is not None else 0
    n_all = n_img + (pos_tac.shape[0] if pos_tac is not None else 0)
    if idx is None:
        dense = dtokens
    else:
        B, K, D = dtokens.shape
        dense = torch.zeros(B, n_all, D, dtype=torch.float32, device=dtokens.device)
        L.check(L.lib().m3l_scatter_tokens(L.ptr(dtokens), B, n_all, D, L.ptr(idx.contiguous()), K, L.ptr(dense), _stream()), "m3l_scatter_tokens")
    g = dense.sum(dim=0)
    return (g[:n_img] if want_i else None), (g[n_img:n_all] if want_t else None)


# -------------------------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def _deferred_join(sink, direct, *keep):
    """Direct-gradient mode without communication: the library may leave this call's weight-gradient kernels running on its side stream
    (m3l_set_defer_join); GradSync.finish() joins them and drops `keep`, the references that hold their operands alive until then."""
    defer = direct and sink[0]._defer
    if defer:
        sink[0]._keep.append(keep)
        L.lib().m3l_set_defer_join(1)
    try:
        yield
    finally:
        if defer:
            L.lib().m3l_set_defer_join(0)


class EmbedFn(torch.autograd.Function):
    """inputs: image / tactile tensors (no grad), then the 15 tensors of the embed group (see header)."""

    @staticmethod
    def forward(ctx, sink, geom, D, dt, idx, cnt_img, L_tok, image, tactiles, *tensors):
        ref = image if image is not None else tactiles[0]
        _require_cuda(ref, "MAE input")
        B, dev = ref.shape[0], ref.device
        image = _f32c(image)
        tactiles = [_f32c(t) for t in tactiles]
        tens = [_f32c(t) for t in tensors]
        ws = _ws(L.lib().m3l_embed_ws_bytes(C.byref(geom), D, dt, B, L_tok), dev)
        tokens = torch.empty(B, L_tok, D, dtype=torch.float32, device=dev)
        tac_arr = L.ptr_array(tactiles)
        L.check(L.lib().m3l_embed_fwd(C.byref(geom), D, dt, B, L_tok, cnt_img, L.ptr(idx), L.ptr(image), tac_arr,
                                      L.ptr_array(tens), L.ptr(ws), L.ptr(tokens), _stream()), "m3l_embed_fwd")
        ctx.saved = (geom, D, dt, idx, cnt_img, L_tok, image, tactiles, tens, ws)
        ctx.params, ctx.sink = tensors, sink
        ctx.versions = _versions(tensors)
        return tokens

    @staticmethod
    def backward(ctx, dtokens):
        _check_versions(ctx, "EmbedFn")
        geom, D, dt, idx, cnt_img, L_tok, image, tactiles, tens, ws = ctx.saved
        B = dtokens.shape[0]
        dtokens = _f32c(dtokens)
        # parameters of a modality that is absent from this call stay without gradient (as in the reference graph)
        used = [image is not None] * 6 + [len(tactiles) > 0] * 6 + [True, False, False]
        grads, direct = _grad_targets(ctx.sink, ctx.params, used)
        sink = ctx.sink if direct else None
        with _deferred_join(sink, direct, ws, grads, dtokens):
            L.check(L.lib().m3l_embed_bwd(C.byref(geom), D, dt, B, L_tok, cnt_img, L.ptr(idx), L.ptr(image),
                                          L.ptr_array(tactiles), L.ptr_array(tens), L.ptr(ws), L.ptr(dtokens),
                                          L.ptr_array(grads), _stream()), "m3l_embed_bwd")
        _done(sink)
        out = list(_returned(sink, grads))
        out[13], out[14] = _pos_grads(ctx.params[13], ctx.params[14], dtokens, idx)
        return (None,) * 9 + tuple(out)


class TransformerFn(torch.autograd.Function):
    """x (B, n, D) f32 -> (y_t compute-type, y32 f32) = final LayerNorm output, twice (the heads consume the
    compute-type copy, torch-side consumers the f32 one)."""

    @staticmethod
    def forward(ctx, sink, cfg, x, *tensors):
        _require_cuda(x, "transformer input")
        B, n, D = x.shape
        assert D == cfg.dim, (D, cfg.dim)
        x = _f32c(x)
        tens = [_f32c(t) for t in tensors]
        ws = _ws(L.lib().m3l_transformer_ws_bytes(C.byref(cfg), B, n), x.device)
        y32 = torch.empty(B, n, D, dtype=torch.float32, device=x.device)
        y_t = torch.empty(B, n, D, dtype=torch.bfloat16, device=x.device) if cfg.dtype == DT_BF16 else None
        L.check(L.lib().m3l_transformer_fwd(C.byref(cfg), B, n, L.ptr(x), L.ptr_array(tens), L.ptr(ws),
                                            L.ptr(y_t), L.ptr(y32), _stream()), "m3l_transformer_fwd")
        ctx.saved = (cfg, x, tens, ws)
        ctx.params, ctx.sink = tensors, sink
        ctx.versions = _versions(tensors)
        ctx.set_materialize_grads(False)     # an unused output must arrive as None in backward, not as a zero tensor to add and convert
        if y_t is None:
            y_t = y32.clone()     # f32 compute: two distinct autograd outputs over the same values
        return y_t, y32

    @staticmethod
    def backward(ctx, dy_t, dy32):
        _check_versions(ctx, "TransformerFn")
        cfg, x, tens, ws = ctx.saved
        B, n, D = x.shape
        if dy_t is None and dy32 is None:         # nothing downstream used this stack
            return (None, None, None) + (None,) * len(ctx.params)
        if dy_t is not None and dy32 is not None:
            dy, code = (dy32 + dy_t.float()).contiguous(), DT_F32
        elif dy_t is not None:
            dy = dy_t.contiguous()
            code = DT_BF16 if dy.dtype == torch.bfloat16 else DT_F32
        else:
            dy, code = _f32c(dy32), DT_F32
        grads, direct = _grad_targets(ctx.sink, ctx.params)
        dx = torch.empty_like(x)
        sink = ctx.sink if direct else None
        # direct mode without communication: the weight gradients still running on the library's side stream when the call returns
        # are joined by GradSync.finish(), which also drops the references that keep their operands alive
        defer = direct and sink[0]._defer
        if defer:
            sink[0]._keep.append((ws, grads, dy))
            L.lib().m3l_set_defer_join(1)
        try:
            chunk = BWD_CHUNK_LAYERS
            if chunk is None and sink is not None and sink[1] is not None and sink[0]._comm:
                chunk = sink[0].layers_per_chunk
            if not chunk or chunk >= cfg.depth:
                L.check(L.lib().m3l_transformer_bwd(C.byref(cfg), B, n, L.ptr(x), L.ptr_array(tens), L.ptr(ws), L.ptr(dy), code,
                                                    L.ptr(dx), L.ptr_array(grads), _stream()), "m3l_transformer_bwd")
                _done(sink)
            else:
                tens_a, grads_a = L.ptr_array(tens), L.ptr_array(grads)
                hi = cfg.depth
                while hi > 0:
                    lo = max(0, hi - chunk)
                    L.check(L.lib().m3l_transformer_bwd_range(C.byref(cfg), B, n, L.ptr(x), tens_a, L.ptr(ws), L.ptr(dy), code, L.ptr(dx),
                                                              grads_a, hi, lo, _stream()), "m3l_transformer_bwd_range")
                    if sink is not None and sink[1] is not None:
                        done = list(ctx.params[11 * lo:11 * hi]) + (list(ctx.params[11 * cfg.depth:]) if hi == cfg.depth else [])
                        sink[0].range_done(sink[1], done, last=(lo == 0))
                    hi = lo
        finally:
            if defer:
                L.lib().m3l_set_defer_join(0)
        return (None, None, dx) + _returned(sink, grads)


class UnshuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sink, geom, D, dd, dt, unmasked, masked, enc_t, enc32, *tensors):
        B, nvis = unmasked.shape
        nmask = masked.shape[1]
        dev = enc32.device
        tens = [_f32c(t) for t in tensors]
        ws = _ws(L.lib().m3l_unshuffle_ws_bytes(C.byref(geom), D, dd, dt, B, nvis, nmask), dev)
        dec_in = torch.empty(B, nvis + nmask, dd, dtype=torch.float32, device=dev)
        enc_t = enc_t.contiguous()
        enc32 = enc32.contiguous()
        L.check(L.lib().m3l_unshuffle_fwd(C.byref(geom), D, dd, dt, B, nvis, nmask, L.ptr(unmasked), L.ptr(masked),
                                          L.ptr(enc32), L.ptr(enc_t), L.ptr_array(tens), L.ptr(ws), L.ptr(dec_in),
                                          _stream()), "m3l_unshuffle_fwd")
        ctx.saved = (geom, D, dd, dt, unmasked, masked, enc_t, tens, ws)
        ctx.params, ctx.sink = tensors, sink
        ctx.versions = _versions(tensors)
        return dec_in

    @staticmethod
    def backward(ctx, d_dec_in):
        _check_versions(ctx, "UnshuffleFn")
        geom, D, dd, dt, unmasked, masked, enc_t, tens, ws = ctx.saved
        B, nvis = unmasked.shape
        nmask = masked.shape[1]
        dev = d_dec_in.device
        d_dec_in = _f32c(d_dec_in)
        proj = tens[0] is not None
        d_enc = torch.empty(B, nvis, D, dtype=(tdtype(dt) if proj else torch.float32), device=dev)
        grads, direct = _grad_targets(ctx.sink, ctx.params, [True, True, True, True, False, False])
        sink = ctx.sink if direct else None
        code = C.c_int(0)
        L.check(L.lib().m3l_unshuffle_bwd(C.byref(geom), D, dd, dt, B, nvis, nmask, L.ptr(unmasked), L.ptr(masked),
                                          L.ptr(enc_t), L.ptr_array(tens), L.ptr(ws), L.ptr(d_dec_in), L.ptr(d_enc),
                                          C.byref(code), L.ptr_array(grads), _stream()), "m3l_unshuffle_bwd")
        _done(sink)
        out = list(_returned(sink, grads))
        out[4], out[5] = _pos_grads(ctx.params[4], ctx.params[5], d_dec_in, None)      # learned decoder positions (decoder_pos_emb)
        if proj:      # gradient flows through the compute-type encoder output
            return (None,) * 7 + (d_enc, None) + tuple(out)
        return (None,) * 7 + (None, d_enc) + tuple(out)


class HeadsLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sink, geom, dd, dt, masked, nm_img, image, tactiles, dump, dec_t, *tensors):
        B, N, _ = dec_t.shape
        nmask = masked.shape[1]
        dev = dec_t.device
        image = _f32c(image)
        tactiles = [_f32c(t) for t in tactiles]
        tens = [_f32c(t) for t in tensors]
        dec_t = dec_t.contiguous()
        ws = _ws(L.lib().m3l_heads_ws_bytes(C.byref(geom), dd, dt, B, nmask), dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        outs = [None] * 4
        if dump is not None:
            pd_i = geom.image_channels * geom.image_patch ** 2
            pd_t = geom.tactile_channels * geom.tactile_patch ** 2
            n_t = nmask - nm_img
            outs = [torch.zeros(B, nm_img, pd_i, device=dev), torch.zeros(B, nm_img, pd_i, device=dev),
                    torch.zeros(B, n_t, pd_t, device=dev), torch.zeros(B, n_t, pd_t, device=dev)]
            dump.update(pred_pixel=outs[0], target_pixel=outs[1], pred_tactile=outs[2], target_tactile=outs[3])
        parts = None
        if dump is not None:
            parts = torch.zeros(2, dtype=torch.float32, device=dev)
            dump["loss_parts"] = parts
        L.check(L.lib().m3l_heads_loss_fwd2(C.byref(geom), dd, dt, B, N, nmask, nm_img, L.ptr(masked), L.ptr(image),
                                            L.ptr_array(tactiles), L.ptr(dec_t), L.ptr_array(tens), L.ptr(ws), L.ptr(loss), L.ptr(parts),
                                            L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(outs[3]), _stream()),
                "m3l_heads_loss_fwd")
        ctx.saved = (geom, dd, dt, masked, nm_img, tens, ws, (B, N), dec_t.dtype)
        ctx.used = [image is not None] * 2 + [len(tactiles) > 0] * 2
        ctx.params, ctx.sink = tensors, sink
        ctx.versions = _versions(tensors)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        _check_versions(ctx, "HeadsLossFn")
        geom, dd, dt, masked, nm_img, tens, ws, (B, N), ddtype = ctx.saved
        nmask = masked.shape[1]
        dloss = _f32c(dloss)
        d_dec = torch.empty(B, N, dd, dtype=ddtype, device=dloss.device)
        grads, direct = _grad_targets(ctx.sink, ctx.params, ctx.used)
        sink = ctx.sink if direct else None
        with _deferred_join(sink, direct, ws, grads, dloss):
            L.check(L.lib().m3l_heads_loss_bwd(C.byref(geom), dd, dt, B, N, nmask, nm_img, L.ptr(masked), L.ptr_array(tens),
                                               L.ptr(ws), L.ptr(dloss), L.ptr(d_dec), L.ptr_array(grads), _stream()),
                    "m3l_heads_loss_bwd")
        _done(sink)
        return (None,) * 9 + (d_dec,) + _returned(sink, grads)


# -------------------------------------------------------------------------------------------------------------------
# The whole MAE step as ONE autograd node over two C calls (csrc/mae_step.hip): the module chain above runs inside the library.
FUSED_STEP = True     # tests switch it off to compare against the per-module path (results are bit-identical)


def _dev_ptrs(tensors, keep):
    """void*[] of f32 device tensors; a tensor that is not f32-contiguous is converted and parked in `keep` (alive until backward)."""
    arr = (L.c_p * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        if t is None:
            continue
        if t.dtype is not torch.float32 or not t.is_contiguous():
            t = t.detach().contiguous().float()
            keep.append(t)
        arr[i] = t.data_ptr()
    return arr


class StepPlan:
    """Everything one fused step needs, assembled by VTMAE._step_fused: cfg (L.MaeCfg), the five tensor groups as one list, which of
    them take a gradient in this call, the inputs and the GradSync (or None)."""
    __slots__ = ("cfg", "tensors", "used", "image", "tactiles", "noises", "sync", "B", "nmask", "nvis", "ws", "keep", "versions", "tens_arr",
                 "tac_arr", "masked", "unmasked", "extra", "head_cfg", "in_versions")


def _comm_plan(sync, plan, cfg):
    """m3l_comm_plan for this model on `sync` (cached: the layout never changes): stage ends in the flat gradient buffer."""
    key = (cfg.enc.depth, cfg.dec.depth, sync.layers_per_chunk, sync.min_bucket_elems, cfg.early_conv, cfg.learned_pos)
    cached = getattr(sync, "_step_plan", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    groups = [19 if cfg.early_conv else 15, 11 * cfg.enc.depth + 2, 6, 11 * cfg.dec.depth + 2, 4]
    off = [0]
    for n in groups:
        off.append(off[-1] + n)
    T = plan.tensors

    def end_of(ts):
        spans = [sync._span[id(t)] for t in ts if t is not None and id(t) in sync._span]
        return max(b for _, b in spans) if spans else 0

    def tf_ends(base, depth):
        chunk = sync.layers_per_chunk
        fin = T[base + 11 * depth: base + 11 * depth + 2]
        if not chunk or chunk >= depth:
            return [end_of(T[base: base + 11 * depth + 2])]
        ends, hi = [], depth
        while hi > 0:
            lo = max(0, hi - chunk)
            ends.append(end_of(T[base + 11 * lo: base + 11 * hi] + (fin if hi == depth else [])))
            hi = lo
        return ends
    ends = [end_of(T[off[4]:off[5]])] + tf_ends(off[3], cfg.dec.depth) + [end_of(T[off[2]:off[3]])] + tf_ends(off[1], cfg.enc.depth) + [end_of(T[off[0]:off[1]])]
    run = 0
    for i, e in enumerate(ends):          # a stage with no parameters of its own (Identity enc_to_dec ...) keeps the previous prefix
        run = max(run, e)
        ends[i] = run
    if not cfg.learned_pos:               # whatever is left (nothing, in the layouts GradSync builds) goes with the last stage; learned position
        ends[-1] = sync.flat.numel()      # tables own the trailing span, final only after autograd has accumulated them: finish() sends it
    arr = (C.c_long * len(ends))(*ends)
    sent = C.c_long(0)
    cp = L.CommPlan(sync.flat.data_ptr(), sync.flat.numel(), sync.min_bucket_elems, int(sync.layers_per_chunk or 0), len(ends), arr, C.pointer(sent))
    sync._step_plan = (key, (cp, arr, sent))
    return sync._step_plan[1]


class MaeStepFn(torch.autograd.Function):
    """loss = VTMAE.forward(x) (models/pretrain_models.py:146-342) through m3l_mae_step_fwd / m3l_mae_step_bwd.  `tensors` are the
    autograd inputs: every parameter of the step when autograd has to receive the gradients (no GradSync), or a single anchor parameter
    when the kernels write the gradients straight into the GradSync's flat buffer."""

    @staticmethod
    def forward(ctx, plan, *tensors):
        lib = L.lib()
        dev = plan.image.device if plan.image is not None else plan.tactiles[0].device
        cfg = plan.cfg
        plan.keep = []
        plan.ws = _ws(lib.m3l_mae_step_ws_bytes(C.byref(cfg), plan.B), dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        plan.masked = torch.empty(plan.B, plan.nmask, dtype=torch.int64, device=dev)
        plan.unmasked = torch.empty(plan.B, plan.nvis, dtype=torch.int64, device=dev)
        plan.tens_arr = _dev_ptrs(plan.tensors, plan.keep)
        plan.tac_arr = L.ptr_array(plan.tactiles)
        L.check(lib.m3l_mae_step_fwd(C.byref(cfg), plan.B, L.ptr(plan.image), plan.tac_arr, L.ptr_array(plan.noises), plan.tens_arr,
                                     L.ptr(plan.ws), L.ptr(loss), L.ptr(plan.masked), L.ptr(plan.unmasked), _stream()), "m3l_mae_step_fwd")
        plan.versions = _versions(plan.tensors)
        plan.in_versions = _versions([plan.image] + list(plan.tactiles))
        plan.noises = None
        ctx.plan = plan
        ctx.n_in = len(tensors)
        ctx.anchor_mode = plan.extra is not None
        return loss

    @staticmethod
    def backward(ctx, dloss):
        plan = ctx.plan
        lib = L.lib()
        for i, (t, v) in enumerate(zip(plan.tensors, plan.versions)):
            if t is not None and t._version != v:
                raise RuntimeError(f"MaeStepFn: parameter {i} was modified by an inplace operation between forward and backward "
                                   f"(version {t._version}, expected {v})")
        _check_inputs("MaeStepFn", [plan.image] + list(plan.tactiles), plan.in_versions)
        sync = plan.sync
        grads, direct = _grad_targets((sync, None) if sync is not None else None, plan.tensors, plan.used)
        dloss = _f32c(dloss)
        comm = None
        if direct and sync._comm:
            comm = _comm_plan(sync, plan, plan.cfg)
            comm[2].value = 0
        defer = direct and sync._defer
        if defer:
            sync._keep.append((plan.ws, plan.keep, grads, dloss, plan.image, plan.tactiles))
            lib.m3l_set_defer_join(1)
        try:
            L.check(lib.m3l_mae_step_bwd(C.byref(plan.cfg), plan.B, L.ptr(plan.image), plan.tac_arr, L.ptr(plan.masked), L.ptr(plan.unmasked), plan.tens_arr,
                                         L.ptr(plan.ws), L.ptr(dloss),
                                         L.ptr_array(grads), C.byref(comm[0]) if comm is not None else None, _stream()), "m3l_mae_step_bwd")
        finally:
            if defer:
                lib.m3l_set_defer_join(0)
        # anchor mode: autograd sees the anchor parameter and the tensors whose gradient must travel through autograd (`plan.extra`: the
        # slices of the learned position tables — non-leaf views, their gradient reaches the parameter through the slice's backward)
        extras = tuple(grads[i] for i in plan.extra) if ctx.anchor_mode else ()
        if direct:
            sync._reduced.update(sync._bucket_ids)
            if comm is not None:
                sync._sent_end = int(comm[2].value)
                sync._unscaled = True
            return (None, None) + extras
        if ctx.anchor_mode:
            # anchor mode, second backward before zero_grad() (gradient accumulation / the reference's separate_optimizer=False update):
            # autograd sees one input, so the fresh gradients are added into the flat views here (_grad_targets has joined the side stream)
            dst, src = [], []
            for i, (g, t) in enumerate(zip(grads, plan.tensors)):
                if g is None or t is None or not t.requires_grad or i in plan.extra:
                    continue
                if t.grad is None:
                    t.grad = g
                else:
                    dst.append(t.grad)
                    src.append(g)
            if dst:
                torch._foreach_add_(dst, src)
            return (None, None) + extras
        return (None,) + tuple(g for g, t in zip(grads, plan.tensors) if t is not None)      # the autograd inputs = the non-None tensors


FUSED_EXTRACTOR = True     # tests switch it off to compare against the per-module path


class ExtractorFn(torch.autograd.Function):
    """MAEExtractor.forward (models/pretrain_models.py:819-841) after vt_load — get_embeddings over all tokens -> the extractor's 1-layer
    Transformer -> mean over tokens — through m3l_extractor_fwd / m3l_extractor_bwd: one autograd node, two host calls (one under
    no_grad: the rollout path at B = number of envs).  `tensors`: every tensor of the chain (front | encoder | head groups); gradients
    go back through autograd (they accumulate: the policy loss reaches the MAE after the MAE loss, ppo_mae.py:249-280)."""

    @staticmethod
    def forward(ctx, plan, *tensors):
        lib = L.lib()
        dev = plan.image.device if plan.image is not None else plan.tactiles[0].device
        plan.keep = []
        plan.ws = _ws(lib.m3l_extractor_ws_bytes(C.byref(plan.cfg), C.byref(plan.head_cfg), plan.B), dev)
        out = torch.empty(plan.B, plan.cfg.enc.dim, dtype=torch.float32, device=dev)
        plan.tens_arr = _dev_ptrs(plan.tensors, plan.keep)
        plan.tac_arr = L.ptr_array(plan.tactiles)
        L.check(lib.m3l_extractor_fwd(C.byref(plan.cfg), C.byref(plan.head_cfg), plan.B, L.ptr(plan.image), plan.tac_arr, plan.tens_arr, L.ptr(plan.ws),
                                      L.ptr(out), _stream()), "m3l_extractor_fwd")
        plan.versions = _versions(plan.tensors)
        plan.in_versions = _versions([plan.image] + list(plan.tactiles))
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        for i, (t, v) in enumerate(zip(plan.tensors, plan.versions)):
            if t is not None and t._version != v:
                raise RuntimeError(f"ExtractorFn: parameter {i} was modified by an inplace operation between forward and backward "
                                   f"(version {t._version}, expected {v})")
        _check_inputs("ExtractorFn", [plan.image] + list(plan.tactiles), plan.in_versions)
        grads = [torch.zeros_like(t) if (t is not None and u) else None for t, u in zip(plan.tensors, plan.used)]
        dout = _f32c(dout)
        L.check(L.lib().m3l_extractor_bwd(C.byref(plan.cfg), C.byref(plan.head_cfg), plan.B, L.ptr(plan.image), plan.tac_arr, plan.tens_arr, L.ptr(plan.ws),
                                          L.ptr(dout), L.ptr_array(grads), _stream()), "m3l_extractor_bwd")
        return (None,) + tuple(g for g, t in zip(grads, plan.tensors) if t is not None)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of an f32 tensor (final norm of the DINO-style encoder, models/VTT.py:354)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _require_cuda(x, "layernorm input")
        x = _f32c(x)
        D = x.shape[-1]
        M = x.numel() // D
        y = torch.empty_like(x)
        L.check(L.lib().m3l_layernorm_fwd(DT_F32, L.ptr(x), M, D, L.ptr(_f32c(gamma)), L.ptr(_f32c(beta)), float(eps),
                                          L.ptr(y), None, _stream()), "m3l_layernorm_fwd")
        ctx.saved = (x, _f32c(gamma), float(eps))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, eps = ctx.saved
        D = x.shape[-1]
        M = x.numel() // D
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
        ws = _ws(L.lib().m3l_layernorm_ws_bytes(D), x.device)
        L.check(L.lib().m3l_layernorm_bwd(DT_F32, L.ptr(dy), L.ptr(x), M, D, L.ptr(gamma), eps, None, L.ptr(dx), L.ptr(ws),
                                          L.ptr(dg), L.ptr(db), _stream()), "m3l_layernorm_bwd")
        return dx, dg, db, None


class GatherTokensFn(torch.autograd.Function):
    """x (B, N, D) f32, idx (B, K) int64 -> (B, K, D)   [tactile_ssl.utils.apply_masks for one mask]."""

    @staticmethod
    def forward(ctx, x, idx):
        _require_cuda(x, "gather input")
        x = _f32c(x)
        idx = idx.contiguous()
        B, N, D = x.shape
        K = idx.shape[1]
        y = torch.empty(B, K, D, dtype=torch.float32, device=x.device)
        L.check(L.lib().m3l_gather_tokens(L.ptr(x), B, N, D, L.ptr(idx), K, L.ptr(y), _stream()), "m3l_gather_tokens")
        ctx.saved = (idx, N)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, N = ctx.saved
        dy = _f32c(dy)
        B, K, D = dy.shape
        dx = torch.zeros(B, N, D, dtype=torch.float32, device=dy.device)
        L.check(L.lib().m3l_scatter_tokens(L.ptr(dy), B, N, D, L.ptr(idx), K, L.ptr(dx), _stream()), "m3l_scatter_tokens")
        return dx, None


class EarlyCnnFn(torch.autograd.Function):
    """EarlyCNN stem (pretrain_models.py:37-56) over a list of NCHW inputs that share the stem's weights (processed as one
    batch): -> (len(srcs) * B, h * w, dim) f32 tokens, source-major.  tensors: conv1.w, conv1.b, ..., conv4.w, conv4.b."""

    @staticmethod
    def forward(ctx, sink, cfg, srcs, *tensors):
        _require_cuda(srcs[0], "MAE input")
        B, dev = srcs[0].shape[0], srcs[0].device
        srcs = [_f32c(t) for t in srcs]
        tens = [_f32c(t) for t in tensors]
        nsrc = len(srcs)
        ws = _ws(L.lib().m3l_earlycnn_ws_bytes(C.byref(cfg), B, nsrc), dev)
        down = 4 if cfg.tactile else 8
        hw = (cfg.height // down) * (cfg.width // down)
        out = torch.empty(nsrc * B, hw, cfg.dim, dtype=torch.float32, device=dev)
        L.check(L.lib().m3l_earlycnn_fwd(C.byref(cfg), B, nsrc, L.ptr_array(srcs), L.ptr_array(tens), L.ptr(ws), L.ptr(out),
                                         _stream()), "m3l_earlycnn_fwd")
        ctx.saved = (cfg, B, nsrc, srcs, tens, ws)
        ctx.params, ctx.sink = tensors, sink
        ctx.versions = _versions(tensors)
        ctx.in_versions = _versions(srcs)
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg, B, nsrc, srcs, tens, ws = ctx.saved
        _check_inputs("EarlyCnnFn", srcs, ctx.in_versions)
        dout = _f32c(dout)
        grads, direct = _grad_targets(ctx.sink, ctx.params)
        sink = ctx.sink if direct else None
        L.check(L.lib().m3l_earlycnn_bwd(C.byref(cfg), B, nsrc, L.ptr_array(srcs), L.ptr_array(tens), L.ptr(ws), L.ptr(dout),
                                         L.ptr_array(grads), _stream()), "m3l_earlycnn_bwd")
        _done(sink)
        return (None, None, None) + _returned(sink, grads)


class TokensAssembleFn(torch.autograd.Function):
    """stem tokens (+ modality embedding + sincos) -> (B, N, D) encoder tokens (pretrain_models.py:202-216)."""

    @staticmethod
    def forward(ctx, sink, geom, D, img_tok, tac_tok, mod, pos_img, pos_tac):
        ref = img_tok if img_tok is not None else tac_tok
        dev = ref.device
        c = mask_counts(geom, 0.5)
        N = c["num_masked"] + c["num_unmasked"]
        B = img_tok.shape[0] if img_tok is not None else tac_tok.shape[0] // geom.num_tactiles
        img_tok, tac_tok = _f32c(img_tok), _f32c(tac_tok)
        tens = [_f32c(mod), _f32c(pos_img), _f32c(pos_tac)]
        tokens = torch.empty(B, N, D, dtype=torch.float32, device=dev)
        L.check(L.lib().m3l_tokens_assemble_fwd(C.byref(geom), D, B, L.ptr(img_tok), L.ptr(tac_tok), L.ptr_array(tens),
                                                L.ptr(tokens), _stream()), "m3l_tokens_assemble_fwd")
        ctx.saved = (geom, D, B, None if img_tok is None else img_tok.shape, None if tac_tok is None else tac_tok.shape)
        ctx.params, ctx.sink = (mod,), sink
        ctx.pos = (pos_img, pos_tac)
        return tokens

    @staticmethod
    def backward(ctx, dtok):
        geom, D, B, ishape, tshape = ctx.saved
        dtok = _f32c(dtok)
        dev = dtok.device
        d_img = torch.empty(ishape, dtype=torch.float32, device=dev) if ishape is not None else None
        d_tac = torch.empty(tshape, dtype=torch.float32, device=dev) if tshape is not None else None
        (gmod,), direct = _grad_targets(ctx.sink, ctx.params)
        sink = ctx.sink if direct else None
        gmod.zero_()                                  # rows of sensors absent from this call keep a zero gradient
        ws = _ws(L.lib().m3l_tokens_assemble_ws_bytes(C.byref(geom), D), dev)
        L.check(L.lib().m3l_tokens_assemble_bwd(C.byref(geom), D, B, L.ptr(dtok), L.ptr(d_img), L.ptr(d_tac), L.ptr(ws), L.ptr(gmod),
                                                _stream()), "m3l_tokens_assemble_bwd")
        _done(sink)
        return (None, None, None, d_img, d_tac) + _returned(sink, [gmod]) + _pos_grads(ctx.pos[0], ctx.pos[1], dtok, None)


# -------------------------------------------------------------------------------------------------------------------
# Trainable fusion MLP of the cfg-5 extractor (models/pretrain_models_dino_cat_mae.py:828-836,899-903) on the HIP GEMMs.  f32 compute
# (exact f32 MFMA): the matrices are (B, 2 D) x (2 D, 2 D) — a few MFLOP — and f32 keeps the policy head's numerics those of nn.Linear.
class LinearActFn(torch.autograd.Function):
    """y = Dropout(ReLU(x W^T + b)) with optional ReLU / keep-mask.  x (M, K) f32, W (N, K), b (N); mask (M, N) uint8 or None, scale =
    1 / (1 - p).  Forward: one NT GEMM with bias (+ ReLU) epilogue (+ mask kernel).  Backward: dz = dy * scale * [y > 0] (the saved output
    carries both the ReLU and the dropout pattern), dW = dz^T x (TN GEMM), db = column sums, dx = dz W (NT GEMM on a transposed copy)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, mask, scale):
        _require_cuda(x, "fusion MLP input")
        lib = L.lib()
        x, w, b = _f32c(x), _f32c(weight), _f32c(bias)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        L.check(lib.m3l_op_gemm_nt(DT_F32, L.ptr(x), K, L.ptr(w), K, M, N, K, L.ptr(b), None, L.ptr(y), None, None, None, 2 if relu else 0, N,
                                   _stream()), "m3l_op_gemm_nt (fusion MLP)")
        if mask is not None:
            L.check(lib.m3l_op_mask_scale(0, L.ptr(y), L.ptr(mask), float(scale), y.numel(), L.ptr(y), _stream()), "m3l_op_mask_scale")
        ctx.saved = (x, w, y if (relu or mask is not None) else None, relu, mask is not None, float(scale) if mask is not None else 1.0)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x, w, y, relu, dropped, scale = ctx.saved
        M, K = x.shape
        N = w.shape[0]
        dz = _f32c(dy)
        if y is not None:
            if relu:          # ReLU (and the dropout pattern, where there is one): on where the saved output is > 0
                out = torch.empty_like(dz)
                L.check(lib.m3l_op_mask_scale(1, L.ptr(dz), L.ptr(y), scale, dz.numel(), L.ptr(out), _stream()), "m3l_op_mask_scale")
                dz = out
            else:
                raise NotImplementedError("Dropout without ReLU in front is not part of the reference's fusion MLP")
        dev = x.device
        dW = torch.empty(N, K, dtype=torch.float32, device=dev)
        wsb = lib.m3l_op_gemm_tn_ws_bytes(M, N, K)
        ws = _ws(wsb, dev)
        L.check(lib.m3l_op_gemm_tn(DT_F32, L.ptr(dz), N, L.ptr(x), K, M, N, K, L.ptr(ws), wsb, L.ptr(dW), K, _stream()), "m3l_op_gemm_tn (fusion MLP)")
        db = torch.empty(N, dtype=torch.float32, device=dev)
        ws2 = _ws(lib.m3l_op_colsum_ws_bytes(N), dev)
        L.check(lib.m3l_op_colsum(DT_F32, L.ptr(dz), M, N, N, L.ptr(ws2), L.ptr(db), _stream()), "m3l_op_colsum")
        dx = None
        if ctx.needs_input_grad[0]:
            wT = torch.empty(K, N, dtype=torch.float32, device=dev)
            L.check(lib.m3l_op_prep_weight(DT_F32, L.ptr(w), N, K, None, L.ptr(wT), _stream()), "m3l_op_prep_weight")
            dx = torch.empty(M, K, dtype=torch.float32, device=dev)
            L.check(lib.m3l_op_gemm_nt(DT_F32, L.ptr(dz), N, L.ptr(wT), N, M, K, N, None, None, L.ptr(dx), None, None, None, 0, K, _stream()),
                    "m3l_op_gemm_nt (fusion MLP dgrad)")
        return dx, dW, db, None, None, None


class Concat2Fn(torch.autograd.Function):
    """torch.cat((a, b), dim=-1) for two (rows, n) f32 matrices; b may carry no gradient (the frozen DINOv2 feature)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        rows, na, nb = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty(rows, na + nb, dtype=torch.float32, device=a.device)
        L.check(L.lib().m3l_op_concat2(L.ptr(a), na, L.ptr(b), nb, rows, L.ptr(out), 0, _stream()), "m3l_op_concat2")
        ctx.dims = (rows, na, nb)
        return out

    @staticmethod
    def backward(ctx, d):
        rows, na, nb = ctx.dims
        d = _f32c(d)
        da = torch.empty(rows, na, dtype=torch.float32, device=d.device)
        db = torch.empty(rows, nb, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[1] else None
        L.check(L.lib().m3l_op_concat2(L.ptr(da), na, L.ptr(db), nb, rows, L.ptr(d), 1, _stream()), "m3l_op_concat2 (split)")
        return da, db
