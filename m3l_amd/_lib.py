"""ctypes binding of the C ABI in include/m3l_amd.h (libm3l_amd.so, hand-written HIP for gfx950).

The product path has NO CPU / PyTorch fallback: if the shared library is missing this module raises at first use.
Build it with `python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3L_LIB_PATH: another build of the same library (A/B of two builds in one job: tools/ab_libs.sh); default = the in-tree library
LIB_PATH = os.environ.get("M3L_LIB_PATH") or os.path.join(_HERE, "lib", "libm3l_amd.so")

c_p = C.c_void_p
c_i = C.c_int
c_sz = C.c_size_t


class Geom(C.Structure):
    _fields_ = [(n, c_i) for n in ("image_h", "image_w", "image_patch", "image_channels", "tactile_h", "tactile_w",
                                   "tactile_patch", "tactile_channels", "num_tactiles", "use_vision", "use_tactile")]


class CnnCfg(C.Structure):
    _fields_ = [(n, c_i) for n in ("in_channels", "height", "width", "dim", "tactile", "dtype")]


class TfCfg(C.Structure):
    _fields_ = [(n, c_i) for n in ("dim", "depth", "heads", "mlp_dim", "project_out", "dtype")]


class MaeCfg(C.Structure):
    _fields_ = [("geom", Geom), ("enc", TfCfg), ("dec", TfCfg), ("masking_ratio", C.c_double), ("early_conv", c_i), ("learned_pos", c_i)]


class CommPlan(C.Structure):
    _fields_ = [("flat", c_p), ("total", C.c_long), ("min_bucket", C.c_long), ("layers_per_chunk", c_i), ("n_stages", c_i),
                ("stage_end", C.POINTER(C.c_long)), ("sent_out", C.POINTER(C.c_long))]


_SIGS = {
    "m3l_version": (c_i, []),
    "m3l_last_error": (c_i, [C.c_char_p, c_sz]),
    "m3l_set_rowln": (c_i, [c_i]),
    "m3l_set_attn_block": (c_i, [c_i]),
    "m3l_set_attn_phase_buffer": (None, [c_p]),
    "m3l_set_enc_mega": (c_i, [c_i]),
    "m3l_set_drop_h": (c_i, [c_i]),
    "m3l_set_direct_conv": (c_i, [c_i]),
    "m3l_set_t192": (c_i, [c_i]),
    "m3l_set_t192_tt": (c_i, [c_i]),
    "m3l_set_t192_stagger": (c_i, [c_i, c_i]),
    "m3l_set_residual_bf16": (c_i, [c_i]),
    "m3l_set_defer_join": (c_i, [c_i]),
    "m3l_set_wgrad_inline": (c_i, [c_i]),
    "m3l_side_join": (c_i, [c_p]),
    "m3l_side_stream": (c_p, []),
    "m3l_side_fork": (c_i, [c_p, C.POINTER(c_p)]),
    "m3l_side_mark_pending": (c_i, []),
    "m3l_comm_available": (c_i, []),
    "m3l_comm_unique_id": (c_i, [c_p]),
    "m3l_comm_init": (c_i, [c_p, c_i, c_i]),
    "m3l_comm_world": (c_i, []),
    "m3l_comm_allreduce": (c_i, [c_p, c_sz, c_p]),
    "m3l_comm_destroy": (c_i, []),
    "m3l_side_pending": (c_i, []),
    "m3l_mask_counts": (c_i, [C.POINTER(Geom), C.c_double, C.POINTER(c_i)]),
    "m3l_mask_sample": (c_i, [C.POINTER(Geom), C.c_double, c_i, c_p, c_p, c_p, c_p]),
    "m3l_mask_sample_counts": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "m3l_embed_ws_bytes": (c_sz, [C.POINTER(Geom), c_i, c_i, c_i, c_i]),
    "m3l_embed_fwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_embed_bwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_transformer_ws_bytes": (c_sz, [C.POINTER(TfCfg), c_i, c_i]),
    "m3l_transformer_fwd": (c_i, [C.POINTER(TfCfg), c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_transformer_bwd": (c_i, [C.POINTER(TfCfg), c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "m3l_transformer_bwd_range": (c_i, [C.POINTER(TfCfg), c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p]),
    "m3l_frozen_vit_ws_bytes": (c_sz, [C.POINTER(TfCfg), c_i, c_i]),
    "m3l_frozen_vit_fwd": (c_i, [C.POINTER(TfCfg), C.c_float, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "m3l_unshuffle_ws_bytes": (c_sz, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_i]),
    "m3l_unshuffle_fwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_unshuffle_bwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                C.POINTER(c_i), c_p, c_p]),
    "m3l_heads_ws_bytes": (c_sz, [C.POINTER(Geom), c_i, c_i, c_i, c_i]),
    "m3l_heads_loss_fwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                 c_p, c_p, c_p]),
    "m3l_heads_loss_fwd2": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                  c_p, c_p, c_p, c_p]),
    "m3l_heads_loss_bwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_mae_step_num_tensors": (c_i, [C.POINTER(MaeCfg)]),
    "m3l_mae_step_ws_bytes": (c_sz, [C.POINTER(MaeCfg), c_i]),
    "m3l_mae_step_fwd": (c_i, [C.POINTER(MaeCfg), c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_mae_step_bwd": (c_i, [C.POINTER(MaeCfg), c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, C.POINTER(CommPlan), c_p]),
    "m3l_extractor_num_tensors": (c_i, [C.POINTER(MaeCfg), C.POINTER(TfCfg)]),
    "m3l_extractor_ws_bytes": (c_sz, [C.POINTER(MaeCfg), C.POINTER(TfCfg), c_i]),
    "m3l_extractor_fwd": (c_i, [C.POINTER(MaeCfg), C.POINTER(TfCfg), c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_extractor_bwd": (c_i, [C.POINTER(MaeCfg), C.POINTER(TfCfg), c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_earlycnn_ws_bytes": (c_sz, [C.POINTER(CnnCfg), c_i, c_i]),
    "m3l_earlycnn_fwd": (c_i, [C.POINTER(CnnCfg), c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "m3l_earlycnn_bwd": (c_i, [C.POINTER(CnnCfg), c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_tokens_assemble_ws_bytes": (c_sz, [C.POINTER(Geom), c_i]),
    "m3l_tokens_assemble_fwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "m3l_tokens_assemble_bwd": (c_i, [C.POINTER(Geom), c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_layernorm_ws_bytes": (c_sz, [c_i]),
    "m3l_layernorm_fwd": (c_i, [c_i, c_p, c_i, c_i, c_p, c_p, C.c_float, c_p, c_p, c_p]),
    "m3l_layernorm_bwd": (c_i, [c_i, c_p, c_p, c_i, c_i, c_p, C.c_float, c_p, c_p, c_p, c_p, c_p, c_p]),
    "m3l_gather_tokens": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "m3l_scatter_tokens": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "m3l_vt_load": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "m3l_vt_load2": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, C.c_double, C.c_double, c_p, c_p, c_i, c_i, c_i, c_i, c_i, C.c_double, C.c_double, c_p, c_p]),
    "m3l_adam_step_dev": (c_i, [c_p, c_p, c_p, c_p, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_p, c_p, c_p]),
    "m3l_adamw_step": (c_i, [c_p, c_p, c_p, c_p, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_i, C.c_float, C.c_float, c_p, c_i, c_p]),
    "m3l_adam_step": (c_i, [c_p, c_p, c_p, c_p, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_i, c_p]),
    "m3l_adam_step_scaled": (c_i, [c_p, c_p, c_p, c_p, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_i, C.c_float, c_p]),
    "m3l_prof_begin": (None, [C.c_char_p, c_i]),
    "m3l_prof_end": (None, []),
    "m3l_prof_event_overhead_us": (C.c_double, [c_p, c_i]),
    "m3l_prof_count": (c_i, []),
    "m3l_prof_get": (c_i, [c_i, C.c_char_p, c_sz, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double),
                           C.POINTER(C.c_double)]),
    "m3l_op_gemm_nt": (c_i, [c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "m3l_op_gemm_tn_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "m3l_op_gemm_tn": (c_i, [c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_sz, c_p, c_i, c_p]),
    "m3l_op_gemm_tn_grouped_ws_bytes": (c_sz, [c_i, c_i, c_i, c_p, c_p]),
    "m3l_op_gemm_tn_grouped": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "m3l_op_colsum_ws_bytes": (c_sz, [c_i]),
    "m3l_op_colsum": (c_i, [c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "m3l_op_prep_weight": (c_i, [c_i, c_p, c_i, c_i, c_p, c_p, c_p]),
    "m3l_op_mask_scale": (c_i, [c_i, c_p, c_p, C.c_float, C.c_long, c_p, c_p]),
    "m3l_op_concat2": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p, c_i, c_p]),
    "m3l_op_patch_cols": (c_i, [c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "m3l_op_vit_tokens": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "m3l_op_attn_fwd": (c_i, [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "m3l_op_attn_bwd": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
}

EXPORTS = tuple(_SIGS.keys())
_lib = None


class M3LError(RuntimeError):
    pass


def lib():
    """The loaded shared library (raises if it has not been built — there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M3LError(
                f"{LIB_PATH} not found: the HIP extension has not been built. Run "
                "`python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc, --offload-arch=gfx950). "
                "m3l_amd has no CPU / eager fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().m3l_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise M3LError(f"{what} failed (code {rc}): {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """void*[] of device pointers (host array handed to the C side)."""
    arr = (c_p * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr
