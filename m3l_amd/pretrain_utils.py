"""Device-side `vt_load` — reference: /root/reference/utils/pretrain_utils.py:7-57.

obs dict {'image': (B,H,W,3*fs), 'tactile': (B,3*S*fs,h,w)} (numpy or torch, any float/uint dtype) ->
{'image': (B,3*fs,H,W), 'tactile1..S': (B,3*fs,h,w)} float32 CUDA tensors: NHWC->NCHW permute, per-sensor channel pick
`idx + 3*s` with idx = {f*3S + c}, tactile rescale (x + 1) / 2, the 'tactile' key removed.  The permute / channel pick
run in one HIP kernel each (m3l_vt_load) instead of the reference's host-side copies.
"""
import numpy as np
import torch

from . import _lib as L


def _as_cuda_f32(a, device):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(device=device, dtype=torch.float32).contiguous()


def vt_load(x, image_normalization=[0, 1], tactile_normalization=[-1, 1], squeeze=False, frame_stack=1, device="cuda"):
    if isinstance(x, str):
        x = np.load(x, allow_pickle=False).item()
    if list(image_normalization) != [0, 1] or list(tactile_normalization) != [-1, 1]:
        raise NotImplementedError("m3l_amd.vt_load implements the reference defaults image [0,1] / tactile [-1,1]")
    out = {k: v for k, v in x.items() if k not in ("image", "tactile")}
    img = tac = None
    if "image" in x:
        img = x["image"][None] if len(x["image"].shape) == 3 else x["image"]
        assert img.shape[-1] == 3 * frame_stack
        img = _as_cuda_f32(img, device)
    if "tactile" in x:
        tac = x["tactile"][None] if len(x["tactile"].shape) == 3 else x["tactile"]
        assert tac.shape[1] == 3 * frame_stack or tac.shape[1] == 6 * frame_stack or tac.shape[1] == 12 * frame_stack
        tac = _as_cuda_f32(tac, device)
    stream = torch.cuda.current_stream().cuda_stream
    img_out, tac_outs, S = None, [], 0
    B = H = W = Cc = th = tw = 0
    if img is not None:
        B, H, W, Cc = img.shape
        img_out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=img.device)
    if tac is not None:
        Bt, CH, th, tw = tac.shape
        S = (CH // frame_stack) // 3
        tac_outs = [torch.empty(Bt, 3 * frame_stack, th, tw, dtype=torch.float32, device=tac.device) for _ in range(S)]
        B = B or Bt
    L.check(L.lib().m3l_vt_load(L.ptr(img), B, H, W, Cc, L.ptr(img_out), L.ptr(tac), th, tw, S, frame_stack,
                                L.ptr_array(tac_outs), stream), "m3l_vt_load")
    if img_out is not None:
        out["image"] = img_out
    for s, t in enumerate(tac_outs):
        out["tactile" + str(s + 1)] = t
    if squeeze:
        for key in out:
            out[key] = out[key].squeeze()
    return out
