"""Device-side `vt_load` — reference: /root/reference/utils/pretrain_utils.py:7-57.

obs dict {'image': (B,H,W,3*fs), 'tactile': (B,3*S*fs,h,w)} (numpy or torch; float32 or uint8 go to the device as they are, other
dtypes are cast to float32 first) -> {'image': (B,3*fs,H,W), 'tactile1..S': (B,3*fs,h,w)} float32 CUDA tensors: NHWC->NCHW permute,
per-sensor channel pick `idx + 3*s` with idx = {f*3S + c}, `(x - lo) / (hi - lo)` with the reference's normalisation arguments
(defaults: image identity, tactile (x + 1) / 2), the 'tactile' key removed.  One HIP kernel each (m3l_vt_load2) instead of the
reference's host-side copies.

Deviation: the reference's `vt_load(path)` unpickles a dict from a .npy file (np.load(allow_pickle=True)); this drop-in refuses
to unpickle and accepts an .npz of arrays (np.load(path) -> keys 'image' / 'tactile') instead.
"""
import numpy as np
import torch

from . import _lib as L


def _as_cuda(a, device):
    """-> (contiguous CUDA tensor, is_uint8): float32 and uint8 observations travel as they are."""
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    if a.dtype == torch.uint8:
        return a.to(device=device).contiguous(), 1
    return a.to(device=device, dtype=torch.float32).contiguous(), 0


def vt_load(x, image_normalization=[0, 1], tactile_normalization=[-1, 1], squeeze=False, frame_stack=1, device="cuda"):
    if isinstance(x, str):
        if not x.endswith(".npz"):
            raise NotImplementedError("m3l_amd.vt_load(path): the reference loads a PICKLED dict from a .npy file "
                                      "(np.load(allow_pickle=True)); unpickling is refused here on purpose — save the observation "
                                      "arrays with np.savez(path, image=..., tactile=...) and pass the .npz, or pass the dict itself")
        with np.load(x, allow_pickle=False) as z:
            x = {k: z[k] for k in z.files}
    out = {k: v for k, v in x.items() if k not in ("image", "tactile")}
    img = tac = None
    img_u8 = tac_u8 = 0
    if "image" in x:
        img = x["image"][None] if len(x["image"].shape) == 3 else x["image"]
        assert img.shape[-1] == 3 * frame_stack
        img, img_u8 = _as_cuda(img, device)
    if "tactile" in x:
        tac = x["tactile"][None] if len(x["tactile"].shape) == 3 else x["tactile"]
        assert tac.shape[1] == 3 * frame_stack or tac.shape[1] == 6 * frame_stack or tac.shape[1] == 12 * frame_stack
        tac, tac_u8 = _as_cuda(tac, device)
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
    L.check(L.lib().m3l_vt_load2(L.ptr(img), img_u8, B, H, W, Cc, float(image_normalization[0]), float(image_normalization[1]), L.ptr(img_out),
                                 L.ptr(tac), tac_u8, th, tw, S, frame_stack, float(tactile_normalization[0]), float(tactile_normalization[1]),
                                 L.ptr_array(tac_outs), stream), "m3l_vt_load")
    if img_out is not None:
        out["image"] = img_out
    for s, t in enumerate(tac_outs):
        out["tactile" + str(s + 1)] = t
    if squeeze:
        for key in out:
            out[key] = out[key].squeeze()
    return out
