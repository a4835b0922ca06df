"""cfg-5 fusion head: MAE token embeddings + a frozen DINOv2 image feature -> policy features
(reference: `MAEExtractor` in models/pretrain_models_dino_cat_mae.py:793-904, built by train_dino_cat_mae.py:140-197).

    obs --vt_load--> x --mae.get_embeddings--> (B, N, D) --1-layer Transformer--> mean over tokens --+
    x['image'][:, middle RGB frame] --frozen DINOv2-S/14-reg--> (B, D) ------------------------------cat--> MLP (2D -> 2D -> 2D -> D)

Everything runs on this package's HIP kernels: VTMAE.get_embeddings, Transformer, DinoV2Frozen, the concat and the three-Linear fusion MLP
(`self.mlp` keeps the reference's nn.Sequential layout, so state dicts match; its arithmetic goes through the f32 NT / TN GEMMs of the C
ABI with the bias + ReLU epilogue — functional.LinearActFn — and Dropout(0.1) is a keep-mask drawn with torch's generator, as the mask
noise of the MAE is, applied by a HIP kernel).

The reference class derives from stable-baselines3's `BaseFeaturesExtractor`; SB3 is not part of this package, so this is a plain
`nn.Module` with the same constructor arguments after `observation_space`, the same sub-module / parameter names (`vit_layer`,
`mlp`, `query`, `query_projection`, `key_projection` — the last three are created but unused by the reference's forward too) and
`features_dim`.  INTEGRATION.md shows the two-line SB3 wrapper.

Reference defect NOT reproduced: the middle-frame slice `image[:, 3*mid-3 : 3*mid]` with `mid = frame_stack // 2`
(:883-885) is EMPTY for frame_stack == 1; here frame_stack == 1 uses the only frame (channels 0..2).
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from .pretrain_models import VTT
from .pretrain_utils import vt_load


def pooled_embeddings(mae, head, vt, use_tactile=True):
    """torch.mean(head(mae.get_embeddings(vt, eval=False, use_tactile=...)), dim=1) — the chain both reference extractors run
    (models/pretrain_models.py:834-838, models/pretrain_models_dino_cat_mae.py:887-896) — as ONE autograd node over m3l_extractor_fwd /
    m3l_extractor_bwd (csrc/mae_step.hip) when the library chain applies, else through the per-module Functions.  head: a Transformer."""
    mae.train()                                               # get_embeddings(eval=False) leaves the MAE in train mode (:590-593)
    image, tactiles, geom, ref = mae._inputs(vt, True, use_tactile)
    enc_tf = mae.encoder.transformer
    fused = (Fn.FUSED_EXTRACTOR and head.dim == enc_tf.dim and head.compute_dtype == enc_tf.compute_dtype
             and (not mae.early_conv_masking or (mae.encoder.image_patch_height == 8 and mae.encoder.tactile_patch_height == 4)))
    if not fused:
        tokens = mae.get_embeddings(vt, eval=False, use_tactile=use_tactile)
        return torch.mean(head(tokens), dim=1)
    Fn._require_cuda(ref, "MAE input")
    plan = Fn.StepPlan()
    learned = not mae.use_sincosmod_encodings
    plan.cfg = L.MaeCfg(geom, enc_tf._cfg(), enc_tf._cfg(), 0.5, int(mae.early_conv_masking), int(learned))
    plan.head_cfg = head._cfg()
    has_img, has_tac = image is not None, len(tactiles) > 0
    front, used = mae._front_tensors(geom, has_img, has_tac)
    plan.tensors = front + enc_tf._tensors() + head._tensors()
    plan.used = used + [True] * (11 * enc_tf.depth + 2) + [True] * (11 * head.depth + 2)
    for i, t in enumerate(plan.tensors):
        if t is None:
            plan.used[i] = False
    plan.image = Fn._f32c(image)
    plan.tactiles = [Fn._f32c(t) for t in tactiles]
    plan.B = ref.shape[0]
    return Fn.ExtractorFn.apply(plan, *[t for t in plan.tensors if t is not None])


def _flatten_frame_stack(observations):
    """(B, fs, H, W, 3) -> (B, H, W, 3*fs) and (B, fs, 6, h, w) -> (B, 6*fs, h, w), as both reference extractors do before vt_load."""
    obs = dict(observations)
    if "image" in obs and len(obs["image"].shape) == 5:
        im = obs["image"].permute(0, 2, 3, 1, 4)
        obs["image"] = im.reshape(im.shape[0], im.shape[1], im.shape[2], -1)
    if "tactile" in obs and len(obs["tactile"].shape) == 5:
        t = obs["tactile"]
        obs["tactile"] = t.reshape(t.shape[0], -1, t.shape[3], t.shape[4])
    return obs


class MAEExtractor(nn.Module):
    """Policy-side consumer of the MAE (reference `MAEExtractor`, models/pretrain_models.py:788-841): observations -> vt_load ->
    `mae_model.get_embeddings(eval=False)` -> 1-layer Transformer (`vit_layer.transformer`) -> mean over tokens -> (B, dim_embeddings).
    Same constructor arguments after `observation_space`, same sub-module names; a plain nn.Module (SB3 is not part of this package)."""

    def __init__(self, mae_model, dim_embeddings, vision_only_control, frame_stack, observation_space=None):
        super().__init__()
        self.features_dim = dim_embeddings
        self.flatten = nn.Flatten()
        self.mae_model = mae_model
        self.running_buffer = {}
        self.vision_only_control = vision_only_control
        self.frame_stack = frame_stack
        self.vit_layer = VTT(image_size=(64, 64), tactile_size=(32, 32), image_patch_size=8, tactile_patch_size=4,    # sizes unused:
                             dim=dim_embeddings, depth=1, heads=4, mlp_dim=dim_embeddings * 2, num_tactiles=2)       # only .transformer runs
        self.vit_layer.transformer.compute_dtype = getattr(mae_model, "compute_dtype", "fp32")     # one compute type for the whole chain

    def forward(self, observations):
        dev = self.vit_layer.pos_embedding.device
        vt = vt_load(_flatten_frame_stack(observations), frame_stack=self.frame_stack, device=dev)
        return self.flatten(pooled_embeddings(self.mae_model, self.vit_layer.transformer, vt, use_tactile=not self.vision_only_control))


class DinoCatMAEExtractor(nn.Module):
    def __init__(self, dino_model, mae_model, dim_embeddings, vision_only_control, frame_stack, observation_space=None):
        super().__init__()
        self.features_dim = dim_embeddings
        self.dim_embeddings = dim_embeddings
        self.flatten = nn.Flatten()
        self.mae_model = mae_model
        self.dino_model = dino_model
        self.running_buffer = {}
        self.vision_only_control = vision_only_control
        self.frame_stack = frame_stack
        self.vit_layer = VTT(image_size=(70, 70), tactile_size=(70, 70), image_patch_size=14, tactile_patch_size=14,   # sizes unused:
                             dim=dim_embeddings, depth=1, heads=4, mlp_dim=dim_embeddings * 2, num_tactiles=2)        # only .transformer runs
        self.vit_layer.transformer.compute_dtype = getattr(mae_model, "compute_dtype", "fp32")      # one compute type for the whole chain
        self.mlp = nn.Sequential(nn.Linear(dim_embeddings * 2, dim_embeddings * 2), nn.ReLU(), nn.Dropout(0.1),
                                 nn.Linear(dim_embeddings * 2, dim_embeddings * 2), nn.ReLU(), nn.Dropout(0.1),
                                 nn.Linear(dim_embeddings * 2, dim_embeddings))
        self.query = nn.Parameter(torch.randn(1, 1, dim_embeddings))
        self.query_projection = nn.Linear(dim_embeddings, dim_embeddings)
        self.key_projection = nn.Linear(dim_embeddings, dim_embeddings)

    def middle_frame(self, image):
        """(B, 3*fs, H, W) -> (B, 3, H, W): frame fs // 2 - 1 as the reference slices it (:883-885); the only frame when fs == 1."""
        mid = self.frame_stack // 2
        lo = 3 * mid - 3 if mid >= 1 else 0
        return image[:, lo:lo + 3]

    def forward(self, observations):
        dev = self.query.device
        vt = vt_load(_flatten_frame_stack(observations), frame_stack=self.frame_stack, device=dev)
        pooled = pooled_embeddings(self.mae_model, self.vit_layer.transformer, vt, use_tactile=not self.vision_only_control)
        dino = self.dino_model(self.middle_frame(vt["image"]))
        return self.run_mlp(Fn.Concat2Fn.apply(self.flatten(pooled), dino))

    def run_mlp(self, x):
        """self.mlp = Linear, ReLU, Dropout(0.1), Linear, ReLU, Dropout(0.1), Linear (:828-836) through the HIP GEMMs."""
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        drops = [m for m in self.mlp if isinstance(m, nn.Dropout)]
        for i, l in enumerate(lin):
            last = i + 1 == len(lin)
            mask, scale = None, 1.0
            if not last and self.training and drops[i].p > 0:
                mask = (torch.rand(x.shape[0], l.out_features, device=x.device) >= drops[i].p).to(torch.uint8)
                scale = 1.0 / (1.0 - drops[i].p)
            x = Fn.LinearActFn.apply(x, l.weight, l.bias, not last, mask, scale)
        return x
