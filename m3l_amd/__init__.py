"""m3l_amd — MI355X-native (gfx950, hand-written HIP) masked multimodal auto-encoder training step: a drop-in for the
VTT / VTMAE representation path of Leonhard111/M3L (models/pretrain_models.py, models/VTT.py) and nothing else."""
from ._lib import LIB_PATH, M3LError  # noqa: F401
from .pretrain_models import VTMAE, VTT, Transformer  # noqa: F401
from .dino_vtt import VTT as DinoVTT  # noqa: F401  (reference: models/VTT.py — a second class that is also called VTT)
from .pretrain_utils import vt_load  # noqa: F401
from .dinov2 import DinoV2Frozen  # noqa: F401  (reference: torch.hub dinov2_vits14_reg, train_dino_cat_mae.py:29)
from .fusion import DinoCatMAEExtractor, MAEExtractor  # noqa: F401  (reference: MAEExtractor, models/pretrain_models.py:788-841 and models/pretrain_models_dino_cat_mae.py:793-904)

__all__ = ["VTT", "VTMAE", "Transformer", "DinoVTT", "DinoV2Frozen", "DinoCatMAEExtractor", "MAEExtractor", "vt_load", "M3LError", "LIB_PATH"]
