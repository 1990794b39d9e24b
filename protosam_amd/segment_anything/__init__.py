"""HIP-backed counterpart of the reference's `segment_anything` package surface (models/segment_anything/__init__.py)."""
from .build_sam import build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l, sam_model_registry  # noqa: F401
from .predictor import SamPredictor  # noqa: F401
from .automatic_mask_generator import SamAutomaticMaskGenerator  # noqa: F401
