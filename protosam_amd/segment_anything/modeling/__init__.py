from .image_encoder import ImageEncoderViT
from .mask_decoder import MaskDecoder
from .prompt_encoder import PromptEncoder
from .sam import Sam
from .transformer import TwoWayTransformer

SamBatched = Sam  # the vendored registry's class name (build_sam.py:66); here a variant flag of `Sam`
