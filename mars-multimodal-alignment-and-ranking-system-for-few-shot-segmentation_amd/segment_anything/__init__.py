"""Segment-Anything proposal stage on the HIP kernels (reference: /root/reference/segment_anything)."""
from . import amg  # noqa: F401
from .automatic_mask_generator import ProposalGenerator, SamAutomaticMaskGenerator  # noqa: F401
from .build_sam import build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l, sam_model_registry  # noqa: F401
from .modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam  # noqa: F401
from .predictor import SamPredictor  # noqa: F401
