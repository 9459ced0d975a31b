"""SamPredictor on the HIP kernels (interface of /root/reference/segment_anything/predictor.py:17-251
and utils/transforms.py:16-102)."""
from __future__ import annotations

from copy import deepcopy
from typing import Optional, Tuple

import numpy as np
import torch

from .. import ops
from .modeling import Sam


class ResizeLongestSide:
    def __init__(self, target_length: int):
        self.target_length = target_length

    @staticmethod
    def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
        scale = long_side_length * 1.0 / max(oldh, oldw)
        return int(oldh * scale + 0.5), int(oldw * scale + 0.5)

    def apply_coords(self, coords: np.ndarray, original_size) -> np.ndarray:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = deepcopy(coords).astype(float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes(self, boxes: np.ndarray, original_size) -> np.ndarray:
        """transforms.py:59-65: [B, 4] boxes as two corner points each."""
        return self.apply_coords(np.asarray(boxes).reshape(-1, 2, 2), original_size).reshape(-1, 4)


class SamPredictor:
    def __init__(self, sam_model: Sam):
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.cfg.img_size)
        self.reset_image()

    @property
    def device(self):
        return self.model.device

    def reset_image(self):
        self.is_image_set = False
        self.features_tokens = None
        self.original_size = self.input_size = None

    @property
    def features(self) -> torch.Tensor:
        """[1, C, G, G] view of the stored embedding (reference attribute)."""
        G = self.model.image_encoder.G
        return self.features_tokens.view(1, G, G, -1).permute(0, 3, 1, 2)

    def set_image(self, image: np.ndarray, image_format: str = "RGB", stored_features: Optional[torch.Tensor] = None):
        """uint8 HWC image -> embedding (predictor.py:36-100).  The Pillow bilinear resize of
        ResizeLongestSide.apply_image runs as two 8-bit resample passes on the GPU."""
        assert image_format in ("RGB", "BGR")
        if image_format != self.model.image_format:
            image = image[..., ::-1]
        self.reset_image()
        dev, m = self.device, self.model
        h, w = image.shape[:2]
        nh, nw = self.transform.get_preprocess_shape(h, w, m.cfg.img_size)
        src = torch.as_tensor(np.ascontiguousarray(image)).to(dev).permute(2, 0, 1).contiguous()
        tmp = torch.empty(3, h, nw, dtype=torch.uint8, device=dev)
        dst = torch.empty(3, nh, nw, dtype=torch.uint8, device=dev)
        ops.resample_axis_u8(src, 3, h, w, 1, m._table("pil", w, nw), tmp)
        ops.resample_axis_u8(tmp, 3, h, nw, 0, m._table("pil", h, nh), dst)
        self.original_size, self.input_size = (h, w), (nh, nw)
        if stored_features is not None:
            f = torch.as_tensor(stored_features, device=dev).float()
            self.features_tokens = f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).contiguous()
        else:
            self.features_tokens = m.image_encoder.forward_tokens(m.preprocess_u8(dst)).clone()
        self.is_image_set = True

    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output: bool = True,
                      return_logits: bool = False, upscale: bool = True):
        """predictor.py:177-251: coords [B, N, 2] and boxes [B, 4] in the model's input frame, labels [B, N],
        mask_input [B or 1, 1, 4G, 4G] low-resolution logits of an earlier prediction.  ``upscale=False`` (not in the reference)
        skips postprocess_masks and returns (None, iou, low)."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        m = self.model
        sparse = m.prompt_encoder.embed_prompts((point_coords, point_labels) if point_coords is not None else None, boxes)
        dense_tokens = None
        if mask_input is not None:                                # low-resolution logits [B or 1, 1, 4G, 4G]
            mi = torch.as_tensor(mask_input).to(self.device).float()
            dense_tokens = m.prompt_encoder.embed_masks_tokens(mi.expand(sparse.shape[0], -1, -1, -1))
        masks4, iou4 = m.mask_decoder.predict_masks(self.features_tokens, m.prompt_encoder.dense_pe_tokens, sparse,
                                                    m.prompt_encoder.no_mask, dense_tokens=dense_tokens)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        if not upscale:             # (for callers that evaluate the image-size logits on the fly: a view of the decoder's buffer,
            return None, iou4[:, sl].contiguous(), masks4[:, sl]        #  valid until the next prediction)
        low, iou = masks4[:, sl].contiguous(), iou4[:, sl].contiguous()
        masks = m.postprocess_masks(low, self.input_size, self.original_size)
        if not return_logits:
            masks = masks > m.mask_threshold
        return masks, iou, low
