"""AlphaCLIP prompt scoring of generated masks.

Interface of /root/reference/mars/ClipPipeline.py:22-238: ``set_reference``, ``set_target``,
``set_class_name``, ``set_candidate_mask``, ``process_masks(text) -> (masks, {idx: [1, P] array})``,
``predict``, ``reset_clip_pipeline``.

The reference runs one AlphaCLIP pass per mask (lines 156-169); here the alpha planes of all masks
are resampled by one kernel and go through the tower as one batch, and the N x P table of
100 * cosine comes from one launch.  The mask generator, prompt generator and mask merger are
objects of the caller (``generate(image_np)``, ``prompt_generation(name)``, ``merge_masks(...)``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .components.FilteringMergingModule import FilteringMergingModule


class CLIPPipeline:
    def __init__(self, alphaclip_model, alphaclip_transforms=(None, None), prompt_generator=None, mask_merger=None, generator=None,
                 vlm=None, input_size: int = 518, min_mask_region_area: int = 0, containment_threshold: float = 0.0,
                 visualize: bool = False, device="cuda:0", tokenizer=None):
        self.alphaclip_model = alphaclip_model
        self.alphaclip_img_transforms, self.alphaclip_mask_transforms = alphaclip_transforms[0], alphaclip_transforms[1]
        self.prompt_generator, self.mask_merger, self.generator, self.vlm = prompt_generator, mask_merger, generator, vlm
        self.input_size = input_size
        self.min_mask_region_area, self.containment_threshold = min_mask_region_area, containment_threshold
        self.visualize = visualize
        self.device = torch.device(device)
        # the AlphaCLIP transforms and the batched tower pass are the ones of the filtering module
        self._fmm = FilteringMergingModule(alphaclip_model, alphaclip_transforms[0], alphaclip_transforms[1], alpha=0.0,
                                           static_threshold=0.0, dynamic_threshold=0.0, device=device, tokenizer=tokenizer)
        self.reset_clip_pipeline()

    # ---------------------------------------------------------------- lines 70-122
    def set_reference(self, reference_image: torch.Tensor, reference_mask: torch.Tensor):
        ref = reference_image.flatten(0, 2).permute(1, 2, 0).cpu().numpy()
        assert ref.shape[0] == self.input_size
        mask = reference_mask.clone()
        if mask.sum() == 0:
            sh, sw = mask.shape[-2:]
            mask[..., (sh // 2 - 7):(sh // 2 + 7), (sw // 2 - 7):(sw // 2 + 7)] = 1
        self.reference_img, self.reference_mask = ref, mask.permute(1, 0, 2, 3)

    def set_target(self, target_image: torch.Tensor):
        t = target_image.flatten(0, 1)
        assert t.shape[-1] == self.input_size
        self._target_chw = t.to(self.device).float().contiguous()
        self.target_img = t.permute(1, 2, 0).cpu().numpy()

    def set_class_name(self, class_name: str):
        self.class_name = class_name

    def set_candidate_mask(self, mask):
        self.candidate_mask = mask

    # ---------------------------------------------------------------- lines 128-171
    def process_masks(self, text: List[str]) -> Tuple[List[dict], Dict[int, np.ndarray]]:
        generated_masks = list(self.generator.generate(self.target_img)) if self.generator is not None else []
        if self.candidate_mask is not None and len(self.candidate_mask) > 0:
            for c_mask in self.candidate_mask:
                generated_masks.append({"segmentation": c_mask, "type": "matcher"})
        if not generated_masks:
            return generated_masks, {}
        segs = [torch.as_tensor(np.asarray(g["segmentation"])) for g in generated_masks]
        masks = torch.stack(segs).to(self.device).float().contiguous()
        tfeat = self._fmm._compute_alphaclip_text_feats(text).float().contiguous()
        feats = self._fmm._compute_alphaclip_vis_feats(self._target_chw, masks)
        table = torch.empty(masks.shape[0], tfeat.shape[0], dtype=torch.float32, device=self.device)
        ops.cosine_table(feats[: masks.shape[0]], tfeat, table, scale=100.0)
        host = table.cpu().numpy()
        return generated_masks, {i: host[i:i + 1].copy() for i in range(host.shape[0])}

    # ---------------------------------------------------------------- lines 176-206
    def predict(self):
        if self.class_name is None:
            if self.vlm is None:
                raise ValueError("If the class name is not set, the VLM object must be set.")
            self.class_name = self.vlm.fetch_class_name(self.reference_img,
                                                        self.reference_mask.squeeze(0).squeeze(0).cpu().numpy()).lower()
        if self.prompt_generator is None or self.mask_merger is None:
            raise ValueError("predict needs a prompt generator and a mask merger; process_masks works without them")
        self.prompts = self.prompt_generator.prompt_generation(self.class_name)
        generated_masks, mask_sim_pairs = self.process_masks(self.prompts)
        self.predicted_mask = self.mask_merger.merge_masks(masks=generated_masks, mask_sim_pairs=mask_sim_pairs,
                                                           prompts=self.prompts, show=self.visualize)
        return self.predicted_mask["merged_mask"]

    def reset_clip_pipeline(self):
        self.reference_img = self.reference_mask = self.target_img = None
        self._target_chw = None
        self.class_name = self.description = None
        self.prompts: List[str] = []
        self.predicted_mask = None
        self.candidate_mask = None
        if self.generator is not None and hasattr(self.generator, "reset_stored_features"):
            self.generator.reset_stored_features()
