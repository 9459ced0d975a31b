"""MARS facade with the reference's surface (/root/reference/mars/MARS.py:14-116).

``predict(support_images, support_masks, query_image, mask_proposals=None) -> Tensor[H, W]``,
``clear()``, the three public timing attributes and ``build_MARS_fss(args)``.
"""
from __future__ import annotations

import os
import time
from typing import Dict, Optional

import torch

from .. import ops, resample
from .components.FilteringMergingModule import FilteringMergingModule, build_filtering_and_merging_module
from .components.TextRetrieverModule import TextRetrieverModule, build_text_retriever_component
from .components.VisualTextAlignmentModule import VisualTextAlignmentModule, build_visual_text_alignment_component
from .components.VisualVisualAlignmentModule import VisualVisualAlignmentModule, build_visual_visual_alignment_component


class MARS:
    def __init__(self, text_retriever_component: TextRetrieverModule,
                 visual_text_alignment_component: VisualTextAlignmentModule,
                 visual_visual_alignment_component: VisualVisualAlignmentModule,
                 filtering_merging_component: FilteringMergingModule, mask_generator=None):
        self.text_retriever_component = text_retriever_component
        self.visual_text_alignment_component = visual_text_alignment_component
        self.visual_visual_alignment_component = visual_visual_alignment_component
        self.filtering_merging_component = filtering_merging_component
        self.mask_generator = mask_generator
        self.time_start_ranking = None
        self.time_start_ranking_after_text_extraction = None
        self.time_end_ranking = None
        self._tables: Dict = {}
        self.vva = None
        self.vta = None
        self.overlap = os.environ.get("MARS_OVERLAP", "0") == "1"

    def _nearest_table(self, n_in: int, n_out: int, device):
        """F.interpolate(mode='nearest') source indices: floor(dst * in/out) in float32 (line 77-81)."""
        key = (n_in, n_out)
        if key not in self._tables:
            import numpy as np
            scale = np.float32(n_in) / np.float32(n_out)
            idx = np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int32), n_in - 1)
            self._tables[key] = (torch.from_numpy(idx.astype(np.int32)).to(device),
                                 torch.ones(n_out, dtype=torch.int32, device=device),
                                 torch.ones(n_out, 1, dtype=torch.float32, device=device))
        return self._tables[key]

    def predict(self, support_images: torch.Tensor, support_masks: torch.Tensor, query_image: torch.Tensor,
                mask_proposals: Optional[torch.Tensor] = None):
        # the kernels run on the current device's current stream: make the components' GPU current for the call
        device = getattr(self.filtering_merging_component, "device", None)
        if isinstance(device, torch.device) and device.type == "cuda":
            with torch.cuda.device(device):
                return self._predict(support_images, support_masks, query_image, mask_proposals)
        return self._predict(support_images, support_masks, query_image, mask_proposals)

    def _predict(self, support_images, support_masks, query_image, mask_proposals):
        self.time_start_ranking = time.time()
        assert (mask_proposals is not None or self.mask_generator is not None)
        if self.mask_generator is not None:
            mask_proposals = self.mask_generator.generate(support_images, support_masks, query_image)

        name, description = self.text_retriever_component.get_conceptual_information(
            support_images=support_images, support_masks=support_masks)
        self.time_start_ranking_after_text_extraction = time.time()

        # lines 85-89; the AlphaCLIP batch needs nothing from the alignment towers, so with MARS_OVERLAP=1 it
        # is queued first, on a stream of its own, and the towers run beside it.  Off by default: at 64
        # proposals the batch already fills the chip and sharing it cost 6 % (52.9 against 49.9 ms a query).
        # Giving the towers a CU range of their own (a stream with a CU mask, 24-64 of the 256 CUs) was worse
        # still, 90-109 ms: their products fill most of the chip one tile per CU, so on a fifth of the CUs
        # they take five times as long and the batch waits for them.
        text_alpha_clip = [f"a {name}."] if description == "" else [f"a {name}, {description}."]
        fmm = self.filtering_merging_component
        if self.overlap and hasattr(fmm, "prefetch"):
            fmm.prefetch(query_image, mask_proposals, text_alpha_clip)

        vva_mod = self.visual_visual_alignment_component
        vva = vva_mod.compute(support_imgs=support_images, support_masks=support_masks, query_img=query_image)
        vta = self.visual_text_alignment_component.compute(query_image=query_image, fg_label=name, bg_labels=[])

        # lines 77-82: nearest resize of the text prior onto the visual grid, then min-max
        G, Gv = vva.shape[0], vta.shape[0]
        dev = vva.device
        t = self._nearest_table(Gv, G, dev)
        tmp = torch.empty(Gv, G, dtype=torch.float32, device=dev)
        vta_g = torch.empty(G, G, dtype=torch.float32, device=dev)
        ops.resample_axis(vta.contiguous(), 1, Gv, Gv, 1, t, tmp)
        ops.resample_axis(tmp, 1, Gv, G, 0, t, vta_g)
        ops.minmax(vta_g.view(-1), vta_g.view(-1))
        self.vva, self.vta = vva.clone(), vta_g

        handle = vva_mod.similarity_handle() if hasattr(vva_mod, "similarity_handle") else None
        predicted_mask = self.filtering_merging_component.compute(
            query_img=query_image, mask_proposals=mask_proposals, support_mask=support_masks,
            cost_matrix=handle if handle is not None else vva_mod.cost_matrix,
            patch_features_spatial_dimension=vva_mod.model_embedding_spatial_dimensions,
            vva=vva, vta=vta_g, text=text_alpha_clip)
        # the components write into buffers they reuse on the next call; the reference hands out fresh
        # tensors (FilteringMergingModule.py:219), so the caller gets a copy that the next predict() leaves alone
        predicted_mask = predicted_mask.clone()
        torch.cuda.synchronize(dev)  # kernels are asynchronous: the end stamp must include them
        self.time_end_ranking = time.time()
        return predicted_mask

    def clear(self):
        self.visual_visual_alignment_component.clear()


def build_MARS_fss(args, state_dicts: Optional[dict] = None):
    """Reference builder (lines 110-116).  ``state_dicts`` = dict(dino=..., clip=..., alphaclip=...)
    injects weights directly (synthetic runs); otherwise checkpoints are read from ``args.models_path``."""
    sd = state_dicts or {}
    return MARS(
        text_retriever_component=build_text_retriever_component(args=args),
        visual_text_alignment_component=build_visual_text_alignment_component(args=args, state_dict=sd.get("clip")),
        visual_visual_alignment_component=build_visual_visual_alignment_component(args=args, state_dict=sd.get("dino")),
        filtering_merging_component=build_filtering_and_merging_module(args=args, state_dict=sd.get("alphaclip")),
    )
