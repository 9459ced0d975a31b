"""MARS facade with the reference's surface (/root/reference/mars/MARS.py:14-116).

``predict(support_images, support_masks, query_image, mask_proposals=None) -> Tensor[H, W]``,
``clear()``, the three public timing attributes and ``build_MARS_fss(args)``.
"""
from __future__ import annotations

import inspect
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
        # start the proposal batch right behind the visual-visual tower (MARS_EARLY_BATCH=0: in the reference's order)
        self.early_batch = os.environ.get("MARS_EARLY_BATCH", "1") != "0"
        # the three stages side by side on streams of their own (see _predict)
        self.concurrent = os.environ.get("MARS_CONCURRENT", "1") != "0"
        self.vva_after_block = int(os.environ.get("MARS_VVA_AFTER_BLOCK", "1"))
        self._streams = None

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

        # lines 85-89.  What depends on what: the visual prior and the similarity matrix need the visual-visual
        # tower; the text prior needs the text-alignment tower and its prompt encoder; the proposal stage needs
        # the priors only for its last kernels - its pooling, its AlphaCLIP batch and its transport problems
        # depend on the query, the proposals and the similarity matrix alone.  The reference runs the three one
        # after the other.  Here (``concurrent``, the default) they are queued side by side:
        #   the visual-visual tower on a high-priority stream: the host solver waits for its similarity matrix;
        #   the proposal batch, 3/4 of the work, on the caller's stream; it starts at once;
        #   the text-alignment tower and the two prompt encoders on streams of their own: chains of several
        #   hundred small launches that fill what the batch leaves free (its rounds' tails, the gaps between
        #   its launches) instead of running alone on an idle chip behind it.
        # Same kernels on the same inputs, so results do not change.  Measured (one process, interleaved, a step
        # of the benchmark): everything on one stream 37.2-39.1 ms depending on the box, side by side 34.6-34.9;
        # all streams at one priority 36.0; the small chains at LOW priority 38.7 (they starve until the batch
        # ends); the batch at high priority and the visual-visual tower at normal 34.6-34.9 as well, but the
        # solver then starts late.  ``concurrent = False`` (MARS_CONCURRENT=0) queues everything on the
        # caller's stream, the batch right behind the visual-visual tower.
        text_alpha_clip = [f"a {name}."] if description == "" else [f"a {name}, {description}."]
        fmm = self.filtering_merging_component
        vva_mod = self.visual_visual_alignment_component
        vta_mod = self.visual_text_alignment_component
        G = vva_mod.model_embedding_spatial_dimensions
        dev = getattr(fmm, "device", None)
        staged = hasattr(fmm, "begin") and hasattr(vta_mod, "prepare_text") and hasattr(fmm, "prepare_text")
        if self.concurrent and staged and isinstance(dev, torch.device) and dev.type == "cuda":
            main = torch.cuda.current_stream(dev)
            st = self._get_streams(dev)
            for s_ in st.values():
                s_.wait_stream(main)
            # Order of issue: the host needs 2.4 ms to queue the batch and 3.1 ms to queue the visual-visual tower.
            # The batch goes first (the chip is full from the start instead of running the tower's mid-sized
            # launches alone for 3 ms), but only its first blocks: the tower is issued from inside the batch's
            # loop, behind block ``vva_after_block``, so that the solver does not get its matrix 2 ms later.
            queued = {}

            def queue_vva(i):
                if i == self.vva_after_block and "vva" not in queued:
                    with torch.cuda.stream(st["high"]):
                        queued["vva"] = vva_mod.compute(support_imgs=support_images, support_masks=support_masks, query_img=query_image)

            if "on_block" in inspect.signature(fmm.begin).parameters:
                fmm.begin(query_image, mask_proposals, support_masks, None, G, on_block=queue_vva)
            else:                               # a filtering module whose begin() takes no hook
                fmm.begin(query_image, mask_proposals, support_masks, None, G)
            if "vva" not in queued:             # fewer blocks than that, or no hook: behind the whole batch
                queue_vva(self.vva_after_block)
            vva = queued["vva"]
            handle = vva_mod.similarity_handle() if hasattr(vva_mod, "similarity_handle") else None
            if handle is not None:
                fmm.start_solver(handle)
            vta_mod.prepare_text(name, st["text0"])
            fmm.prepare_text(text_alpha_clip, st.get("text1", st["text0"]))
            with torch.cuda.stream(st["vta"]):
                vta = vta_mod.compute(query_image=query_image, fg_label=name, bg_labels=[])
                vta_g = self._text_prior_on_visual_grid(vta, vva.shape[0], dev)
            main.wait_stream(st["high"])
            main.wait_stream(st["vta"])
            # results of the side streams that the caller's stream reads from here on: tell the allocator, so that a
            # block freed meanwhile is not handed out again before this stream is done with it
            for t_ in (vva, vta, vta_g):
                if torch.is_tensor(t_) and t_.is_cuda:
                    t_.record_stream(main)
            if handle is None:
                fmm.start_solver(vva_mod.cost_matrix)
        else:
            vva = vva_mod.compute(support_imgs=support_images, support_masks=support_masks, query_img=query_image)
            handle = vva_mod.similarity_handle() if hasattr(vva_mod, "similarity_handle") else None
            if self.early_batch and hasattr(fmm, "begin"):
                fmm.begin(query_image, mask_proposals, support_masks, handle if handle is not None else vva_mod.cost_matrix, G)
            vta = vta_mod.compute(query_image=query_image, fg_label=name, bg_labels=[])
            vta_g = self._text_prior_on_visual_grid(vta, vva.shape[0], vva.device)
        self.vva, self.vta = vva.clone(), vta_g

        predicted_mask = self.filtering_merging_component.compute(
            query_img=query_image, mask_proposals=mask_proposals, support_mask=support_masks,
            cost_matrix=handle if handle is not None else vva_mod.cost_matrix,
            patch_features_spatial_dimension=G, vva=vva, vta=vta_g, text=text_alpha_clip)
        # the components write into buffers they reuse on the next call; the reference hands out fresh
        # tensors (FilteringMergingModule.py:219), so the caller gets a copy that the next predict() leaves alone
        predicted_mask = predicted_mask.clone()
        torch.cuda.synchronize(vva.device)  # kernels are asynchronous: the end stamp must include them
        self.time_end_ranking = time.time()
        return predicted_mask

    def _text_prior_on_visual_grid(self, vta, G, dev):
        """Lines 77-82: nearest resize of the text prior onto the visual grid, then min-max."""
        Gv = vta.shape[0]
        t = self._nearest_table(Gv, G, dev)
        tmp = torch.empty(Gv, G, dtype=torch.float32, device=dev)
        vta_g = torch.empty(G, G, dtype=torch.float32, device=dev)
        ops.resample_axis(vta.contiguous(), 1, Gv, Gv, 1, t, tmp)
        ops.resample_axis(tmp, 1, Gv, G, 0, t, vta_g)
        ops.minmax(vta_g.view(-1), vta_g.view(-1))
        return vta_g

    def _get_streams(self, dev):
        if self._streams is None or self._streams["device"] != dev:
            side = lambda: torch.cuda.Stream(device=dev)      # noqa: E731
            # A process has four hardware queues on this platform: the caller's stream, the tower's, the text-alignment
            # tower's and ONE for both prompt encoders keep every stream on a queue of its own (a fifth would share a
            # queue with one of the others, which one being the runtime's choice).  MARS_TEXT_STREAMS=2: one each.
            self._streams = dict(device=dev, high=torch.cuda.Stream(device=dev, priority=-1), vta=side(), text0=side())
            if os.environ.get("MARS_TEXT_STREAMS", "1") == "2":
                self._streams["text1"] = side()
        return {k: v for k, v in self._streams.items() if k != "device"}

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
