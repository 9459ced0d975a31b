"""Visual-visual alignment on the GPU.

Interface of /root/reference/mars/components/VisualVisualAlignmentModule.py:13-131:
``compute(support_imgs, support_masks, query_img) -> [G, G]`` with the side outputs
``similarity_matrix`` / ``cost_matrix`` (CPU tensors) and ``clear()``.

Differences in *how*, not *what*: one batched ViT pass over query + supports instead of three
passes (the attention maps are averaged inside it and never materialised), one similarity GEMM
instead of two, the normalisation fused into the patch extraction, and the
device-to-host copy of S is asynchronous so the exact-EMD stage can overlap the GPU.
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import ops
from .PriorInformationRefinementModule import PriorInformationRefinementModule, _pad64

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class SimilarityHandle:
    """Pinned host copy of S with the event that completes it (consumed by the EMD stage)."""

    def __init__(self, host: torch.Tensor, event: torch.cuda.Event, P: int):
        self.host, self.event, self.P = host, event, P

    def wait(self) -> torch.Tensor:
        self.event.synchronize()
        return self.host


class VisualVisualAlignmentModule:
    def __init__(self, model, model_transforms, model_patch_size: int, model_embedding_spatial_dimensions: int,
                 model_num_regs: int, vva_refinement_box_threshold: float, last_n_attention_maps_for_refinement: int, device):
        self.model = model
        self.model_transforms = model_transforms  # kept for API parity; the normalisation runs inside im2col
        self.model_patch_size = model_patch_size
        self.model_embedding_spatial_dimensions = int(model_embedding_spatial_dimensions)
        self.model_num_regs = model_num_regs
        self.device = torch.device(device)
        self.pir = PriorInformationRefinementModule(
            box_threshold=vva_refinement_box_threshold,
            last_n_attention_maps_for_refinement=last_n_attention_maps_for_refinement,
            device=device, num_regs=model_num_regs)
        self.last_n = last_n_attention_maps_for_refinement
        self._mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=self.device)
        self._std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=self.device)
        self._handle: Optional[SimilarityHandle] = None
        self._ws = {}
        self.vva_raw = None
        self.mean_attn = None
        self.support_bits = None

    # -- reference side outputs (lines 69-70) --
    @property
    def similarity_matrix(self):
        if self._handle is None:
            return None
        return self._handle.wait()[:, : self._handle.P].clone()

    @property
    def cost_matrix(self):
        s = self.similarity_matrix
        return None if s is None else (1 - s) / 2

    def similarity_handle(self) -> Optional[SimilarityHandle]:
        return self._handle

    def _buffers(self, ns: int, P: int, D: int):
        key = (ns, P, D)
        w = self._ws.get(key)
        if w is None:
            ld = _pad64(P)
            dev = self.device
            w = self._ws[key] = dict(
                ld=ld,
                F=torch.zeros((ns + 1) * P, D, dtype=torch.float32, device=dev),
                S=torch.zeros(ns * P, ld, dtype=torch.float32, device=dev),
                S_host=torch.zeros(ns * P, ld, dtype=torch.float32).pin_memory(),
                bits=torch.zeros(ns * P, dtype=torch.uint8, device=dev),
                ws=torch.zeros(4 * 32 * P, dtype=torch.float32, device=dev),
                prior=torch.zeros(P, dtype=torch.float32, device=dev),
                out=torch.zeros(P, dtype=torch.float32, device=dev),
            )
        return w

    def compute(self, support_imgs: torch.Tensor, support_masks: torch.Tensor, query_img: torch.Tensor) -> torch.Tensor:
        support_imgs = support_imgs.to(self.device)
        support_masks = support_masks.to(self.device)
        query_img = query_img.to(self.device)
        m = self.model
        G = self.model_embedding_spatial_dimensions
        P, R, D = G * G, self.model_num_regs, m.embed_dim
        ns = support_imgs.shape[1]
        w = self._buffers(ns, P, D)
        # lines 66-68: support features, query features and the query's attention maps in ONE batched
        # pass (query first, so that the attention-mean side output reads image 0)
        acc = self.pir.new_accumulator(P)
        imgs = torch.cat([query_img, support_imgs[0]], dim=0)
        X = m.forward_tokens(imgs, self._mean, self._std, mean_attn=(acc, self.last_n))
        ops.l2norm_rows(X, 1 + R, P, w["F"])
        w_fq, w_fs = w["F"][:P], w["F"][P:]
        # line 69: S = F_s F_q^T, then the asynchronous host copy the EMD stage waits on
        S = w["S"]
        ops.linear(w_fs, w_fq, S[:, :P])
        w["S_host"].copy_(S, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._handle = SimilarityHandle(w["S_host"], ev, P)
        # lines 72-102: pooled support mask, fg/bg column statistics, min-max
        ops.pool_masks(support_masks[0].contiguous().float(), G, w["bits"])
        ops.vva_prior(S, w["bits"], P, w["ws"], 32, w["prior"])
        self.support_bits = w["bits"]
        # lines 105-109: refinement and final min-max
        refined = self.pir.refine(w["prior"].view(G, G), acc)
        ops.minmax(refined.reshape(-1), w["out"])
        self.vva_raw, self.mean_attn = w["prior"].view(G, G), acc
        return w["out"].view(G, G)

    def _extract_patch_features(self, imgs):
        """Reference helper (lines 113-127): unit-length patch features [b*P, D]."""
        imgs = imgs.to(self.device)
        G, R = self.model_embedding_spatial_dimensions, self.model_num_regs
        X = self.model.forward_tokens(imgs, self._mean, self._std)
        out = torch.empty(X.shape[0] * G * G, self.model.embed_dim, dtype=torch.float32, device=self.device)
        ops.l2norm_rows(X, 1 + R, G * G, out)
        return out

    def clear(self):
        self._handle = None


def build_visual_visual_alignment_component(args, state_dict=None):
    """Builder with the reference's argument names (lines 133-169).  ``state_dict`` supplies the
    DINOv2 weights directly; otherwise they are read from ``args.models_path``."""
    from ...utils.backbone_loader import BackboneLoader
    dino_kwargs = dict(img_size=args.input_size, patch_size=14, init_values=1e-5, ffn_layer="mlp", block_chunks=0,
                       num_register_tokens=args.num_regs, qkv_bias=True, proj_bias=True, ffn_bias=True)
    model, transforms = BackboneLoader.load_backbone(
        backbone_name="dinov2", backbone_size=args.dino_backbone, device=args.device,
        backbone_weights_path=getattr(args, "dino_weights", None), encoder_kwargs=dict(dino_kwargs, state_dict=state_dict,
                                                                                      dtype=getattr(args, "dtype", torch.float32)))
    return VisualVisualAlignmentModule(
        model=model, model_transforms=transforms, model_patch_size=14,
        model_embedding_spatial_dimensions=args.input_size // 14, model_num_regs=args.num_regs,
        vva_refinement_box_threshold=args.vva_refinement_box_threshold,
        last_n_attention_maps_for_refinement=args.last_n_attn_for_vva_refinement, device=args.device)
