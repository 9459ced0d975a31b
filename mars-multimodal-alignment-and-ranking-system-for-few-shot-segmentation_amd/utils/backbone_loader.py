"""Backbone factory with the reference's seam (/root/reference/utils/backbone_loader.py:14-52):
``BackboneLoader.load_backbone(backbone_name, backbone_size, backbone_weights_path, device,
encoder_kwargs) -> (encoder, transforms)``.  The encoders are the HIP-backed classes of
``models``; the returned transforms are descriptors only, because resizing and normalisation
are fused into the kernels that consume the image.

Weights come, in this order, from ``encoder_kwargs['state_dict']``, from a checkpoint file read
with ``torch.load(..., weights_only=True)`` (same key layout as the reference checkpoints), or --
when neither is given -- from the seeded synthetic generator (``encoder_kwargs['seed']``).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .. import models, synth


def _load_file(path: Optional[str]):
    if path and os.path.isfile(path):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "teacher" in sd:   # dinov2/utils/utils.py:25-27 checkpoint_key
            sd = {k.replace("module.", "").replace("backbone.", ""): v for k, v in sd["teacher"].items()}
        return sd
    return None


class BackboneLoader:
    AVAILABLE_BACKBONES = {"dinov2", "alphaclip", "clip"}

    @classmethod
    def load_backbone(cls, backbone_name: str, backbone_size: str = None, backbone_weights_path: str = None,
                      device: str = "cuda", encoder_kwargs: dict = None):
        assert backbone_name in cls.AVAILABLE_BACKBONES, \
            f"Backbone {backbone_name} not available. Choose from {cls.AVAILABLE_BACKBONES}"
        kw = dict(encoder_kwargs or {})
        sd = kw.get("state_dict") or _load_file(backbone_weights_path)
        dtype = kw.get("dtype", torch.float32)
        seed = kw.get("seed", 0)
        if backbone_name == "dinov2":
            cfg = synth.dino_cfg(backbone_size, img_size=kw.get("img_size", 518), patch_size=kw.get("patch_size", 14),
                                 num_regs=kw.get("num_register_tokens", 4))
            if sd is None:
                sd = synth.dinov2_state_dict(cfg, seed)
            enc = models.DinoVisionTransformer(cfg, sd, device=device, dtype=dtype)
            return enc, {"normalize": ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))}
        cfg = synth.CLIP_ARCHS[backbone_size]
        if backbone_name == "clip":
            if sd is None:
                sd = synth.clip_state_dict(cfg, seed, alpha=False)
            enc = models.CLIP(cfg, sd, device=kw.get("device", device), dtype=dtype)
            return enc, {"normalize": ((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))}
        if sd is None:
            sd = synth.clip_state_dict(cfg, seed, alpha=True)
        enc = models.AlphaCLIP(cfg, sd, device=kw.get("device", device), dtype=dtype)
        img_t = {"resize": cfg.image_resolution, "interpolation": "bicubic(PIL, 8-bit)",
                 "normalize": ((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))}
        mask_t = {"resize": (cfg.image_resolution, cfg.image_resolution), "interpolation": "bilinear(antialias)",
                  "normalize": (0.5, 0.26)}
        return enc, (img_t, mask_t)
