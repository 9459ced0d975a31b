"""Model registry (interface of /root/reference/segment_anything/build_sam.py:14-106)."""
from __future__ import annotations

from typing import Optional

import torch

from ..synth import SAM_ARCHS, sam_state_dict
from .modeling import Sam


def _build(arch: str, checkpoint: Optional[str] = None, state_dict=None, device="cuda", dtype=torch.float32, seed: int = 0) -> Sam:
    cfg = SAM_ARCHS[arch]
    if state_dict is None:
        if checkpoint is not None:
            state_dict = torch.load(checkpoint, map_location="cpu", weights_only=True)
        else:
            state_dict = sam_state_dict(cfg, seed)            # no checkpoint offline: seeded synthetic weights
    return Sam(cfg, state_dict, device=device, dtype=dtype)


def build_sam_vit_h(checkpoint=None, **kw):
    return _build("vit_h", checkpoint, **kw)


def build_sam_vit_l(checkpoint=None, **kw):
    return _build("vit_l", checkpoint, **kw)


def build_sam_vit_b(checkpoint=None, **kw):
    return _build("vit_b", checkpoint, **kw)


build_sam = build_sam_vit_h
sam_model_registry = {"default": build_sam_vit_h, "vit_h": build_sam_vit_h, "vit_l": build_sam_vit_l, "vit_b": build_sam_vit_b,
                      "tiny": lambda checkpoint=None, **kw: _build("tiny", checkpoint, **kw),
                      "tiny80": lambda checkpoint=None, **kw: _build("tiny80", checkpoint, **kw),
                      "geom": lambda checkpoint=None, **kw: _build("geom", checkpoint, **kw),
                      "win14": lambda checkpoint=None, **kw: _build("win14", checkpoint, **kw)}
