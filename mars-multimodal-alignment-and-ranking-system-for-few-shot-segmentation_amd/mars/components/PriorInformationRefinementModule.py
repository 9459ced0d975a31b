"""Attention-affinity refinement of a prior map on the GPU.

Same interface as /root/reference/mars/components/PriorInformationRefinementModule.py:7-122
(``compute(prior, attn_maps)``) plus ``refine(prior, mean_attn)`` for callers that already
hold the head/layer mean of the attention maps.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from ... import ops


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class PriorInformationRefinementModule:
    def __init__(self, box_threshold: float, last_n_attention_maps_for_refinement: int, device, num_regs: int = 0):
        self.threshold = box_threshold
        self.last_n_attention_maps_for_refinement = last_n_attention_maps_for_refinement
        self.device = torch.device(device)
        self.num_regs = num_regs
        self._ws: Dict[int, dict] = {}

    def _buffers(self, P: int) -> dict:
        w = self._ws.get(P)
        if w is None:
            ld = _pad64(P)
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
            w = self._ws[P] = dict(ld=ld, A=z(P, ld), D=z(P, ld), DDT=z(P, ld), R=z(P, ld), RT=z(P, ld), R2=z(P, ld),
                                   scratch=z(33 * P), Bm=z(P), out=z(P))
        return w

    def new_accumulator(self, P: int) -> torch.Tensor:
        """Zeroed [P, pad64(P)] f32 buffer the attention kernels add the mean map into."""
        w = self._buffers(P)
        w["A"].zero_()
        return w["A"]

    def refine(self, prior: torch.Tensor, mean_attn: torch.Tensor) -> torch.Tensor:
        """Lines 53-89 of the reference given A = mean attention ([P, ld] f32, ld = pad64(P)).
        Returns the refined prior with the shape of ``prior`` (a view of an internal buffer)."""
        G = prior.shape[0]
        P = G * G
        w = self._buffers(P)
        ld = w["ld"]
        assert mean_attn.shape[0] >= P and mean_attn.stride(0) == ld, "mean attention must be [P, pad64(P)]"
        prior = prior.contiguous().float()
        ops.box_mask(prior, self.threshold, w["Bm"])
        ops.sinkhorn1(mean_attn, w["D"], P, w["scratch"])
        ops.gemm(w["D"], w["D"], w["DDT"], P, P, ld, lda=ld, ldb=ld, ldc=ld)           # D D^T
        ops.max_and_transpose(w["D"], w["DDT"], P, w["R"], w["RT"])                    # R = max(D, DD^T)
        ops.gemm(w["R"], w["RT"], w["R2"], P, P, ld, lda=ld, ldb=ld, ldc=ld)           # R R
        ops.masked_matvec(w["R2"], w["Bm"], prior, P, w["out"])                        # (R o B) prior
        return w["out"].view(G, G)

    def compute(self, prior: torch.Tensor, attn_maps: Sequence[torch.Tensor]) -> torch.Tensor:
        """Reference signature: ``attn_maps`` is a list of [B,h,T,T] or [B,T,T] probability tensors."""
        prior = prior.to(self.device)
        G = prior.shape[0]
        P = G * G
        acc = self.new_accumulator(P)
        maps = list(attn_maps)[-self.last_n_attention_maps_for_refinement:]
        off = 1 + self.num_regs
        for aw in maps:
            aw = aw.to(self.device).float()
            if aw.dim() == 4:
                aw = aw[0]
            aw = aw.contiguous()
            H, T, _ = aw.shape
            ops.attn_mean_accum(aw, 0, T, T * T, H, off, P, 1.0 / (len(maps) * H), acc)
        return self.refine(prior, acc).clone()
