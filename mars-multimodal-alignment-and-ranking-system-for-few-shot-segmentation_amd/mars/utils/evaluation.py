"""Intersection / union areas of a prediction (/root/reference/mars/utils/evaluation.py:5-39)."""
from __future__ import annotations

import torch

from ... import ops


class Evaluator:
    @classmethod
    def initialize(cls):
        cls.ignore_index = 255

    @classmethod
    def classify_prediction(cls, pred_mask: torch.Tensor, batch: dict):
        """pred_mask [B,H,W] and batch['query_mask'] [B,H,W] (optionally 'query_ignore_idx') ->
        (area_inter [2,B], area_union [2,B]) as float tensors on the prediction's device."""
        gt_mask = batch.get("query_mask").to(pred_mask.device)
        ignore = batch.get("query_ignore_idx")
        if ignore is not None:
            ignore = ignore.to(pred_mask.device)
        B = pred_mask.shape[0]
        counts = torch.zeros(B, 4, dtype=torch.float32, device=pred_mask.device)
        for b in range(B):
            ops.iou_counts(pred_mask[b].contiguous().float(), gt_mask[b].contiguous().float(),
                           ignore[b].contiguous().float() if ignore is not None else None, counts[b])
        return counts[:, 0:2].t(), counts[:, 2:4].t()          # views: [2, B] each
