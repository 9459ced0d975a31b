"""Per-class IoU accumulation (/root/reference/mars/utils/logger.py:12-78, AverageMeter only;
the file / tensorboard / Comet sinks of the reference are not part of the ranking path)."""
from __future__ import annotations

from typing import Sequence

import torch

NCLASS = {"pascal": 20, "pascal5i": 20, "coco": 80, "fss": 1000, "paco_part": 448, "pascal_part": 100, "lvis": 1203}


def all_reduce_(t: torch.Tensor, op=None) -> torch.Tensor:
    """In-place all-reduce of ``t`` over the default group.  RCCL reduces device tensors where they are; a gloo
    group (CPU runs, rehearsals of several ranks on one card) gets a host copy and the result is copied back."""
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


class AverageMeter:
    def __init__(self, dataset, device="cuda"):
        """``dataset`` needs ``benchmark`` and ``class_ids`` (the fold's classes), as the reference's
        episode datasets have (logger.py:18-27)."""
        self.benchmark = dataset.benchmark
        self.device = torch.device(device)
        ids = list(dataset.class_ids)
        if self.benchmark == "pascal5i":                     # the PASCAL-5i reader keeps 1-based labels (logger.py:20-22)
            ids = [i - 1 for i in ids]
        self.class_ids_interest = torch.tensor(ids, dtype=torch.long, device=self.device)
        self.nclass = NCLASS[self.benchmark]
        self.intersection_buf = torch.zeros(2, self.nclass, dtype=torch.float32, device=self.device)
        self.union_buf = torch.zeros(2, self.nclass, dtype=torch.float32, device=self.device)
        self.ones = torch.ones_like(self.union_buf)
        self.loss_buf = []

    def update(self, inter_b, union_b, class_id, loss=None):
        class_id = class_id.to(self.device).long()
        self.intersection_buf.index_add_(1, class_id, inter_b.float())
        self.union_buf.index_add_(1, class_id, union_b.float())
        self.loss_buf.append(torch.tensor(0.0) if loss is None else loss)

    def all_reduce(self):
        """Sum the two [2, nclass] buffers over ranks (counts are integers < 2^24, so the float sum
        is exact and order independent): the only collective of a data-parallel run."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            all_reduce_(self.intersection_buf, dist.ReduceOp.SUM)
            all_reduce_(self.union_buf, dist.ReduceOp.SUM)

    def compute_iou(self):
        iou = self.intersection_buf / torch.max(torch.stack([self.union_buf, self.ones]), dim=0)[0]
        iou = iou.index_select(1, self.class_ids_interest)
        miou = iou[1].mean() * 100
        fb_iou = (self.intersection_buf.index_select(1, self.class_ids_interest).sum(dim=1) /
                  self.union_buf.index_select(1, self.class_ids_interest).sum(dim=1)).mean() * 100
        return miou, fb_iou, iou[1][:min(len(iou[1]), 20)]
