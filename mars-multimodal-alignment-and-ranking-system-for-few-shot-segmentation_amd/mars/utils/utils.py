"""Small helpers of the test loop (/root/reference/mars/utils/utils.py:8-40)."""
from __future__ import annotations

import random

import numpy as np
import torch


def fix_randseed(seed):
    """Seeds numpy and torch (host and every GPU); ``None`` draws a seed from ``random``."""
    if seed is None:
        seed = int(random.random() * 1e5)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def mean(x):
    return sum(x) / len(x) if len(x) > 0 else 0.0


def to_cuda(batch: dict, device="cuda") -> dict:
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def to_cpu(tensor: torch.Tensor) -> torch.Tensor:
    return tensor.detach().clone().cpu()
