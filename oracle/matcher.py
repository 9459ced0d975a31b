"""TEST INFRASTRUCTURE ONLY - CPU restatement of the Matcher scoring pieces and of
CLIPPipeline.process_masks (plain torch / numpy, fp32).  Imported by tests only.

Pinning: the functions below sit on third-party arithmetic that is absent here (cv2.resize
INTER_AREA, POT ot.emd2, torchvision transforms) and the reference classes themselves cannot be
imported (missing dependencies), so this part of the oracle is PARITY UNPINNED: it follows the
reference text line by line and re-uses the pinned pieces (DINOv2 / AlphaCLIP towers, exact EMD).
cv2.resize(INTER_AREA) from 518 to 37 is an exact 14 x 14 block mean (integer ratio).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import vit
from .emd import emd2_uniform
from .pipeline import IMAGENET_MEAN, IMAGENET_STD, _normalize, alphaclip_image_transform, alphaclip_mask_transform


def reference_masks_verification(masks: torch.Tensor) -> torch.Tensor:
    """/root/reference/matcher/Matcher.py:141-154: an empty reference mask gets a centred 14x14 square."""
    masks = masks.clone()
    if masks.sum() == 0:
        sh, sw = masks.shape[-2:]
        masks[..., (sh // 2 - 7):(sh // 2 + 7), (sw // 2 - 7):(sw // 2 + 7)] = 1
    return masks


def set_reference_pool(masks: torch.Tensor, patch: int, mask_threshold: float = 0.0) -> torch.Tensor:
    """Matcher.set_reference, lines 168-180: masks [1, ns, H, W] -> flat {0,1} vector [ns * G * G]."""
    m = reference_masks_verification(masks).permute(1, 0, 2, 3)
    pooled = F.avg_pool2d(m.float(), (patch, patch))
    return (pooled > mask_threshold).float().reshape(-1)


def extract_img_feats(dino_sd, dino_kw: dict, ref_imgs: torch.Tensor, tar_img: torch.Tensor, use_registers: bool = True):
    """Matcher.extract_img_feats, lines 266-302: unit-length patch features, ref [ns*P, D] and tar [P, D]."""
    num_regs = dino_kw["num_regs"]
    skip = 1 + (num_regs if use_registers else 0)

    def feats(x):
        x = torch.stack([_normalize(i, IMAGENET_MEAN, IMAGENET_STD) for i in x])
        tok = vit.dino_forward(dino_sd, x, **dino_kw)[:, skip:]
        return F.normalize(tok.reshape(-1, tok.shape[-1]), dim=1, p=2)
    return feats(ref_imgs), feats(tar_img)


def similarity_and_cost(ref_feats: torch.Tensor, tar_feat: torch.Tensor):
    """patch_level_matching, lines 437-440."""
    S = ref_feats @ tar_feat.t()
    return S, (1 - S) / 2


def get_mask_scores(mask: np.ndarray, all_points: np.ndarray, emd_cost: torch.Tensor, ref_masks_pool: torch.Tensor, g: int):
    """RobustPromptSampler.get_mask_scores, lines 1167-1210, for one mask [1, H, W] (bool / 0-1).
    Returns (purity, coverage, emd_score)."""
    ori = mask[0]
    h, w = ori.shape
    small = ori.astype(np.float32).reshape(g, h // g, g, w // g).mean(axis=(1, 3))     # INTER_AREA at an integer ratio
    thres = small.max() - 1e-6 if small.max() <= 0 else 0
    small = small > thres
    rows = ref_masks_pool.flatten().bool()
    M = emd_cost[rows, :][:, torch.from_numpy(small.flatten())]
    emd_score = 1 - emd2_uniform(M.cpu().numpy().astype(np.float64))
    pts = all_points.astype(np.int64)[:, ::-1]                      # (y, x)
    pts = np.clip(pts, 0, [h - 1, w - 1])
    inside = int(np.count_nonzero(ori[pts[:, 0], pts[:, 1]]))
    area = max(float(small.sum()), 1.0)
    purity = np.float32(inside / area) + np.float32(1e-6)           # torch.tensor([...]) is float32
    coverage = np.float32(inside / all_points.shape[0]) + np.float32(1e-6)
    return float(purity), float(coverage), float(emd_score)


def matcher_scores(masks: np.ndarray, all_points: np.ndarray, emd_cost, ref_masks_pool, g: int, alpha: float, beta: float, exp: float):
    """mask_generation, lines 698-720: per-mask metrics and score = alpha*emd + beta*purity*coverage**exp (float32)."""
    pur, cov, emd = [], [], []
    for m in masks:
        p, c, e = get_mask_scores(m, all_points, emd_cost, ref_masks_pool, g)
        pur.append(p); cov.append(c); emd.append(e)
    pur_t, cov_t, emd_t = (torch.tensor(v, dtype=torch.float32) for v in (pur, cov, emd))
    return dict(purity=pur_t, coverage=cov_t, emd=emd_t, scores=alpha * emd_t + beta * pur_t * cov_t ** exp)


def clip_pipeline_process_masks(aclip_sd, aclip_kw: dict, target_img: torch.Tensor, masks: Sequence[np.ndarray],
                                text_tokens: torch.Tensor) -> Dict[int, np.ndarray]:
    """/root/reference/mars/ClipPipeline.py:136-171 (fp32): idx -> [1, P] array of 100 * cos(image|mask, prompt)."""
    size = aclip_kw["resolution"]
    image = alphaclip_image_transform(target_img, size)[None]
    text = vit.clip_encode_text(aclip_sd, text_tokens, layers=aclip_kw["text_layers"], heads=aclip_kw["text_heads"])
    text = text / text.norm(dim=-1, keepdim=True)
    out = {}
    for idx, m in enumerate(masks):
        alpha = alphaclip_mask_transform(torch.from_numpy(np.asarray(m, dtype=np.float32)), size)[None]
        f = vit.alphaclip_visual(aclip_sd, image, alpha, layers=aclip_kw["layers"], heads=aclip_kw["heads"], patch=aclip_kw["patch"])
        f = f / f.norm(dim=-1, keepdim=True)
        out[idx] = (100.0 * f @ text.t()).numpy()
    return out
