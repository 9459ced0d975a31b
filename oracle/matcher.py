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


# ----------------------------------------------------------------------------------------
# prompt-sampling stage (Matcher.predict): scipy's linear_sum_assignment is the reference's own solver,
# so the assignment step is pinned; the rest follows /root/reference/matcher/Matcher.py
# ----------------------------------------------------------------------------------------
def _inds_to_points(inds: torch.Tensor, g: int, patch: int, input_size) -> np.ndarray:
    """Lines 512-547: duplicates removed through a Python set, patch centres, points outside dropped."""
    uniq = torch.tensor(list(set(inds.cpu().tolist())))
    xs = ((uniq % g) * patch + patch // 2).tolist()
    ys = ((uniq // g) * patch + patch // 2).tolist()
    out = []
    for x, y in zip(xs, ys):
        if int(x) < input_size[1] and int(y) < input_size[0]:
            out.append([int(x), int(y)])
    return np.array(out)


def negatives_from_discarded(idxs_forward, sim_scores_forward, idxs_reverse, idxs_mask, g, patch, input_size):
    """Lines 304-348."""
    discarded_ind = torch.isin(idxs_reverse[1], idxs_mask, invert=True)
    if (discarded_ind == False).all().item():  # noqa: E712
        return None, None
    cols = idxs_forward[1][discarded_ind]
    sims = sim_scores_forward[discarded_ind]
    n = len(sims) // 2 if len(sims) > 40 else len(sims)
    _, order = torch.sort(sims, descending=False)
    return _inds_to_points(cols[order[:n]], g, patch, input_size), n


def negatives_from_cost(C: torch.Tensor, idxs_mask, g, patch, input_size):
    """Lines 350-417 (the sorted positions of the filtered costs index the unfiltered column list, as there)."""
    from scipy.optimize import linear_sum_assignment
    fwd = [torch.as_tensor(i, dtype=torch.int64) for i in linear_sum_assignment(C, maximize=True)]
    cost_f = C[fwd[0], fwd[1]]
    rev = [torch.as_tensor(i, dtype=torch.int64) for i in linear_sum_assignment(C.t()[fwd[1]], maximize=True)]
    retain = torch.isin(rev[1], idxs_mask, invert=True)
    if not (retain == False).all().item():  # noqa: E712
        cost_f = cost_f[retain]
    n = len(cost_f) // 2 if len(cost_f) > 40 else len(cost_f)
    _, order = torch.sort(cost_f, descending=True)
    return _inds_to_points(fwd[1][order[:n]], g, patch, input_size), n


def points_box(points: np.ndarray, input_size) -> np.ndarray:
    """Lines 567-573: the box prompt is the bounding box of the matched points."""
    return np.array([max(points[:, 0].min(), 0), max(points[:, 1].min(), 0),
                     min(points[:, 0].max(), input_size[1] - 1), min(points[:, 1].max(), input_size[0] - 1)])


def patch_level_matching(S: torch.Tensor, ref_masks_pool: torch.Tensor, g: int, patch: int, input_size,
                         neg_from_discarded: bool = False, neg_from_cost: bool = False):
    """Lines 419-577 (the box is ``points_box``).  S [ns*P, P] on the CPU.  Returns (points, negatives-or-discarded, C, n, n_neg)."""
    from scipy.optimize import linear_sum_assignment
    C = (1 - S) / 2
    fg = ref_masks_pool.flatten().bool()
    S_forward = S[fg]
    fwd = [torch.as_tensor(i, dtype=torch.int64) for i in linear_sum_assignment(S_forward, maximize=True)]
    sim_f = S_forward[fwd[0], fwd[1]]
    idx_mask = ref_masks_pool.flatten().nonzero()[:, 0]
    S_reverse = S.t()[fwd[1]]
    rev = [torch.as_tensor(i, dtype=torch.int64) for i in linear_sum_assignment(S_reverse, maximize=True)]
    retain = torch.isin(rev[1], idx_mask)
    pos, neg, sim_pos = fwd, fwd, sim_f.clone()
    if not (retain == False).all().item():  # noqa: E712
        pos = [fwd[0][retain], fwd[1][retain]]
        neg = [fwd[0][~retain], fwd[1][~retain]]
        sim_pos = sim_f[retain]
    n = len(sim_pos) // 2 if len(sim_pos) > 40 else len(sim_pos)
    _, order = torch.sort(sim_pos, descending=True)
    points = _inds_to_points(pos[1][order[:n]], g, patch, input_size)
    discarded = _inds_to_points(neg[1], g, patch, input_size)
    priors, n_neg = [], []
    if neg_from_discarded:
        p, k = negatives_from_discarded(fwd, sim_f, rev, idx_mask, g, patch, input_size)
        priors.append(p); n_neg.append(k)
    if neg_from_cost:
        p, k = negatives_from_cost(C, idx_mask, g, patch, input_size)
        priors.append(p); n_neg.append(k)
    return points, (priors if len(priors) > 0 else discarded), C, n, n_neg


def _combinations(n: int, k: int):
    """Lines 1212-1224."""
    if k > n:
        return []
    if k == 0:
        return [[]]
    if k == n:
        return [[i for i in range(n)]]
    res = []
    for i in range(n):
        for j in _combinations(i, k - 1):
            res.append(j + [i])
    return res


def sample_points(points, negative_points=None, sample_range=(4, 6), max_iterations: int = 30):
    """Lines 1226-1295; draws from the ``random`` module in the reference's order."""
    import random
    sample_list, label_list = [], []
    for i in range(min(sample_range[0], len(points)), min(sample_range[1], len(points)) + 1):
        sample_neg = label_neg = None
        if len(points) > 8:
            index = [random.sample(range(len(points)), i) for j in range(max_iterations)]
            sample = np.take(points, index, axis=0)
            if negative_points is not None:
                if len(negative_points) > 8:
                    index_neg = [random.sample(range(len(negative_points)), i) for j in range(max_iterations)]
                else:
                    index_neg = [random.choices(range(len(negative_points)), k=i) for j in range(max_iterations)]
                sample_neg = np.take(negative_points, index_neg, axis=0)
                label_neg = np.zeros((sample_neg.shape[0], i))
        else:
            index = _combinations(len(points), i)
            sample = np.take(points, index, axis=0)
            if negative_points is not None:
                index_neg = [random.choices(range(len(negative_points)), k=i) for j in range(len(index))]
                sample_neg = np.take(negative_points, index_neg, axis=0)
                label_neg = np.zeros((sample_neg.shape[0], i))
        sample_list.append(sample)
        label_list.append(np.ones((sample.shape[0], i)))
        if sample_neg is not None:
            sample_list.append(sample_neg)
            label_list.append(label_neg)
    if negative_points is not None:
        pts = [np.hstack((sample_list[i], sample_list[i + 1])) for i in range(0, len(sample_list), 2)]
        lbs = [np.hstack((label_list[i], label_list[i + 1])) for i in range(0, len(label_list), 2)]
        return pts, lbs
    return sample_list, label_list


def mask_generation(sam_sd, sam_kw: dict, gen_kw: dict, tar_img_np: np.ndarray, points, all_points, ref_masks_pool, C, g: int,
                    negative_points=None, *, use_neg: bool = False, merge_prompt_types: bool = False, sample_range=(4, 6),
                    max_iterations: int = 30, alpha=1.0, beta=0.0, exp=0.0, score_filter_cfg=None, num_merging_mask=10, box=None,
                    target_mask=None):
    """Lines 619-834.  Returns (merged [1, H, W] float, score, unfiltered masks, merged-from)."""
    from . import sam as osam
    cfg = {"emd": 0.0, "purity": 0.0, "coverage": 0.0, "score_filter": False, "score": 0.33, "score_norm": 0.1,
           "topk_scores_threshold": 0.0}
    cfg.update(score_filter_cfg or {})
    samples_list, label_list = [], []
    if use_neg:
        for negative_pts in negative_points:
            if negative_pts is not None and len(negative_pts) > 0:
                s, l = sample_points(points, negative_pts, sample_range, max_iterations)
            else:
                s, l = sample_points(points, None, sample_range, max_iterations)
            samples_list.extend(s); label_list.extend(l)
        if merge_prompt_types:
            s, l = sample_points(points, None, sample_range, max_iterations)
            samples_list.extend(s); label_list.extend(l)
    else:
        samples_list, label_list = sample_points(points, None, sample_range, max_iterations)
    recs = osam.generate(sam_sd, sam_kw, tar_img_np, select_point_coords=samples_list, select_point_labels=label_list,
                         select_box=[box] if box is not None else None,
                         select_mask_input=target_mask.cpu().numpy() if target_mask is not None else None, **gen_kw)
    tar_masks = np.stack([r["segmentation"] for r in recs])[:, None] > 0
    coords = [r["point_coords"] for r in recs]
    sc = matcher_scores(tar_masks, all_points, C, ref_masks_pool, g, alpha, beta, exp)
    metric = {k: sc[k] for k in ("purity", "coverage", "emd")}
    scores = sc["scores"]
    pred = tar_masks[:, 0]
    unfiltered = torch.tensor(pred.copy() > 0, dtype=torch.float)
    for name in ["coverage", "emd", "purity"]:
        if cfg[name] > 0:
            thres = min(cfg[name], metric[name].max())
            idx = torch.where(metric[name] >= thres)[0]
            scores, pred = scores[idx], pred[idx.numpy()]
            coords = [coords[int(i)] for i in idx]
            metric = {k: v[idx] for k, v in metric.items()}
    if cfg["score_filter"]:
        distances, rank = torch.sort(1 - scores, descending=False)
        norm = (distances - distances.min()) / (distances.max() + 1e-6)
        f = distances < cfg["score"]
        f[..., 0] = True
        f = f * (norm < cfg["score_norm"])
        chosen = rank[f][:num_merging_mask]
        final = scores[chosen].mean()
    else:
        topk = min(num_merging_mask, scores.size(0))
        topk_idx = scores.topk(topk)[1]
        topk_scores = scores[topk_idx].numpy()
        if cfg["topk_scores_threshold"] > 0:
            topk_scores = topk_scores / topk_scores.max()
        keep = topk_scores > cfg["topk_scores_threshold"]
        chosen = topk_idx[torch.from_numpy(keep)]
        final = topk_scores[keep].mean()
    merged_from = pred[chosen.numpy()].reshape(-1, pred.shape[-2], pred.shape[-1])
    merged = merged_from.sum(0) > 0
    return torch.tensor(merged[None], dtype=torch.float), float(final), unfiltered, torch.tensor(merged_from.copy() > 0, dtype=torch.float)
