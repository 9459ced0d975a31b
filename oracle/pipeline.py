"""fp32 CPU restatement of the MARS scoring pipeline (test infrastructure only).

Follows /root/reference/mars/MARS.py:59-102 and the four components it calls.
Models are passed in as state dicts + shape keywords and evaluated with
``oracle.vit``; nothing here touches the GPU.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import vit
from .emd import emd2_uniform

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _normalize(img: torch.Tensor, mean, std) -> torch.Tensor:
    m = torch.tensor(mean, dtype=img.dtype).view(-1, 1, 1)
    s = torch.tensor(std, dtype=img.dtype).view(-1, 1, 1)
    return (img - m) / s


def minmax(x: torch.Tensor) -> torch.Tensor:
    return (x - x.min()) / (1e-7 + x.max() - x.min())


# ----------------------------------------------------------------------------
# prior-information refinement
# ----------------------------------------------------------------------------
def scoremap_boxes(scoremap: np.ndarray, threshold: float) -> np.ndarray:
    """Boxes of the thresholded prior, /root/reference/mars/components/
    PriorInformationRefinementModule.py:91-122.

    ``cv2.findContours(RETR_TREE)`` + ``boundingRect`` is restated as "bounding box of
    every 8-connected component" (hole contours lie inside their component's box, so
    the filled union is the same).  opencv is absent here: parity unpinned against
    the wheel; the restatement is checked against scipy.ndimage.label in the tests.
    """
    h, w = scoremap.shape
    img = (scoremap * 255).astype(np.uint8)
    thr = int(threshold * np.max(img))
    binary = img > thr
    if not binary.any():
        return np.asarray([[0, 0, 0, 0]])
    # 8-connected labelling by flood fill (plain python: maps are <= 37x37)
    labels = np.zeros((h, w), dtype=np.int32)
    boxes = []
    cur = 0
    for y in range(h):
        for x in range(w):
            if binary[y, x] and labels[y, x] == 0:
                cur += 1
                stack = [(y, x)]
                labels[y, x] = cur
                y0 = y1 = y
                x0 = x1 = x
                while stack:
                    cy, cx = stack.pop()
                    y0, y1, x0, x1 = min(y0, cy), max(y1, cy), min(x0, cx), max(x1, cx)
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            ny, nx = cy + dy, cx + dx
                            if 0 <= ny < h and 0 <= nx < w and binary[ny, nx] and labels[ny, nx] == 0:
                                labels[ny, nx] = cur
                                stack.append((ny, nx))
                bx1 = min(x1 + 1, w - 1)
                by1 = min(y1 + 1, h - 1)
                boxes.append([x0, y0, bx1, by1])
    return np.asarray(boxes)


def box_mask(prior: torch.Tensor, threshold: float) -> torch.Tensor:
    boxes = scoremap_boxes(prior.cpu().numpy(), threshold)
    B = torch.zeros(prior.shape[0], prior.shape[1])
    for x0, y0, x1, y1 in boxes:
        B[y0:y1, x0:x1] = 1
    return B


def pir_refine(prior: torch.Tensor, mean_attn: torch.Tensor, threshold: float) -> torch.Tensor:
    """PriorInformationRefinementModule.compute, lines 53-89, given the already averaged
    attention map A (lines 31-45)."""
    B = box_mask(prior, threshold).reshape(1, -1)
    D = mean_attn / mean_attn.sum(dim=0, keepdim=True)
    D = D / D.sum(dim=1, keepdim=True)
    R = torch.max(D, D @ D.t())
    R = R @ R
    return ((R * B) @ prior.reshape(-1, 1)).reshape(prior.shape)


def mean_attention(attn_maps: Sequence[torch.Tensor], last_n: int, num_regs: int) -> torch.Tensor:
    """Lines 31-45: drop cls/register rows and columns, keep the last n blocks, mean
    over blocks and heads."""
    if attn_maps[0].dim() == 4:
        a = torch.stack([aw[0, :, 1 + num_regs:, 1 + num_regs:] for aw in attn_maps], dim=0)[-last_n:]
    else:
        a = torch.stack([aw[:, 1 + num_regs:, 1 + num_regs:] for aw in attn_maps], dim=0)[-last_n:]
    return a.mean(dim=(0, 1)).float()


# ----------------------------------------------------------------------------
# visual-visual alignment
# ----------------------------------------------------------------------------
def pool_mask(mask: torch.Tensor, g: int) -> torch.Tensor:
    """adaptive_max_pool2d to g x g; for 518 -> 37 this is a 14x14 max pool."""
    return F.adaptive_max_pool2d(mask.float(), (g, g))


def vva_compute(dino_sd, dino_kw: dict, support_imgs, support_masks, query_img, *,
                box_threshold: float = 0.8, last_n: int = 24):
    """/root/reference/mars/components/VisualVisualAlignmentModule.py:42-127.
    Returns dict(vva, vva_raw, similarity, cost, mean_attn)."""
    g = query_img.shape[-1] // dino_kw["patch"]
    regs = dino_kw["num_regs"]

    def feats(imgs):
        x = torch.stack([_normalize(i, IMAGENET_MEAN, IMAGENET_STD) for i in imgs])
        t = vit.dino_forward(dino_sd, x, **dino_kw)[:, 1 + regs:]
        return F.normalize(t.reshape(-1, t.shape[-1]), p=2, dim=1)

    fs = feats(support_imgs[0])
    xq = torch.stack([_normalize(query_img[0], IMAGENET_MEAN, IMAGENET_STD)])
    tq, attn = vit.dino_forward(dino_sd, xq, want_attn=True, **dino_kw)
    fq = F.normalize(tq[:, 1 + regs:].reshape(-1, tq.shape[-1]), p=2, dim=1)
    S = fs @ fq.t()
    C = (1 - S) / 2
    pooled = pool_mask(support_masks.permute(1, 0, 2, 3), g).flatten().bool()
    S_fg, S_bg = S[pooled], S[~pooled]
    vva = S_fg.mean(dim=0) * S_fg.max(dim=0).values
    if S_bg.shape[0] != 0:
        vva = vva - S_bg.mean(dim=0) * S_bg.max(dim=0).values
    vva = minmax(vva.reshape(g, g))
    A = mean_attention(attn, last_n, regs)
    refined = minmax(pir_refine(vva, A, box_threshold))
    return dict(vva=refined, vva_raw=vva, similarity=S, cost=C, mean_attn=A, support_feats=fs, query_feats=fq)


# ----------------------------------------------------------------------------
# visual-text alignment
# ----------------------------------------------------------------------------
def clip_preprocess(img: torch.Tensor, size: int) -> torch.Tensor:
    """torchvision ``Resize((size,size), BICUBIC)`` on a float tensor followed by the
    CLIP normalisation (/root/reference/mars/components/VisualTextAlignmentModule.py:
    88-92).  The tensor path of torchvision lowers to ``F.interpolate(mode='bicubic',
    align_corners=False, antialias=True)``; torchvision itself is absent (parity
    unpinned against the wheel)."""
    r = F.interpolate(img[None], size=(size, size), mode="bicubic", align_corners=False, antialias=True)[0]
    return _normalize(r, CLIP_MEAN, CLIP_STD)


def vta_text_feats(clip_sd, clip_kw: dict, tok_fg: torch.Tensor, tok_bg: torch.Tensor) -> torch.Tensor:
    """/root/reference/mars/components/SoftmaxGradCAM.py:63-109 for one label and one
    prompt each: normalise, mean over the (single) prompt, normalise again."""
    out = []
    for tok in (tok_fg, tok_bg):
        e = vit.clip_encode_text(clip_sd, tok, layers=clip_kw["text_layers"], heads=clip_kw["text_heads"])
        e = e / e.norm(dim=-1, keepdim=True)
        e = e.mean(dim=0)
        out.append(e / e.norm())
    return torch.stack(out, dim=0)  # [2, E]: foreground first


def vta_compute(clip_sd, clip_kw: dict, query_img, tok_fg, tok_bg, *, box_threshold: float = 0.4,
                last_n: int = 8):
    """/root/reference/mars/components/VisualTextAlignmentModule.py:49-74 +
    SoftmaxGradCAM.py:123-175.  Returns dict(vta, cam, mean_attn, logits)."""
    patch = clip_kw["patch"]
    size = int(math.ceil(query_img.shape[-1] / patch) * patch)
    x = clip_preprocess(query_img[0], size)[None]
    feats, attn = vit.clip_encode_image(clip_sd, x, layers=clip_kw["layers"], heads=clip_kw["heads"], patch=patch)
    text = vta_text_feats(clip_sd, clip_kw, tok_fg, tok_bg)
    cam, attn_last, logits = vit.gradcam(clip_sd, feats, text, layers=clip_kw["layers"], heads=clip_kw["heads"])
    attn = attn + [attn_last]
    A = mean_attention(attn, last_n, 0)
    refined = pir_refine(cam, A, box_threshold)
    return dict(vta=refined, cam=cam, mean_attn=A, logits=logits, text=text)


# ----------------------------------------------------------------------------
# filtering / merging
# ----------------------------------------------------------------------------
def alphaclip_image_transform(img: torch.Tensor, size: int) -> torch.Tensor:
    """Image branch of /root/reference/utils/backbone_loader.py:172-180 as driven by
    /root/reference/mars/components/FilteringMergingModule.py:188-189: float HWC numpy
    -> ToPILImage (x255, truncated to uint8) -> PIL bicubic resize of the short side
    -> centre crop -> ToTensor -> CLIP normalisation."""
    from PIL import Image
    arr = (img.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
    pil = Image.fromarray(arr, mode="RGB")
    w, h = pil.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    pil = pil.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    pil = pil.crop((left, top, left + size, top + size))
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float() / 255.0
    return _normalize(t, CLIP_MEAN, CLIP_STD)


def alphaclip_mask_transform(mask: torch.Tensor, size: int) -> torch.Tensor:
    """Mask branch (backbone_loader.py:182-186, FilteringMergingModule.py:194): uint8
    0/255 -> ToTensor -> bilinear antialiased resize to (size,size) -> (x-0.5)/0.26."""
    m = ((mask.cpu().numpy() * 255).astype(np.uint8)).astype(np.float32) / 255.0
    t = torch.from_numpy(m)[None, None]
    r = F.interpolate(t, size=(size, size), mode="bilinear", align_corners=False, antialias=True)[0]
    return (r - 0.5) / 0.26


def fmm_scores(aclip_sd, aclip_kw: dict, query_img, mask_proposals, support_mask, cost, g: int, vva, vta,
               text_tokens, *, alpha: float = 0.85, batch: int = 8):
    """/root/reference/mars/components/FilteringMergingModule.py:59-140.  Returns a dict
    with the four per-proposal score vectors, the combined score and the ranking."""
    pooled_support = pool_mask(support_mask.permute(1, 0, 2, 3), g)

    text = vit.clip_encode_text(aclip_sd, text_tokens, layers=aclip_kw["text_layers"], heads=aclip_kw["text_heads"])
    text = text / text.norm(dim=-1, keepdim=True)
    size = aclip_kw["resolution"]
    image = alphaclip_image_transform(query_img[0], size)[None]
    feats = []
    for i in range(0, mask_proposals.shape[0], batch):
        a = torch.stack([alphaclip_mask_transform(m, size) for m in mask_proposals[i:i + batch]])
        f = vit.alphaclip_visual(aclip_sd, image.repeat(a.shape[0], 1, 1, 1), a, layers=aclip_kw["layers"],
                                 heads=aclip_kw["heads"], patch=aclip_kw["patch"])
        feats.append(f / f.norm(dim=-1, keepdim=True))
    feats = torch.cat(feats, dim=0)
    aclip = (feats @ text.t())[:, 0].numpy()

    sup = pooled_support.flatten().bool()
    emd = []
    for m in mask_proposals:
        pm = pool_mask(m[None], g)[0].numpy() > 0
        M = cost[sup, :][:, torch.from_numpy(pm).flatten()]
        emd.append(1 - emd2_uniform(M.numpy().astype(np.float64)))
    out = combine_scores(mask_proposals, g, vva, vta, np.asarray(emd), aclip, alpha=alpha)
    out.update(feats=feats, text=text)
    return out


def combine_scores(mask_proposals, g: int, vva, vta, emd, aclip, *, alpha: float = 0.85):
    """Lines 70-81 and 103-138 of FilteringMergingModule._score_proposals given the two model-dependent
    lists: ``emd`` = 1 - emd2 per proposal (float64, what ``_compute_emd`` returns) and ``aclip`` = the
    float32 cosines of line 97.  Dtypes follow the reference's numpy expressions: the pooled sums are
    float32 sums divided in float64, the EMD min-max is float64, the AlphaCLIP min-max stays float32
    (float32 array entries with a Python-float epsilon), the mean of the four is float64."""
    vta_np = vta.detach().cpu().numpy()
    vva_np = vva.detach().cpu().numpy()
    union = (mask_proposals.sum(dim=0) > 0).float()
    pooled_union = pool_mask(union[None], g)[0].numpy() > 0
    emd = [np.float64(e) for e in emd]
    aclip = [np.float32(a) for a in aclip]
    pvv, pvt = [], []
    for m in mask_proposals:
        pm = pool_mask(m[None], g)[0].numpy() > 0
        cnt = np.sum(pm)
        cov = cnt / (1e-7 + np.sum(pooled_union))
        a_vv = np.sum(vva_np[pm]) / (1e-7 + cnt)
        a_vt = np.sum(vta_np[pm]) / (1e-7 + cnt)
        pvv.append(alpha * a_vv + (1 - alpha) * cov)
        pvt.append(alpha * a_vt + (1 - alpha) * cov)
    emd_s = [(s - min(emd)) / (1e-7 + max(emd) - min(emd)) for s in emd]
    eps32 = np.float32(1e-7)
    acl_s = [(s - min(aclip)) / (eps32 + max(aclip) - min(aclip)) for s in aclip]
    assert all(a.dtype == np.float32 for a in acl_s)
    score = [(emd_s[i] + np.float64(acl_s[i]) + pvv[i] + pvt[i]) / 4 for i in range(len(emd))]
    order = sorted(range(len(score)), key=lambda i: score[i], reverse=True)  # stable, like sorted()
    return dict(emd=np.asarray(emd), alphaclip=np.asarray(aclip, dtype=np.float64), pvv=np.asarray(pvv), pvt=np.asarray(pvt),
                emd_scaled=np.asarray(emd_s), alphaclip_scaled=np.asarray(acl_s, dtype=np.float64), score=np.asarray(score),
                order=np.asarray(order))


def fmm_merge(mask_proposals, score: np.ndarray, order: np.ndarray, *, static_threshold: float = 0.55,
              dynamic_threshold: float = 0.95):
    """FilteringMergingModule._merge_masks, lines 209-221.  Returns (mask, kept indices)."""
    top = score[order[0]]
    if top < static_threshold:
        lb = dynamic_threshold * top
        keep = [int(i) for i in order if score[i] >= lb]
    else:
        keep = [int(i) for i in order if score[i] >= static_threshold]
    merged = (torch.stack([mask_proposals[i] for i in keep]).sum(dim=0) > 0).float()
    return merged, keep


# ----------------------------------------------------------------------------
# whole ranking path + evaluation
# ----------------------------------------------------------------------------
def alphaclip_prompt(name: str, description: str) -> List[str]:
    """/root/reference/mars/MARS.py:85-89."""
    return [f"a {name}."] if description == "" else [f"a {name}, {description}."]


def mars_predict(models: dict, batch: dict, tokens: dict, *, vva_thr=0.8, vva_last_n=24, vta_thr=0.4,
                 vta_last_n=8, alpha=0.85, static_threshold=0.55, dynamic_threshold=0.95, aclip_batch=8):
    """/root/reference/mars/MARS.py:59-102 with the text retriever replaced by fixed
    token ids (``tokens`` = dict(fg, bg, alphaclip))."""
    v = vva_compute(models["dino_sd"], models["dino_kw"], batch["support_imgs"], batch["support_masks"],
                    batch["query_img"], box_threshold=vva_thr, last_n=vva_last_n)
    t = vta_compute(models["clip_sd"], models["clip_kw"], batch["query_img"], tokens["fg"], tokens["bg"],
                    box_threshold=vta_thr, last_n=vta_last_n)
    g = v["vva"].shape[0]
    vta = F.interpolate(t["vta"][None, None], (g, g), mode="nearest")[0, 0]
    vta = (vta - vta.min()) / (1e-7 + vta.max() - vta.min())
    s = fmm_scores(models["aclip_sd"], models["aclip_kw"], batch["query_img"], batch["mask_proposals"],
                   batch["support_masks"], v["cost"], g, v["vva"], vta, tokens["alphaclip"], alpha=alpha,
                   batch=aclip_batch)
    mask, keep = fmm_merge(batch["mask_proposals"], s["score"], s["order"], static_threshold=static_threshold,
                           dynamic_threshold=dynamic_threshold)
    return dict(mask=mask, keep=keep, vva=v, vta=t, vta_scaled=vta, scores=s)


def classify_prediction(pred: torch.Tensor, gt: torch.Tensor, ignore: Optional[torch.Tensor] = None):
    """/root/reference/mars/utils/evaluation.py:11-39 for one episode ([1,H,W] masks)."""
    pred = pred.clone()
    if ignore is not None:
        gt = gt + ignore * 255
        pred[gt == 255] = 255
    inter = pred[pred == gt]
    a_i = torch.histc(inter, bins=2, min=0, max=1) if inter.numel() else torch.zeros(2)
    a_p = torch.histc(pred, bins=2, min=0, max=1)
    a_g = torch.histc(gt, bins=2, min=0, max=1)
    return a_i, a_p + a_g - a_i


def compute_iou(inter_buf: torch.Tensor, union_buf: torch.Tensor, class_ids: Sequence[int]):
    """/root/reference/mars/utils/logger.py:68-78."""
    ids = torch.as_tensor(list(class_ids), dtype=torch.long)
    iou = inter_buf / torch.max(union_buf, torch.ones_like(union_buf))
    iou = iou.index_select(1, ids)
    miou = iou[1].mean() * 100
    fb = (inter_buf.index_select(1, ids).sum(dim=1) / union_buf.index_select(1, ids).sum(dim=1)).mean() * 100
    return float(miou), float(fb)
