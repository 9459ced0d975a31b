"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch fp32, functional over a state dict with the
reference's parameter names) of the Segment-Anything proposal generator the reference vendors:
image encoder, prompt encoder, mask decoder, predictor transforms and the automatic mask generator.

Pinning: tests/golden/make_golden.py runs the reference modules (segment_anything.modeling,
importable here) on seeded synthetic weights and asserts equality with these functions; their
outputs are stored in tests/golden/reference_vectors.npz.  torchvision (PIL resize of the
predictor, batched_nms) is absent: the PIL resize is PIL itself, and greedy box NMS is restated
from its published definition (parity unpinned for that one step).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


# ----------------------------------------------------------------------------------------
# image encoder (segment_anything/modeling/image_encoder.py)
# ----------------------------------------------------------------------------------------
def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln2d(x, sd, p, eps=1e-6):
    """common.py:35-43 (normalise over the channel dim of an NCHW tensor)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + ".weight"][:, None, None] * x + sd[p + ".bias"][:, None, None]


def get_rel_pos(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """image_encoder.py:292-322."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def _enc_attention(sd: SD, p: str, x: torch.Tensor, heads: int, use_rel_pos: bool) -> torch.Tensor:
    """image_encoder.py:224-240, 325-361; x [B, H, W, C]."""
    B, H, W, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(B, H * W, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, -1).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if use_rel_pos:
        Rh = get_rel_pos(H, H, sd[p + ".rel_pos_h"])
        Rw = get_rel_pos(W, W, sd[p + ".rel_pos_w"])
        r_q = q.reshape(B * heads, H, W, hd)
        rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
        attn = (attn.view(B * heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B * heads, H * W, H * W)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).view(B, heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def window_partition(x: torch.Tensor, ws: int):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win: torch.Tensor, ws: int, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def image_encoder(sd: SD, x: torch.Tensor, cfg: dict, pre: str = "image_encoder") -> torch.Tensor:
    """ImageEncoderViT.forward (:106-116): [B,3,S,S] normalised + padded -> [B, out_chans, S/16, S/16]."""
    x = F.conv2d(x, sd[pre + ".patch_embed.proj.weight"], sd[pre + ".patch_embed.proj.bias"], stride=cfg["patch"]).permute(0, 2, 3, 1)
    x = x + sd[pre + ".pos_embed"]
    for i in range(cfg["depth"]):
        p = f"{pre}.blocks.{i}"
        ws = 0 if i in cfg["global_attn_indexes"] else cfg["window_size"]
        short = x
        y = _ln(x, sd, p + ".norm1", 1e-6)      # build_sam.py:70: norm_layer = LayerNorm(eps=1e-6)
        if ws > 0:
            H, W = y.shape[1], y.shape[2]
            y, pad_hw = window_partition(y, ws)
        y = _enc_attention(sd, p + ".attn", y, cfg["heads"], cfg.get("use_rel_pos", True))
        if ws > 0:
            y = window_unpartition(y, ws, pad_hw, (H, W))
        x = short + y
        h = F.gelu(F.linear(_ln(x, sd, p + ".norm2", 1e-6), sd[p + ".mlp.lin1.weight"], sd[p + ".mlp.lin1.bias"]))
        x = x + F.linear(h, sd[p + ".mlp.lin2.weight"], sd[p + ".mlp.lin2.bias"])
    x = x.permute(0, 3, 1, 2)
    x = _ln2d(F.conv2d(x, sd[pre + ".neck.0.weight"]), sd, pre + ".neck.1")
    return _ln2d(F.conv2d(x, sd[pre + ".neck.2.weight"], padding=1), sd, pre + ".neck.3")


# ----------------------------------------------------------------------------------------
# prompt encoder (prompt_encoder.py)
# ----------------------------------------------------------------------------------------
def _pe_encoding(sd: SD, coords: torch.Tensor) -> torch.Tensor:
    c = 2 * coords - 1
    c = c @ sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd: SD, size: Tuple[int, int]) -> torch.Tensor:
    """PositionEmbeddingRandom.forward (:194-205) -> [1, C, h, w]."""
    h, w = size
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / h
    x = (grid.cumsum(dim=1) - 0.5) / w
    return _pe_encoding(sd, torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def embed_points(sd: SD, points: torch.Tensor, labels: torch.Tensor, input_image_size, pad: bool = True) -> torch.Tensor:
    """PromptEncoder._embed_points (:73-91): points [B,N,2] in the model's input frame, labels [B,N]."""
    points = points + 0.5
    if pad:
        points = torch.cat([points, torch.zeros((points.shape[0], 1, 2))], dim=1)
        labels = torch.cat([labels, -torch.ones((labels.shape[0], 1))], dim=1)
    c = points.clone()
    c[:, :, 0] = c[:, :, 0] / input_image_size[1]
    c[:, :, 1] = c[:, :, 1] / input_image_size[0]
    e = _pe_encoding(sd, c.to(torch.float))
    e[labels == -1] = 0.0
    e[labels == -1] += sd["prompt_encoder.not_a_point_embed.weight"]
    e[labels == 0] += sd["prompt_encoder.point_embeddings.0.weight"]
    e[labels == 1] += sd["prompt_encoder.point_embeddings.1.weight"]
    return e


def embed_masks(sd: SD, masks: torch.Tensor) -> torch.Tensor:
    """PromptEncoder._embed_masks (:102-105, mask_downscaling :51-58): [B,1,4G,4G] -> [B,C,G,G]."""
    p = "prompt_encoder.mask_downscaling"
    x = F.conv2d(masks.float(), sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2)
    x = F.gelu(_ln2d(x, sd, p + ".1"))
    x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2)
    x = F.gelu(_ln2d(x, sd, p + ".4"))
    return F.conv2d(x, sd[p + ".6.weight"], sd[p + ".6.bias"])


def embed_boxes(sd: SD, boxes: torch.Tensor, input_image_size) -> torch.Tensor:
    """PromptEncoder._embed_boxes (:93-101): boxes [B,4] in the model's input frame -> [B,2,C]."""
    boxes = boxes + 0.5
    c = boxes.reshape(-1, 2, 2).clone()
    c[:, :, 0] = c[:, :, 0] / input_image_size[1]
    c[:, :, 1] = c[:, :, 1] / input_image_size[0]
    e = _pe_encoding(sd, c.to(torch.float))
    e[:, 0, :] += sd["prompt_encoder.point_embeddings.2.weight"]
    e[:, 1, :] += sd["prompt_encoder.point_embeddings.3.weight"]
    return e


# ----------------------------------------------------------------------------------------
# mask decoder (mask_decoder.py, transformer.py)
# ----------------------------------------------------------------------------------------
def _dec_attention(sd: SD, p: str, q, k, v, heads: int):
    """transformer.py:218-240."""
    q = F.linear(q, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"])
    k = F.linear(k, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(v, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])

    def split(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    a = torch.softmax(q @ k.permute(0, 1, 3, 2) / math.sqrt(q.shape[-1]), dim=-1)
    o = (a @ v).transpose(1, 2)
    o = o.reshape(o.shape[0], o.shape[1], -1)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def two_way_transformer(sd: SD, image_embedding, image_pe, point_embedding, cfg: dict, pre="mask_decoder.transformer"):
    """TwoWayTransformer.forward (:62-106) with its blocks (:151-182)."""
    heads = cfg["dec_heads"]
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries = point_embedding
    for i in range(cfg["dec_depth"]):
        p = f"{pre}.layers.{i}"
        if i == 0:
            queries = _dec_attention(sd, p + ".self_attn", queries, queries, queries, heads)
        else:
            q = queries + point_embedding
            queries = queries + _dec_attention(sd, p + ".self_attn", q, q, queries, heads)
        queries = _ln(queries, sd, p + ".norm1")
        q, k = queries + point_embedding, keys + key_pe
        queries = _ln(queries + _dec_attention(sd, p + ".cross_attn_token_to_image", q, k, keys, heads), sd, p + ".norm2")
        m = F.linear(F.relu(F.linear(queries, sd[p + ".mlp.lin1.weight"], sd[p + ".mlp.lin1.bias"])),
                     sd[p + ".mlp.lin2.weight"], sd[p + ".mlp.lin2.bias"])
        queries = _ln(queries + m, sd, p + ".norm3")
        q, k = queries + point_embedding, keys + key_pe
        keys = _ln(keys + _dec_attention(sd, p + ".cross_attn_image_to_token", k, q, queries, heads), sd, p + ".norm4")
    q, k = queries + point_embedding, keys + key_pe
    queries = queries + _dec_attention(sd, pre + ".final_attn_token_to_image", q, k, keys, heads)
    return _ln(queries, sd, pre + ".norm_final_attn"), keys


def _mlp(sd: SD, p: str, x, n: int):
    for i in range(n):
        x = F.linear(x, sd[f"{p}.layers.{i}.weight"], sd[f"{p}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def mask_decoder(sd: SD, image_embeddings, image_pe, sparse, dense, cfg: dict, multimask_output: bool = True):
    """MaskDecoder.forward / predict_masks (:71-149) -> (masks [B, 3|1, 4h, 4w], iou [B, 3|1])."""
    n_tok = 4
    out_tokens = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], dim=0)
    tokens = torch.cat((out_tokens.unsqueeze(0).expand(sparse.size(0), -1, -1), sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0) + dense
    pos = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(sd, src, pos, tokens, cfg)
    iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:1 + n_tok, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src, sd["mask_decoder.output_upscaling.0.weight"], sd["mask_decoder.output_upscaling.0.bias"], stride=2)
    up = F.gelu(_ln2d(up, sd, "mask_decoder.output_upscaling.1"))
    up = F.gelu(F.conv_transpose2d(up, sd["mask_decoder.output_upscaling.3.weight"], sd["mask_decoder.output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([_mlp(sd, f"mask_decoder.output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3) for i in range(n_tok)], dim=1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou = _mlp(sd, "mask_decoder.iou_prediction_head", iou_tok, cfg.get("iou_head_depth", 3))
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl], iou[:, sl]


# ----------------------------------------------------------------------------------------
# predictor (predictor.py, utils/transforms.py, modeling/sam.py)
# ----------------------------------------------------------------------------------------
def preprocess_shape(oldh: int, oldw: int, long_side: int) -> Tuple[int, int]:
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def apply_image(image: np.ndarray, long_side: int) -> np.ndarray:
    """ResizeLongestSide.apply_image (transforms.py:26-31): PIL bilinear resize of a uint8 HWC image."""
    from PIL import Image
    nh, nw = preprocess_shape(image.shape[0], image.shape[1], long_side)
    return np.array(Image.fromarray(image).resize((nw, nh), Image.BILINEAR))


def apply_coords(coords: np.ndarray, original_size, long_side: int) -> np.ndarray:
    oh, ow = original_size
    nh, nw = preprocess_shape(oh, ow, long_side)
    c = np.array(coords, dtype=float, copy=True)
    c[..., 0] = c[..., 0] * (nw / ow)
    c[..., 1] = c[..., 1] * (nh / oh)
    return c


def preprocess(x: torch.Tensor, img_size: int) -> torch.Tensor:
    """Sam.preprocess (sam.py:164-174) on [1,3,h,w] float pixel values 0..255."""
    mean = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    std = torch.tensor(PIXEL_STD).view(-1, 1, 1)
    x = (x - mean) / std
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))


def postprocess_masks(masks: torch.Tensor, img_size: int, input_size, original_size) -> torch.Tensor:
    """Sam.postprocess_masks (sam.py:134-162)."""
    masks = F.interpolate(masks, (img_size, img_size), mode="bilinear", align_corners=False)
    masks = masks[..., : input_size[0], : input_size[1]]
    return F.interpolate(masks, tuple(original_size), mode="bilinear", align_corners=False)


def set_image(sd: SD, cfg: dict, image: np.ndarray):
    """SamPredictor.set_image / set_torch_image (predictor.py:36-100): uint8 HWC RGB -> features."""
    inp = apply_image(image, cfg["img_size"])
    t = torch.as_tensor(inp).permute(2, 0, 1).contiguous()[None, :, :, :].float()
    feats = image_encoder(sd, preprocess(t, cfg["img_size"]), cfg)
    return dict(features=feats, input_size=tuple(t.shape[-2:]), original_size=tuple(image.shape[:2]))


def predict_torch(sd: SD, cfg: dict, state: dict, point_coords: torch.Tensor, point_labels: torch.Tensor, multimask_output=True,
                  boxes: torch.Tensor = None, mask_input: torch.Tensor = None):
    """SamPredictor.predict_torch (predictor.py:177-251) with return_logits=True; prompt tokens as
    PromptEncoder.forward builds them (:147-155): points (padded only when there is no box), then box corners."""
    g = cfg["img_size"] // cfg["patch"]
    size = (cfg["img_size"], cfg["img_size"])
    sparse = embed_points(sd, point_coords.float(), point_labels.float(), size, pad=boxes is None)
    if boxes is not None:
        sparse = torch.cat([sparse, embed_boxes(sd, boxes.float(), size)], dim=1)
    if mask_input is not None:
        dense = embed_masks(sd, mask_input.float().expand(sparse.shape[0], -1, -1, -1))
    else:
        dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(sparse.shape[0], -1, g, g)
    low, iou = mask_decoder(sd, state["features"], dense_pe(sd, (g, g)), sparse, dense, cfg, multimask_output)
    masks = postprocess_masks(low, cfg["img_size"], state["input_size"], state["original_size"])
    return masks, iou, low


# ----------------------------------------------------------------------------------------
# automatic mask generator (automatic_mask_generator.py, utils/amg.py)
# ----------------------------------------------------------------------------------------
def build_point_grid(n: int) -> np.ndarray:
    off = 1 / (2 * n)
    side = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(side[None, :], (n, 1)), np.tile(side[:, None], (1, n))], axis=-1).reshape(-1, 2)


def stability_score(masks: torch.Tensor, thr: float, off: float) -> torch.Tensor:
    inter = (masks > (thr + off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (thr - off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    """batched_mask_to_box (amg.py:310-353) for [n, H, W] bool -> [n, 4] XYXY, zeros when empty."""
    n, h, w = masks.shape
    out = torch.zeros(n, 4, dtype=torch.int64)
    for i in range(n):
        ys, xs = torch.nonzero(masks[i], as_tuple=True)
        if len(ys):
            out[i] = torch.tensor([xs.min(), ys.min(), xs.max(), ys.max()])
    return out


def box_nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """Greedy NMS as torchvision.ops.nms defines it: visit boxes by decreasing score, drop every later
    box whose IoU with a kept one exceeds ``thr`` (areas (x2-x1)(y2-y1)); returns kept indices in
    decreasing score order."""
    b = boxes.double().numpy()
    order = np.argsort(-scores.double().numpy(), kind="stable")
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep, dead = [], np.zeros(len(b), dtype=bool)
    for a_ in range(len(order)):
        i = order[a_]
        if dead[i]:
            continue
        keep.append(i)
        for j in order[a_ + 1:]:
            if dead[j]:
                continue
            iw = min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])
            ih = min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])
            inter = max(iw, 0.0) * max(ih, 0.0)
            if inter / (area[i] + area[j] - inter) > thr:
                dead[j] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def generate_crop_boxes(im_size, n_layers: int, overlap_ratio: float):
    """utils/amg.py:200-238."""
    import math
    from itertools import product
    crop_boxes, layer_idxs = [], []
    im_h, im_w = im_size
    short_side = min(im_h, im_w)
    crop_boxes.append([0, 0, im_w, im_h])
    layer_idxs.append(0)

    def crop_len(orig_len, n_crops, overlap):
        return int(math.ceil((overlap * (n_crops - 1) + orig_len) / n_crops))

    for i_layer in range(n_layers):
        n_crops_per_side = 2 ** (i_layer + 1)
        overlap = int(overlap_ratio * short_side * (2 / n_crops_per_side))
        crop_w = crop_len(im_w, n_crops_per_side, overlap)
        crop_h = crop_len(im_h, n_crops_per_side, overlap)
        crop_box_x0 = [int((crop_w - overlap) * i) for i in range(n_crops_per_side)]
        crop_box_y0 = [int((crop_h - overlap) * i) for i in range(n_crops_per_side)]
        for x0, y0 in product(crop_box_x0, crop_box_y0):
            crop_boxes.append([x0, y0, min(x0 + crop_w, im_w), min(y0 + crop_h, im_h)])
            layer_idxs.append(i_layer + 1)
    return crop_boxes, layer_idxs


def is_box_near_crop_edge(boxes: torch.Tensor, crop_box, orig_box, atol: float = 20.0) -> torch.Tensor:
    """utils/amg.py:78-88."""
    crop_box_torch = torch.as_tensor(crop_box, dtype=torch.float)
    orig_box_torch = torch.as_tensor(orig_box, dtype=torch.float)
    x0, y0 = crop_box[0], crop_box[1]
    boxes = (boxes + torch.tensor([[x0, y0, x0, y0]])).float()
    near_crop_edge = torch.isclose(boxes, crop_box_torch[None, :], atol=atol, rtol=0)
    near_image_edge = torch.isclose(boxes, orig_box_torch[None, :], atol=atol, rtol=0)
    return torch.any(torch.logical_and(near_crop_edge, ~near_image_edge), dim=1)


def mask_to_rle(tensor: torch.Tensor) -> List[dict]:
    """utils/amg.py:107-135 ([b, h, w] bool -> uncompressed column-major run lengths)."""
    b, h, w = tensor.shape
    tensor = tensor.permute(0, 2, 1).flatten(1)
    diff = tensor[:, 1:] ^ tensor[:, :-1]
    change_indices = diff.nonzero()
    out = []
    for i in range(b):
        cur = change_indices[change_indices[:, 0] == i, 1]
        cur = torch.cat([torch.tensor([0], dtype=cur.dtype), cur + 1, torch.tensor([h * w], dtype=cur.dtype)])
        btw = cur[1:] - cur[:-1]
        counts = [] if tensor[i, 0] == 0 else [0]
        counts.extend(btw.tolist())
        out.append({"size": [h, w], "counts": counts})
    return out


def rle_to_mask(rle: dict) -> np.ndarray:
    """utils/amg.py:138-149."""
    h, w = rle["size"]
    mask = np.empty(h * w, dtype=bool)
    idx, parity = 0, False
    for count in rle["counts"]:
        mask[idx: idx + count] = parity
        idx += count
        parity ^= True
    return mask.reshape(w, h).transpose()


def generate(sd: SD, cfg: dict, image: np.ndarray, *, points_per_side: int = 32, points_per_batch: int = 64,
             pred_iou_thresh: float = 0.88, stability_score_thresh: float = 0.95, stability_score_offset: float = 1.0,
             box_nms_thresh: float = 0.7, mask_threshold: float = 0.0, dense_pred: bool = True,
             select_point_coords=None, select_point_labels=None, sel_pred_iou_thresh: float = 0.88,
             sel_stability_score_thresh: float = 0.95, sel_stability_score_offset: float = 1.0,
             multimask_output: bool = True, sel_multimask_output: bool = True, output_layer: int = -1,
             sel_output_layer: int = -1, select_box=None, select_mask_input=None, crop_n_layers: int = 0,
             crop_nms_thresh: float = 0.7, crop_overlap_ratio: float = 512 / 1500, crop_n_points_downscale_factor: int = 1,
             output_mode: str = "binary_mask") -> List[dict]:
    """SamAutomaticMaskGenerator.generate (automatic_mask_generator.py:160-243): crops (:245-293), per crop the dense
    point grid (:385-453) and, on the whole image, the caller's prompt batches (select_point_coords: list of
    [n, k, 2] arrays, one per prompt size; :332-365, 455-556), the crop-edge filter and NMS (:369-383); binary-mask
    or uncompressed-RLE output."""
    oh, ow = image.shape[:2]
    crop_boxes, layer_idxs = generate_crop_boxes((oh, ow), crop_n_layers, crop_overlap_ratio)
    grids = [build_point_grid(int(points_per_side / (crop_n_points_downscale_factor ** i))) for i in range(crop_n_layers + 1)]

    def pick(layer, *ts):
        if sel_multimask_output and layer in (0, 1, 2):
            return [t[:, layer:layer + 1] for t in ts]
        if sel_multimask_output and layer in (3, 4, 5):
            return [t[:, layer - 3:] for t in ts]
        return list(ts)

    allc = dict(masks=[], iou=[], stab=[], boxes=[], low=[], crop=[])
    all_points = []
    for crop_box, layer_idx in zip(crop_boxes, layer_idxs):
        x0, y0, x1, y1 = crop_box
        crop = image[y0:y1, x0:x1, :]
        ch, cw = crop.shape[:2]
        state = set_image(sd, cfg, crop)
        recs = dict(masks=[], iou=[], stab=[], boxes=[], points=[], low=[])

        def batch(coords, labels, plist, multi, layer, iou_thr, stab_thr, stab_off, box=None, mask_input=None):
            tp = torch.as_tensor(apply_coords(coords, (ch, cw), cfg["img_size"]))
            bt = None
            if box is not None:                        # transforms.py:59-65 and automatic_mask_generator.py:483-489
                bt = torch.as_tensor(apply_coords(np.asarray(box, dtype=float).reshape(-1, 2, 2), (ch, cw), cfg["img_size"]).reshape(-1, 4))
                bt = bt.expand(tp.shape[0], -1)
            mi = None
            if mask_input is not None:
                mi = torch.as_tensor(np.asarray(mask_input), dtype=torch.float32)
                mi = mi[None] if mi.dim() == 3 else mi
            masks, iou, low = predict_torch(sd, cfg, state, tp, torch.as_tensor(labels), multi, boxes=bt, mask_input=mi)
            masks, iou, low = pick(layer, masks, iou, low)
            nm = masks.shape[1]
            masks, iou, low = masks.flatten(0, 1), iou.flatten(0, 1), low.flatten(0, 1)
            points = [plist[i // nm] for i in range(masks.shape[0])]
            keep = iou > iou_thr if iou_thr > 0.0 else torch.ones_like(iou, dtype=torch.bool)
            stab = stability_score(masks, mask_threshold, stab_off)
            if stab_thr > 0.0:
                keep = keep & (stab >= stab_thr)
            bin_ = masks > mask_threshold
            boxes = mask_to_box(bin_).float()
            keep = keep & ~is_box_near_crop_edge(boxes, crop_box, [0, 0, ow, oh])
            idx = torch.nonzero(keep).flatten().tolist()
            recs["masks"].append(bin_[idx]); recs["iou"].append(iou[idx]); recs["stab"].append(stab[idx]); recs["low"].append(low[idx])
            recs["boxes"].append(boxes[idx]); recs["points"].extend(points[i] for i in idx)

        if dense_pred:
            pts_all = grids[layer_idx] * np.array([[cw, ch]], dtype=np.float64)
            for b0 in range(0, len(pts_all), points_per_batch):
                pts = pts_all[b0:b0 + points_per_batch]
                batch(pts[:, None, :], np.ones((len(pts), 1), dtype=np.int64), [[p] for p in pts.tolist()], multimask_output,
                      output_layer, pred_iou_thresh, stability_score_thresh, stability_score_offset)
        if layer_idx == 0 and select_point_coords is not None:
            for coords, labels in zip(select_point_coords, select_point_labels):
                coords, labels = np.asarray(coords), np.asarray(labels)
                for b0 in range(0, len(coords), points_per_batch):
                    c = coords[b0:b0 + points_per_batch]
                    batch(c, labels[b0:b0 + points_per_batch], c.tolist(), sel_multimask_output, sel_output_layer,
                          sel_pred_iou_thresh, sel_stability_score_thresh, sel_stability_score_offset,
                          box=select_box[0] if select_box is not None else None, mask_input=select_mask_input)
        if not recs["iou"]:
            continue
        points = recs.pop("points")
        cat = {k: torch.cat(v, dim=0) for k, v in recs.items()}
        if len(cat["iou"]) == 0:
            continue
        keep = box_nms(cat["boxes"], cat["iou"], box_nms_thresh).tolist()
        masks = torch.nn.functional.pad(cat["masks"][keep], (x0, ow - x1, y0, oh - y1), value=0)       # uncrop_masks, amg.py:262-272
        allc["masks"].append(masks); allc["iou"].append(cat["iou"][keep]); allc["stab"].append(cat["stab"][keep])
        allc["low"].append(cat["low"][keep])
        allc["boxes"].append(cat["boxes"][keep] + torch.tensor([[x0, y0, x0, y0]], dtype=torch.float))
        allc["crop"].append(torch.tensor([crop_box] * len(keep), dtype=torch.float).reshape(-1, 4))
        all_points.extend([[px + x0, py + y0] for px, py in points[i]] for i in keep)
    if not allc["iou"]:
        return []
    cat = {k: torch.cat(v, dim=0) for k, v in allc.items()}
    order = list(range(len(cat["iou"])))
    if len(crop_boxes) > 1:
        area = (cat["crop"][:, 2] - cat["crop"][:, 0]) * (cat["crop"][:, 3] - cat["crop"][:, 1])
        order = box_nms(cat["boxes"], 1 / area, crop_nms_thresh).tolist()
    out = []
    for i in order:
        m = cat["masks"][i].numpy()
        x0, y0, x1, y1 = cat["boxes"][i].tolist()
        c0, c1, c2, c3 = cat["crop"][i].tolist()
        seg = m if output_mode == "binary_mask" else mask_to_rle(cat["masks"][i][None])[0]
        out.append(dict(segmentation=seg, area=int(m.sum()), bbox=[x0, y0, x1 - x0, y1 - y0], predicted_iou=float(cat["iou"][i]),
                        point_coords=all_points[i], low_res_masks=cat["low"][i].numpy(),
                        stability_score=float(cat["stab"][i]), crop_box=[c0, c1, c2 - c0, c3 - c1]))
    return out
