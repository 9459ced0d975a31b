"""Host helpers of the automatic mask generator (/root/reference/segment_anything/utils/amg.py): crop boxes and
their point grids, the crop-edge filter, moving results of a crop back into the image frame, run-length codes."""
from __future__ import annotations

import math
from itertools import product
from typing import Any, Dict, List, Tuple

import numpy as np
import torch


def build_point_grid(n_per_side: int) -> np.ndarray:
    """amg.py:179-187: centres of an n x n grid over the unit square, as (x, y) rows."""
    offset = 1 / (2 * n_per_side)
    side = np.linspace(offset, 1 - offset, n_per_side)
    return np.stack([np.tile(side[None, :], (n_per_side, 1)), np.tile(side[:, None], (1, n_per_side))], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    """amg.py:189-198: layer i samples int(n / scale**i) points per side."""
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size: Tuple[int, int], n_layers: int, overlap_ratio: float):
    """amg.py:200-238: the whole image, then for layer i a (2**i x 2**i) tiling whose tiles overlap by
    overlap_ratio * short side * 2 / tiles-per-side pixels.  Returns (XYXY boxes, layer index, box id)."""
    im_h, im_w = im_size
    boxes, layers, ids = [[0, 0, im_w, im_h]], [0], [0]
    for layer in range(1, n_layers + 1):
        n = 2 ** layer
        overlap = int(overlap_ratio * min(im_h, im_w) * (2 / n))
        cw = int(math.ceil((overlap * (n - 1) + im_w) / n))
        ch = int(math.ceil((overlap * (n - 1) + im_h) / n))
        xs = [int((cw - overlap) * i) for i in range(n)]
        ys = [int((ch - overlap) * i) for i in range(n)]
        for x0, y0 in product(xs, ys):
            boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layers.append(layer)
            ids.append(len(ids))
    return boxes, layers, ids


def is_box_near_crop_edge(boxes: torch.Tensor, crop_box: List[int], orig_box: List[int], atol: float = 20.0) -> torch.Tensor:
    """amg.py:78-88: a side of a (crop-frame) box within ``atol`` of the crop's side but not of the image's."""
    crop = torch.as_tensor(crop_box, dtype=torch.float)
    orig = torch.as_tensor(orig_box, dtype=torch.float)
    b = boxes.float() + torch.tensor([[crop_box[0], crop_box[1], crop_box[0], crop_box[1]]], dtype=torch.float)
    near_crop = (b - crop[None]).abs() <= atol
    near_image = (b - orig[None]).abs() <= atol
    return (near_crop & ~near_image).any(dim=1)


def mask_to_rle(mask: np.ndarray) -> Dict[str, Any]:
    """amg.py:107-135 for one [H, W] bool mask: column-major run lengths, starting with the count of zeros."""
    h, w = mask.shape
    flat = np.asarray(mask, dtype=bool).T.reshape(-1)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [h * w]])
    counts = ([] if not flat[0] else [0]) + (bounds[1:] - bounds[:-1]).tolist()
    return {"size": [h, w], "counts": counts}


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:
    """amg.py:138-149."""
    h, w = rle["size"]
    flat = np.empty(h * w, dtype=bool)
    pos, value = 0, False
    for c in rle["counts"]:
        flat[pos:pos + c] = value
        pos += c
        value = not value
    return flat.reshape(w, h).transpose()


def area_from_rle(rle: Dict[str, Any]) -> int:
    return sum(rle["counts"][1::2])


def remove_small_regions(mask: np.ndarray, area_thresh: float, mode: str) -> Tuple[np.ndarray, bool]:
    """amg.py:274-298: fill holes / drop islands smaller than ``area_thresh`` (8-connected); the host routine of the
    library stands in for cv2.connectedComponentsWithStats."""
    from .. import ops
    return ops.remove_small_regions_host(mask, area_thresh, mode)


def batched_mask_to_box(masks: np.ndarray) -> np.ndarray:
    """amg.py:310-353 for host masks [N, H, W]: inclusive XYXY boxes, [0, 0, 0, 0] for an empty mask."""
    masks = np.asarray(masks, dtype=bool)
    out = np.zeros((masks.shape[0], 4), dtype=np.float32)
    for i, m in enumerate(masks):
        ys, xs = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
        if ys.size:
            out[i] = (xs[0], ys[0], xs[-1], ys[-1])
    return out


def coco_encode_rle(uncompressed_rle: Dict[str, Any]) -> Dict[str, Any]:
    """amg.py:301-308: the run lengths as the COCO API's compressed string (there: pycocotools ``frPyObjects``).
    Each count from the fourth on is stored as its difference to the count two places back; a value is cut into
    5-bit groups, low group first, bit 5 says that more groups follow, bit 4 of the last group is the sign, and
    every group is written as the character with code 48 + group."""
    counts = [int(c) for c in uncompressed_rle["counts"]]
    chars = []
    for i, x in enumerate(counts):
        if i > 2:
            x -= counts[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5                                          # arithmetic shift: -1 stays -1
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            chars.append(chr(c + 48))
    h, w = uncompressed_rle["size"]
    return {"size": [h, w], "counts": "".join(chars)}


def coco_decode_rle(rle: Dict[str, Any]) -> Dict[str, Any]:
    """Inverse of ``coco_encode_rle`` (the COCO API's ``rleFrString``): back to uncompressed run lengths."""
    s, counts, p = rle["counts"], [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return {"size": list(rle["size"]), "counts": counts}
