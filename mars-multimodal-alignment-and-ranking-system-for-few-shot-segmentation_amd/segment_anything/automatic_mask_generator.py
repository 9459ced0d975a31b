"""Automatic mask generator on the HIP kernels.

Interface of /root/reference/segment_anything/automatic_mask_generator.py:38-243 (dense point grid,
one crop layer): ``generate(image) -> list of records`` with the reference's keys, plus
``generate_proposals`` which hands the ranking path a fixed-size float {0,1} tensor.

Per batch of points the decoder output is resampled to the image size once; one kernel then gives
the two stability counts and the box of every mask, so the filters (predicted IoU, stability) run
on a few integers per mask on the host, and only the surviving masks are kept (as {0,1} floats on
the GPU).  Box NMS runs on the host.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .. import ops
from .amg import (area_from_rle, batched_mask_to_box, build_all_layer_point_grids, build_point_grid, coco_encode_rle,  # noqa: F401
                  generate_crop_boxes, is_box_near_crop_edge, mask_to_rle, remove_small_regions)
from .modeling import Sam
from .predictor import SamPredictor


class SamAutomaticMaskGenerator:
    def __init__(self, model: Sam, points_per_side: Optional[int] = 32, points_per_batch: int = 64, pred_iou_thresh: float = 0.88,
                 sel_pred_iou_thresh: float = 0.88, stability_score_thresh: float = 0.95, stability_score_offset: float = 1.0,
                 sel_stability_score_thresh: float = 0.95, sel_stability_score_offset: float = 1.0, box_nms_thresh: float = 0.7,
                 crop_n_layers: int = 0, crop_nms_thresh: float = 0.7, crop_overlap_ratio: float = 512 / 1500,
                 crop_n_points_downscale_factor: int = 1, point_grids: Optional[List[np.ndarray]] = None, min_mask_region_area: int = 0,
                 output_mode: str = "binary_mask", multimask_output: bool = True, sel_multimask_output: bool = True,
                 output_layer: int = -1, sel_output_layer: int = -1, dense_pred: bool = True, **unused):
        assert (points_per_side is None) != (point_grids is None), "Exactly one of points_per_side or point_grid must be provided."
        assert output_mode in ("binary_mask", "uncompressed_rle", "coco_rle"), f"Unknown output_mode {output_mode}."
        self.min_mask_region_area = min_mask_region_area
        self.point_grids = (build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
                            if points_per_side is not None else point_grids)
        self.crop_n_layers, self.crop_nms_thresh, self.crop_overlap_ratio = crop_n_layers, crop_nms_thresh, crop_overlap_ratio
        self.crop_n_points_downscale_factor, self.output_mode = crop_n_points_downscale_factor, output_mode
        self.predictor = SamPredictor(model)
        self.points_per_batch = points_per_batch
        # Grid points per decoder launch chain.  The reference decodes the grid 64 points at a time (points_per_batch)
        # to bound memory; every point's masks depend on that point alone, so the dense grid goes through the decoder
        # in chunks of this many points (a multiple of points_per_batch: 1024 = the whole 32 x 32 grid), which costs one
        # chain of launches and one host round trip per chunk instead of one per 64 points.  The caller's own prompts
        # (select_point_coords) keep points_per_batch.
        self.dense_points_per_launch = max(points_per_batch, int(os.environ.get("MARS_SAM_DENSE_CHUNK", "1024")) // points_per_batch * points_per_batch)
        self.pred_iou_thresh = pred_iou_thresh
        self.stability_score_thresh = stability_score_thresh
        self.stability_score_offset = stability_score_offset
        self.sel_pred_iou_thresh = sel_pred_iou_thresh
        self.sel_stability_score_thresh = sel_stability_score_thresh
        self.sel_stability_score_offset = sel_stability_score_offset
        self.box_nms_thresh = box_nms_thresh
        self.multimask_output = multimask_output
        self.sel_multimask_output = sel_multimask_output
        self.output_layer = output_layer
        self.sel_output_layer = sel_output_layer
        self.dense_pred = dense_pred
        self.stored_features: Dict[Any, Any] = {}

    def reset_stored_features(self):
        self.stored_features = {}

    # ------------------------------------------------------------------ lines 385-453 and 455-556
    def _pick_layers(self, layer: int, masks, iou, low):
        """Output selection shared by both prompt paths (lines 405-415, 508-518; both test sel_multimask_output)."""
        cut = lambda t, sl: None if t is None else t[:, sl]
        if self.sel_multimask_output:
            if layer in (0, 1, 2):
                return cut(masks, slice(layer, layer + 1)), iou[:, layer:layer + 1], low[:, layer:layer + 1]
            if layer in (3, 4, 5):
                return cut(masks, slice(layer - 3, None)), iou[:, layer - 3:], low[:, layer - 3:]
        return masks, iou, low

    def _fused_post(self) -> bool:
        """Whether the image-size logits are evaluated on the fly inside the statistics and threshold kernels (no
        [B, C, H, W] tensor); MARS_SAM_FUSED_POST=0 keeps postprocess_masks + mask_stats + threshold_masks."""
        pr = self.predictor
        L = 4 * pr.model.prompt_encoder.image_embedding_size[0]
        return os.environ.get("MARS_SAM_FUSED_POST", "1") != "0" and pr.model.postprocess_plan(L, pr.input_size, pr.original_size) is not None

    def _keep_batch(self, masks, iou, low, points: list, iou_thresh: float, stab_thresh: float, stab_off: float, keep_lists: dict,
                    defer: bool = False):
        """masks [B, C, H, W] logits (or None: they are evaluated from ``low`` [B, C, L, L] on the fly): one statistics
        launch, the two filters on the host, survivors kept as {0,1} floats on the GPU.  ``points`` holds the prompt
        points of every batch row.  ``defer`` (only with masks None, and only when no other prediction follows before
        the crop's NMS): nothing is made of the survivors yet - their plane numbers go to keep_lists["planes"] and
        ``low`` with its plan to keep_lists["deferred"]; _process_crop makes the masks of what the NMS keeps."""
        pr, dev = self.predictor, self.predictor.device
        nm, n = low.shape[1], low.shape[0] * low.shape[1]
        iou = iou.flatten(0, 1)
        stats = torch.empty(n, 6, dtype=torch.int32, device=dev)
        plan, all_bits = None, None
        if masks is None:                                             # logits evaluated from ``low`` inside the kernels
            plan = pr.model.postprocess_plan(low.shape[-1], pr.input_size, pr.original_size)
            if iou_thresh <= 0.0 and stab_thresh <= 0.0 and not defer:    # no filter: every mask is kept, its 0 / 1 form comes out of the same pass
                all_bits = torch.empty(n, int(pr.original_size[0]), int(pr.original_size[1]), dtype=torch.float32, device=dev)
            ops.resample_mask_stats(low, *plan, pr.model.mask_threshold, stab_off, stats, bits=all_bits)
        else:
            masks = masks.flatten(0, 1).contiguous()
            ops.mask_stats(masks, pr.model.mask_threshold, stab_off, stats)
        st, iou_h = stats.cpu(), iou.cpu()
        stab = st[:, 0].float() / st[:, 1].float()                    # intersections / unions (amg.py:173)
        keep = torch.ones(n, dtype=torch.bool)
        if iou_thresh > 0.0:
            keep &= iou_h > iou_thresh
        if stab_thresh > 0.0:
            keep &= stab >= stab_thresh
        idx = torch.nonzero(keep).flatten()
        if idx.numel() == 0:
            return
        keep_lists["iou"].append(iou_h[idx])
        keep_lists["stab"].append(stab[idx])
        keep_lists["boxes"].append(st[idx, 2:6].float())
        keep_lists["points"].append((points, (idx // nm).numpy()))                   # rows repeat once per output mask; resolved for what the NMS keeps
        if defer and masks is None:
            keep_lists["planes"] = idx
            keep_lists["deferred"] = (low, plan)
            return
        sel = idx.to(dev)
        if all_bits is not None:
            binm = all_bits
            low_kept = low.clone(memory_format=torch.contiguous_format).view(n, low.shape[-2], low.shape[-1])      # (low is a view of a reused buffer)
        elif masks is None:
            oh, ow = pr.original_size
            binm = torch.empty(idx.numel(), int(oh), int(ow), dtype=torch.float32, device=dev)
            ops.resample_threshold(low, idx.to(torch.int32).to(dev), *plan, pr.model.mask_threshold, binm)
            low_kept = low[sel // nm, sel % nm]
        else:
            kept = masks[sel].contiguous()
            binm = torch.empty_like(kept)
            ops.threshold_masks(kept, pr.model.mask_threshold, binm)
            low_kept = low.flatten(0, 1)[sel]
        keep_lists["masks"].append(binm)
        keep_lists["low"].append(low_kept)

    def _process_batch(self, points: np.ndarray, im_size, keep_lists: dict, defer: bool = False):
        pr, dev = self.predictor, self.predictor.device
        tp = torch.as_tensor(pr.transform.apply_coords(points, im_size), dtype=torch.float32, device=dev)
        labels = torch.ones(tp.shape[0], 1, dtype=torch.int32, device=dev)
        masks, iou, low = pr.predict_torch(tp[:, None, :], labels, multimask_output=self.multimask_output, return_logits=True,
                                           upscale=not self._fused_post())
        masks, iou, low = self._pick_layers(self.output_layer, masks, iou, low)
        self._keep_batch(masks, iou, low, [[p] for p in points.tolist()], self.pred_iou_thresh, self.stability_score_thresh,
                         self.stability_score_offset, keep_lists, defer=defer and masks is None)

    def _process_sel_batch(self, im_size, point_coords: np.ndarray, point_labels: np.ndarray, keep_lists: dict, box=None,
                           mask_input=None):
        """One batch of prompts of k points each: point_coords [B, k, 2] in image pixels, point_labels [B, k]; ``box``
        (x0, y0, x1, y1 in image pixels) is added to every prompt of the batch."""
        pr, dev = self.predictor, self.predictor.device
        point_coords = np.asarray(point_coords)
        if point_coords.ndim == 2:
            point_coords, point_labels = point_coords[None], np.asarray(point_labels)[None]
        tp = torch.as_tensor(pr.transform.apply_coords(point_coords, im_size), dtype=torch.float32, device=dev)
        labels = torch.as_tensor(np.asarray(point_labels), dtype=torch.int32, device=dev)
        boxes = None
        if box is not None:
            b = torch.as_tensor(pr.transform.apply_boxes(np.asarray(box, dtype=np.float64), im_size), dtype=torch.float32, device=dev)
            boxes = b.expand(tp.shape[0], -1) if b.shape[0] == 1 else b
        if mask_input is not None:                                # one low-resolution mask for every prompt (lines 490-495)
            mask_input = torch.as_tensor(np.asarray(mask_input), dtype=torch.float32, device=dev)
            mask_input = mask_input[None] if mask_input.dim() == 3 else mask_input
        masks, iou, low = pr.predict_torch(tp, labels, boxes=boxes, mask_input=mask_input, multimask_output=self.sel_multimask_output,
                                           return_logits=True, upscale=not self._fused_post())
        masks, iou, low = self._pick_layers(self.sel_output_layer, masks, iou, low)
        self._keep_batch(masks, iou, low, point_coords.tolist(), self.sel_pred_iou_thresh, self.sel_stability_score_thresh,
                         self.sel_stability_score_offset, keep_lists)

    # ------------------------------------------------------------------ lines 160-243, 245-293, 295-383
    def _process_crop(self, image: np.ndarray, crop_box, layer: int, select_point_coords, select_point_labels, select_box,
                      select_mask_input) -> Optional[dict]:
        """One crop: embedding, dense grid and (on the whole image only) the caller's prompts, edge filter, NMS; boxes,
        points and masks come back in the image frame."""
        oh, ow = image.shape[:2]
        x0, y0, x1, y1 = crop_box
        crop = image[y0:y1, x0:x1, :]
        ch, cw = crop.shape[:2]
        self.predictor.set_image(crop)
        lists = dict(masks=[], iou=[], stab=[], boxes=[], points=[], low=[])
        if self.dense_pred:
            pts_all = self.point_grids[layer] * np.array([[cw, ch]], dtype=np.float64)
            # One decoder batch and nothing after it: the NMS below needs boxes and scores only, so the 0 / 1 masks (and the
            # copies of the low-resolution logits) are made afterwards, of what it keeps, not of every survivor of the filters.
            single = len(pts_all) <= self.dense_points_per_launch and not (layer == 0 and select_point_coords is not None)
            for b0 in range(0, len(pts_all), self.dense_points_per_launch):
                self._process_batch(pts_all[b0:b0 + self.dense_points_per_launch], (ch, cw), lists, defer=single)
        if layer == 0 and select_point_coords is not None:
            assert select_point_labels is not None, "select_point_labels should not be None !"
            for coords, labels in zip(select_point_coords, select_point_labels):     # one array per prompt size
                for b0 in range(0, len(coords), self.points_per_batch):
                    self._process_sel_batch((ch, cw), coords[b0:b0 + self.points_per_batch],
                                            labels[b0:b0 + self.points_per_batch], lists, box=select_box,
                                            mask_input=select_mask_input)
        self.predictor.reset_image()
        if not lists["iou"]:
            return None
        points = lists.pop("points")
        deferred, planes = lists.pop("deferred", None), lists.pop("planes", None)
        cat = {k: (v[0] if len(v) == 1 else torch.cat(v, dim=0)) for k, v in lists.items() if v}      # (cat copies even a single tensor)
        ok = ~is_box_near_crop_edge(cat["boxes"], crop_box, [0, 0, ow, oh])           # lines 443-446
        order = ops.box_nms_host(cat["boxes"][ok], cat["iou"][ok], self.box_nms_thresh)
        keep = torch.nonzero(ok).flatten()[order]
        if keep.numel() == 0:
            return None
        out = {k: v[keep.to(v.device)] for k, v in cat.items()}
        if deferred is not None:
            low, plan = deferred
            pk, dev = planes[keep], low.device
            out["masks"] = torch.empty(pk.numel(), ch, cw, dtype=torch.float32, device=dev)
            ops.resample_threshold(low, pk.to(torch.int32).to(dev), *plan, self.predictor.model.mask_threshold, out["masks"])
            pkd = pk.to(dev)
            out["low"] = low[pkd // low.shape[1], pkd % low.shape[1]]
        # the prompt points of the kept masks: entry i of the concatenated survivors lives in chunk c at i - starts[c]
        starts = np.cumsum([0] + [len(rows) for _, rows in points])
        chunk = np.searchsorted(starts, keep.numpy(), side="right") - 1
        out["points"] = [[[x + x0, y + y0] for x, y in points[c][0][int(points[c][1][i - starts[c]])]] for i, c in zip(keep.tolist(), chunk.tolist())]
        out["boxes"] = out["boxes"] + torch.tensor([[x0, y0, x0, y0]], dtype=out["boxes"].dtype)
        if (ch, cw) != (oh, ow):                                                      # masks of a crop, zero outside it
            full = torch.zeros(out["masks"].shape[0], oh, ow, dtype=out["masks"].dtype, device=out["masks"].device)
            full[:, y0:y1, x0:x1] = out["masks"]
            out["masks"] = full
        out["crop_boxes"] = torch.tensor([list(crop_box)] * len(keep), dtype=torch.float32).reshape(-1, 4)
        return out

    def _run(self, image: np.ndarray, select_point_coords=None, select_point_labels=None, select_box=None,
             select_mask_input=None) -> dict:
        if select_box is not None:
            if not (isinstance(select_box, list) and len(select_box) == 1):
                raise NotImplementedError("select_box is a list holding one box (automatic_mask_generator.py:336-341)")
            select_box = select_box[0]
        oh, ow = image.shape[:2]
        boxes, layers, _ = generate_crop_boxes((oh, ow), self.crop_n_layers, self.crop_overlap_ratio)
        parts = [p for p in (self._process_crop(image, cb, ly, select_point_coords, select_point_labels, select_box, select_mask_input)
                             for cb, ly in zip(boxes, layers)) if p is not None]
        if not parts:
            return dict(masks=torch.zeros(0, oh, ow, device=self.predictor.device), iou=torch.zeros(0), stab=torch.zeros(0),
                        boxes=torch.zeros(0, 4), points=[], low=None, crop_boxes=torch.zeros(0, 4), size=(oh, ow))
        out = {k: (parts[0][k] if len(parts) == 1 else torch.cat([p[k] for p in parts], dim=0)) for k in parts[0] if k != "points"}
        out["points"] = [pt for p in parts for pt in p["points"]]
        if len(boxes) > 1:                                                            # duplicates between crops: the smaller crop wins
            cb = out["crop_boxes"]
            scores = 1 / ((cb[:, 2] - cb[:, 0]) * (cb[:, 3] - cb[:, 1]))
            keep = ops.box_nms_host(out["boxes"], scores, self.crop_nms_thresh)
            pts = out.pop("points")
            out = {k: v[keep.to(v.device)] for k, v in out.items()}
            out["points"] = [pts[i] for i in keep.tolist()]
        out["size"] = (oh, ow)
        return out

    def postprocess_small_regions(self, r: dict, masks_h: np.ndarray, min_area: int, nms_thresh: float):
        """Lines 559-607: holes, then islands, below ``min_area`` are removed from every mask; boxes are recomputed and
        a box NMS that prefers untouched masks drops the duplicates this creates.  Only masks that changed take their
        new pixels and box.  Returns the filtered result and host masks."""
        n = masks_h.shape[0]
        if n == 0:
            return r, masks_h
        new_masks, scores = [], []
        for m in masks_h:
            m1, ch1 = remove_small_regions(m, min_area, "holes")
            m2, ch2 = remove_small_regions(m1, min_area, "islands")
            new_masks.append(m2)
            scores.append(float(not (ch1 or ch2)))
        new_masks = np.stack(new_masks)
        boxes = torch.from_numpy(batched_mask_to_box(new_masks))
        keep = ops.box_nms_host(boxes, torch.tensor(scores), nms_thresh)
        out_masks, out_boxes = masks_h.copy(), r["boxes"].clone().float()
        for i in keep.tolist():
            if scores[i] == 0.0:
                out_masks[i], out_boxes[i] = new_masks[i], boxes[i]
        pts = r["points"]
        out = {k: (v[keep.to(v.device)] if torch.is_tensor(v) else v) for k, v in r.items() if k not in ("points", "boxes", "masks")}
        out["boxes"], out["points"] = out_boxes[keep], [pts[i] for i in keep.tolist()]
        out["masks"] = None                                   # the host masks below are the current ones
        return out, out_masks[keep.numpy()]

    def generate(self, image: np.ndarray, select_point_coords=None, select_point_labels=None, select_box=None,
                 select_mask_input=None) -> List[Dict[str, Any]]:
        r = self._run(image, select_point_coords, select_point_labels, select_box, select_mask_input)
        oh, ow = r["size"]
        recs = []
        masks_h = r["masks"].cpu().numpy() > 0 if len(r["iou"]) else np.zeros((0, oh, ow), bool)
        if self.min_mask_region_area > 0:
            r, masks_h = self.postprocess_small_regions(r, masks_h, self.min_mask_region_area,
                                                        max(self.box_nms_thresh, self.crop_nms_thresh))
        for i in range(len(r["iou"])):
            x0, y0, x1, y1 = [float(v) for v in r["boxes"][i].tolist()]
            c0, c1, c2, c3 = [float(v) for v in r["crop_boxes"][i].tolist()]
            seg = masks_h[i] if self.output_mode == "binary_mask" else mask_to_rle(masks_h[i])
            if self.output_mode == "coco_rle":
                seg = coco_encode_rle(seg)
            recs.append({"segmentation": seg, "area": int(masks_h[i].sum()), "bbox": [x0, y0, x1 - x0, y1 - y0],
                         "predicted_iou": float(r["iou"][i]), "point_coords": r["points"][i],
                         "low_res_masks": r["low"][i].cpu().numpy(), "stability_score": float(r["stab"][i]),
                         "crop_box": [c0, c1, c2 - c0, c3 - c1]})
        return recs

    def generate_proposals(self, query_image: torch.Tensor, n: int) -> torch.Tensor:
        """Proposals for the ranking path: [n, H, W] float {0,1} on the GPU from a [1, 3, H, W] float image
        in [0, 1] - the top n by predicted IoU, padded by repetition (an empty result gives all-ones masks)."""
        img = query_image[0].mul(255).byte().permute(1, 2, 0).cpu().numpy()
        r = self._run(img)
        m = r["masks"]
        if m.shape[0] == 0:
            return torch.ones(n, img.shape[0], img.shape[1], dtype=torch.float32, device=self.predictor.device)
        order = torch.argsort(r["iou"], descending=True, stable=True)[:n]
        m = m[order.to(m.device)]
        reps = -(-n // m.shape[0])
        return m.repeat(reps, 1, 1)[:n].contiguous()


class ProposalGenerator:
    """``mask_generator`` object of MARS (/root/reference/mars/MARS.py:46-51): ``generate(support_images,
    support_masks, query_image) -> Tensor[N, H, W]``."""

    def __init__(self, amg: SamAutomaticMaskGenerator, n_proposals: int = 64):
        self.amg, self.n = amg, n_proposals

    def generate(self, support_images, support_masks, query_image) -> torch.Tensor:
        return self.amg.generate_proposals(query_image, self.n)
