"""The reference Matcher (the producer of the stored proposals) on this package's kernels.

Interface of /root/reference/matcher/Matcher.py:
``Matcher.set_reference`` (128-185), ``set_target`` (187-205), ``predict`` (216-249),
``extract_img_feats`` (251-302), negative priors (304-417), ``patch_level_matching`` (419-577),
``clustering`` (579-591), ``mask_generation`` (619-834), ``RobustPromptSampler.get_mask_scores``
(1152-1210) and ``sample_points`` (1226-1295), ``clear``.

Device work: the DINOv2 pass, S = ref . tar^T, the SAM prompt batches, mask pooling and point counts.
Host work: the two assignment problems (C++ shortest augmenting paths), the transport problems
(C++ network simplex, threads), and the list handling of points and prompt combinations, which
follows the reference's order of operations and its draws from ``random`` / ``torch`` generators so
that a seeded run picks the same prompts.  The per-mask scoring loop of the reference is one pooling
launch, one point-count launch and N transport problems.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional

import numpy as np
import torch

from .. import ops
from ..mars.components.FilteringMergingModule import default_emd_threads
from ..mars.components.VisualVisualAlignmentModule import IMAGENET_MEAN, IMAGENET_STD, _pad64


class RobustPromptSampler:
    """Scores of proposed masks (lines 1140-1210); ``get_mask_scores`` takes all masks at once."""

    def __init__(self, encoder_feat_size: int, sample_range=(4, 6), max_iterations: int = 30, emd_threads: int = 0):
        self.encoder_feat_size = int(encoder_feat_size)
        self.sample_range = sample_range
        self.max_iterations = max_iterations
        self.emd_threads = emd_threads or int(os.environ.get("MARS_EMD_THREADS", "0")) or default_emd_threads()

    def get_mask_scores(self, points, masks: torch.Tensor, all_points, emd_cost: torch.Tensor, ref_masks_pool: torch.Tensor):
        """masks [n, 1, H, W] or [n, H, W] (bool / 0-1, on the GPU), all_points [K, 2] (x, y),
        emd_cost C = (1 - S) / 2 as a CPU f32 tensor [ns*P, P], ref_masks_pool flat {0,1} [ns*P].
        Returns (purity [n], coverage [n], emd_score [n]) as CPU float32 tensors."""
        G = self.encoder_feat_size
        P = G * G
        m = masks.reshape(masks.shape[0], masks.shape[-2], masks.shape[-1]).contiguous().float()
        dev = m.device
        n = m.shape[0]
        # lines 1177-1185: area-mean to G x G, > 0  ==  "any pixel of the patch is set" for 0/1 masks;
        # a mask without any set pixel selects every patch (thres = max - 1e-6 < 0)
        bits = torch.empty(n, P, dtype=torch.uint8, device=dev)
        ops.pool_masks(m, G, bits)
        pts = torch.as_tensor(np.asarray(all_points), dtype=torch.float32, device=dev).contiguous()
        inside = torch.empty(n, dtype=torch.int32, device=dev)
        ops.points_in_masks(m, pts, inside)
        bits_h = bits.cpu()
        empty = bits_h.sum(dim=1) == 0
        bits_h[empty] = 1
        fg = (ref_masks_pool.flatten() != 0).to(torch.uint8).cpu().contiguous()
        # lines 1188-1194: exact EMD between the reference foreground patches and the mask's patches
        cost = emd_cost.detach().cpu().float().contiguous()
        emd_score = ops.emd_scores_host(cost, fg, bits_h.contiguous(), self.emd_threads, is_cost=True).float()
        # lines 1200-1209
        inside_h = inside.cpu().double()
        area = bits_h.sum(dim=1).double().clamp(min=1.0)
        purity = (inside_h / area).float() + 1e-6
        coverage = (inside_h / float(pts.shape[0])).float() + 1e-6
        return purity, coverage, emd_score


    # ------------------------------------------------------------------ lines 1212-1295
    @staticmethod
    def combinations(n: int, k: int) -> List[List[int]]:
        """k-subsets of range(n), ordered by their largest element first (the order the prompts are batched in)."""
        if k > n:
            return []
        if k == 0:
            return [[]]
        if k == n:
            return [list(range(n))]
        out = []
        for last in range(n):
            out.extend(head + [last] for head in RobustPromptSampler.combinations(last, k - 1))
        return out

    def sample_points(self, points, negative_points=None):
        """Prompt sets of sample_range[0] .. sample_range[1] points: every combination when there are at most 8
        candidates, else ``max_iterations`` random draws per size.  With negative candidates each set gets as
        many negatives appended (labels 0).  Returns (list of [n, k(, 2k), 2] arrays, list of label arrays)."""
        samples, labels = [], []
        lo, hi = min(self.sample_range[0], len(points)), min(self.sample_range[1], len(points))
        for k in range(lo, hi + 1):
            if len(points) > 8:
                idx = [random.sample(range(len(points)), k) for _ in range(self.max_iterations)]
                pos = np.take(points, idx, axis=0)
                neg = None
                if negative_points is not None:
                    if len(negative_points) > 8:
                        nidx = [random.sample(range(len(negative_points)), k) for _ in range(self.max_iterations)]
                    else:
                        nidx = [random.choices(range(len(negative_points)), k=k) for _ in range(self.max_iterations)]
                    neg = np.take(negative_points, nidx, axis=0)
            else:
                idx = self.combinations(len(points), k)
                pos = np.take(points, idx, axis=0)
                neg = None
                if negative_points is not None:
                    nidx = [random.choices(range(len(negative_points)), k=k) for _ in range(len(idx))]
                    neg = np.take(negative_points, nidx, axis=0)
            if neg is None:
                samples.append(pos)
                labels.append(np.ones((pos.shape[0], k)))
            else:
                assert pos.shape[0] == neg.shape[0]
                samples.append(np.hstack((pos, neg)))
                labels.append(np.hstack((np.ones((pos.shape[0], k)), np.zeros((neg.shape[0], k)))))
        return samples, labels


def kmeans_pp(X, K: int, max_iters: int = 100):
    """k-means++ seeding and Lloyd iterations on host points (/root/reference/matcher/k_means.py:17-57); draws
    from torch's global generator in the reference's order."""
    X = torch.as_tensor(np.asarray(X)).float() if not torch.is_tensor(X) else X.float()
    centers = X[torch.randint(X.size(0), (1,))]
    for _ in range(K - 1):
        d = torch.cdist(X, centers).min(dim=1).values + 1e-6
        centers = torch.cat([centers, X[torch.multinomial(d ** 2 / torch.sum(d ** 2), num_samples=1)]], dim=0)
    assign = None
    for _ in range(max_iters):
        assign = torch.argmin(torch.cdist(X, centers), dim=1)
        new = torch.stack([X[assign == k].mean(dim=0) if bool((assign == k).any()) else centers[k] for k in range(K)], dim=0)
        if torch.equal(new, centers):
            break
        centers = new.clone()
    return centers, assign


class Matcher:
    def __init__(self, encoder, encoder_transforms=None, use_encoder_registers: bool = False, generator=None, input_size=518,
                 num_centers: int = 8, use_box: bool = False, use_points_or_centers: bool = True, sample_range=(4, 6),
                 max_sample_iterations: int = 30, alpha: float = 1.0, beta: float = 0.0, exp: float = 0.0,
                 score_filter_cfg: Optional[dict] = None, num_merging_mask: int = 10,
                 use_negative_priors_from_discarded: bool = False, use_negative_priors_from_cost: bool = False,
                 merge_prompt_types: bool = False, device="cuda:0", **unused):
        self.encoder = encoder
        self.encoder_transform = encoder_transforms     # API parity: the normalisation runs inside the patch extraction
        self.use_encoder_registers = use_encoder_registers
        self.generator = generator
        self.input_size = input_size if isinstance(input_size, tuple) else (input_size, input_size)
        self.num_centers, self.use_box, self.use_points_or_centers = num_centers, use_box, use_points_or_centers
        self.sample_range, self.max_sample_iterations = sample_range, max_sample_iterations
        self.alpha, self.beta, self.exp = alpha, beta, exp
        cfg = {"emd": 0.0, "purity": 0.0, "coverage": 0.0, "score_filter": False, "score": 0.33, "score_norm": 0.1,
               "topk_scores_threshold": 0.0}
        cfg.update(score_filter_cfg or {})
        self.score_filter_cfg = cfg
        self.num_merging_mask = num_merging_mask
        self.use_negative_priors_from_discarded = use_negative_priors_from_discarded
        self.use_negative_priors_from_cost = use_negative_priors_from_cost
        self.merge_prompt_types = merge_prompt_types
        self.device = torch.device(device)
        self.mask_threshold = 0.0                       # generator.predictor.model.mask_threshold of the reference
        self._mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=self.device)
        self._std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=self.device)
        self.rps: Optional[RobustPromptSampler] = None
        self.clear()

    # ---------------------------------------------------------------- state (lines 128-205, 1039-1095)
    def set_reference(self, imgs: torch.Tensor, masks: torch.Tensor):
        imgs = imgs.flatten(0, 1)
        assert imgs.shape[-1] == self.input_size[-1]
        G = imgs.shape[-1] // self.encoder.patch_size
        self.encoder_img_size, self.encoder_feat_size = imgs.shape[-1], G
        masks = masks.clone()
        if masks.sum() == 0:                              # lines 150-153
            sh, sw = masks.shape[-2:]
            masks[..., (sh // 2 - 7):(sh // 2 + 7), (sw // 2 - 7):(sw // 2 + 7)] = 1
        masks = masks.permute(1, 0, 2, 3)                 # ns, 1, h, w
        ns = masks.shape[0]
        # lines 173-180: patch mean > 0, i.e. patch occupancy for non-negative masks
        bits = torch.empty(ns * G * G, dtype=torch.uint8, device=self.device)
        ops.pool_masks(masks.to(self.device)[:, 0].contiguous().float(), G, bits)
        self.ref_imgs, self.ref_masks = imgs, masks
        self.ref_masks_pool = bits.float()
        self.nshot = ns

    def set_target(self, img: torch.Tensor):
        assert img.shape[-2] == self.input_size[0] and img.shape[-1] == self.input_size[1]
        self.tar_img = img
        self.tar_img_np = img.mul(255).byte().squeeze(0).permute(1, 2, 0).cpu().numpy()

    def set_rps(self):
        if self.rps is None:
            assert self.encoder_feat_size is not None
            self.rps = RobustPromptSampler(self.encoder_feat_size, self.sample_range, self.max_sample_iterations)

    def clear(self):
        self.tar_img = self.tar_img_np = None
        self.ref_imgs = self.ref_masks = self.ref_masks_pool = None
        self.nshot = None
        self.stored_ref_feats = self.stored_tar_feat = None
        self.encoder_img_size = self.encoder_feat_size = None
        self.S = self.C = None
        self.unfiltered_generated_masks = None
        self.masks_to_merge = None
        self.metric_filters = {}

    # ---------------------------------------------------------------- lines 251-302
    def extract_img_feats(self):
        if self.stored_ref_feats is not None and self.stored_tar_feat is not None:
            return self.stored_ref_feats, self.stored_tar_feat
        enc = self.encoder
        skip = 1 + (enc.num_register_tokens if self.use_encoder_registers else 0)
        imgs = torch.cat([self.tar_img.to(self.device), self.ref_imgs.to(self.device)], dim=0)   # one batched pass
        X = enc.forward_tokens(imgs, self._mean, self._std)
        rows = X.shape[1] - skip
        F_ = torch.empty(X.shape[0] * rows, enc.embed_dim, dtype=torch.float32, device=self.device)
        ops.l2norm_rows(X, skip, rows, F_)
        self.stored_tar_feat, self.stored_ref_feats = F_[:rows], F_[rows:]
        return self.stored_ref_feats, self.stored_tar_feat

    # ---------------------------------------------------------------- lines 437-445
    def similarity(self):
        """S = ref . tar^T [ns*P, P] and C = (1 - S) / 2, both on the GPU (kept in ``self.S`` / ``self.C``)."""
        ref, tar = self.extract_img_feats()
        R, Pn = ref.shape[0], tar.shape[0]
        S = torch.zeros(R, _pad64(Pn), dtype=torch.float32, device=self.device)
        ops.linear(ref, tar, S[:, :Pn])
        self.S = S[:, :Pn]
        self.C = (1 - self.S) / 2
        return self.S, self.C

    # ---------------------------------------------------------------- lines 698-720
    def score_masks(self, tar_masks, all_points):
        """Per-mask purity / coverage / EMD and ``alpha*emd + beta*purity*coverage**exp``."""
        if self.C is None:
            self.similarity()
        self.set_rps()
        masks = torch.as_tensor(np.asarray(tar_masks) if not torch.is_tensor(tar_masks) else tar_masks).to(self.device)
        purity, coverage, emd = self.rps.get_mask_scores(None, masks, all_points, self.C.cpu(), self.ref_masks_pool)
        scores = self.alpha * emd + self.beta * purity * coverage ** self.exp
        m3 = masks.reshape(masks.shape[0], masks.shape[-2], masks.shape[-1])
        self.unfiltered_generated_masks = (m3 > 0).float()                  # lines 728-729
        return dict(purity=purity, coverage=coverage, emd=emd, scores=scores)

    def get_unfiltered_generated_masks(self):
        return self.unfiltered_generated_masks

    def get_masks_to_merge(self):
        return self.masks_to_merge

    # ---------------------------------------------------------------- lines 419-577, 304-417
    def _patch_points(self, inds) -> np.ndarray:
        """Patch indices -> [x, y] centres in image pixels; duplicates dropped through a set as the reference
        does (its iteration order decides which prompt index means which point), points outside dropped."""
        uniq = torch.tensor(list(set(torch.as_tensor(inds).cpu().tolist())))
        ps, G = self.encoder.patch_size, self.encoder_feat_size
        xs = ((uniq % G) * ps + ps // 2).tolist()
        ys = ((uniq // G) * ps + ps // 2).tolist()
        return np.array([[int(x), int(y)] for x, y in zip(xs, ys) if int(x) < self.input_size[1] and int(y) < self.input_size[0]])

    @staticmethod
    def _halve(n: int) -> int:
        return n // 2 if n > 40 else n

    def _negatives_from_discarded(self, fwd, sim_f, rev, idx_mask):
        discarded = torch.isin(rev[1], idx_mask, invert=True)
        if not bool(discarded.any()):
            return None, None
        cols, sims = fwd[1][discarded], sim_f[discarded]
        keep = self._halve(len(sims))
        order = torch.sort(sims, descending=False)[1][:keep]
        return self._patch_points(cols[order]), keep

    def _negatives_from_cost(self, C_h: torch.Tensor, idx_mask):
        fwd = ops.lap_host(C_h, maximize=True)
        cost_f = C_h[fwd[0], fwd[1]]
        rev = ops.lap_host(C_h.t()[fwd[1]], maximize=True)
        outside = torch.isin(rev[1], idx_mask, invert=True)
        if bool(outside.any()):
            cost_f = cost_f[outside]
        keep = self._halve(len(cost_f))
        order = torch.sort(cost_f, descending=True)[1][:keep]
        return self._patch_points(fwd[1][order]), keep        # indexes the unfiltered columns, as the reference does

    def patch_level_matching(self, ref_feats=None, tar_feat=None):
        """Forward assignment of the reference's foreground patches to target patches, reverse assignment of the
        matched target patches back to reference patches; a pair survives when the way back lands inside the
        reference mask.  Returns (points, negative priors or discarded points, box, S, C, n_points, n_negatives)."""
        S, C = self.similarity()
        S_h = S.cpu()
        fg = self.ref_masks_pool.flatten().bool().cpu()
        idx_mask = fg.nonzero()[:, 0]
        S_fwd = S_h[fg]
        fwd = ops.lap_host(S_fwd, maximize=True)
        sim_f = S_fwd[fwd[0], fwd[1]]
        rev = ops.lap_host(S_h.t()[fwd[1]], maximize=True)
        retain = torch.isin(rev[1], idx_mask)
        pos_cols, neg_cols, sim_pos = fwd[1], fwd[1], sim_f
        if bool(retain.any()):
            pos_cols, neg_cols, sim_pos = fwd[1][retain], fwd[1][~retain], sim_f[retain]
        n_points = self._halve(len(sim_pos))
        best = torch.sort(sim_pos, descending=True)[1][:n_points]
        points = self._patch_points(pos_cols[best])
        discarded = self._patch_points(neg_cols)
        priors, n_neg = [], []
        if self.use_negative_priors_from_discarded:
            pts, k = self._negatives_from_discarded(fwd, sim_f, rev, idx_mask)
            priors.append(pts)
            n_neg.append(k)
        if self.use_negative_priors_from_cost:
            pts, k = self._negatives_from_cost(C.cpu(), idx_mask)
            priors.append(pts)
            n_neg.append(k)
        box = None
        if self.use_box:                                                  # bounding box of the matched points, lines 567-573
            box = np.array([max(points[:, 0].min(), 0), max(points[:, 1].min(), 0),
                            min(points[:, 0].max(), self.input_size[1] - 1), min(points[:, 1].max(), self.input_size[0] - 1)])
        self.sim_scores_after_backward_matching = sim_pos
        return points, (priors if priors else discarded), box, S, C, n_points, n_neg

    def clustering(self, points):
        k = min(self.num_centers, len(points))
        while True:                                                    # re-run until every cluster owns a point
            centers, assign = kmeans_pp(points, k)
            if torch.unique(assign).shape[0] == k:
                return np.array(centers).astype(np.int64)

    # ---------------------------------------------------------------- lines 619-834
    def mask_generation(self, tar_img_np, points, box, all_points, ref_masks_pool, C, negative_points=None, target_mask=None):
        samples, labels = [], []
        if self.use_negative_priors_from_discarded or self.use_negative_priors_from_cost:
            for neg in negative_points:
                s_, l_ = self.rps.sample_points(points, negative_points=neg if neg is not None and len(neg) > 0 else None)
                samples.extend(s_)
                labels.extend(l_)
            if self.merge_prompt_types:
                s_, l_ = self.rps.sample_points(points)
                samples.extend(s_)
                labels.extend(l_)
        else:
            samples, labels = self.rps.sample_points(points)
        recs = self.generator.generate(tar_img_np, select_point_coords=samples, select_point_labels=labels,
                                       select_box=[box] if self.use_box else None,
                                       select_mask_input=target_mask.cpu().numpy() if target_mask is not None else None)
        if len(recs) == 0:
            raise RuntimeError("the mask generator returned no mask for the sampled prompts")
        tar_masks = torch.from_numpy(np.stack([r["segmentation"] for r in recs])).to(self.device)      # n x H x W bool
        point_sets = [r["point_coords"] for r in recs]
        purity, coverage, emd = self.rps.get_mask_scores(points, tar_masks, all_points, C.cpu() if torch.is_tensor(C) else C,
                                                         ref_masks_pool)
        metric = {"purity": purity, "coverage": coverage, "emd": emd}
        scores = self.alpha * emd + self.beta * purity * coverage ** self.exp
        pred = tar_masks.cpu().numpy()
        self.unfiltered_generated_masks = torch.tensor(pred.copy() > 0, dtype=torch.float)
        for name in ("coverage", "emd", "purity"):                       # false-positive fragments
            if self.score_filter_cfg[name] > 0:
                idx = torch.where(metric[name] >= min(self.score_filter_cfg[name], metric[name].max()))[0]
                self.metric_filters[name] = idx
                scores, pred = scores[idx], pred[idx.numpy()]
                point_sets = [point_sets[int(i)] for i in idx]
                metric = {k: v[idx] for k, v in metric.items()}
        if self.score_filter_cfg["score_filter"]:
            dist, rank = torch.sort(1 - scores, descending=False)
            norm = (dist - dist.min()) / (dist.max() + 1e-6)
            ok = dist < self.score_filter_cfg["score"]
            ok[..., 0] = True
            ok = ok * (norm < self.score_filter_cfg["score_norm"])
            chosen = rank[ok][:self.num_merging_mask]
            final_score = scores[chosen].mean()
        else:
            k = min(self.num_merging_mask, scores.size(0))
            top = scores.topk(k)[1]
            top_scores = scores[top].cpu().numpy()
            if self.score_filter_cfg["topk_scores_threshold"] > 0:
                top_scores = top_scores / top_scores.max()
            sel = top_scores > self.score_filter_cfg["topk_scores_threshold"]
            chosen = top[torch.from_numpy(sel)]
            final_score = top_scores[sel].mean()
        merged_from = pred[chosen.numpy()].reshape(-1, pred.shape[-2], pred.shape[-1])
        point_sets = [point_sets[int(i)] for i in chosen]
        self.masks_to_merge = torch.tensor(merged_from.copy() > 0, dtype=torch.float)
        self.number_of_merged_masks = len(merged_from)
        self.number_of_points_used_for_prediction = len({tuple(p) for ps in point_sets for p in ps})
        self.number_of_points_usable_for_prediction = len(all_points)
        merged = merged_from.sum(0) > 0
        self.positive_points_inside_mask = len([p for p in all_points if merged[p[1], p[0]]])
        return torch.tensor(merged[None], device=self.device, dtype=torch.float), final_score

    # ---------------------------------------------------------------- lines 216-249
    def predict(self, target_mask=None):
        if self.generator is None:
            raise RuntimeError("Matcher.predict needs a mask generator (segment_anything.SamAutomaticMaskGenerator)")
        ref_feats, tar_feat = self.extract_img_feats()
        all_points, negative_points, box, S, C, _, _ = self.patch_level_matching(ref_feats, tar_feat)
        if len(all_points) == 0:
            raise RuntimeError("no target patch was matched to the reference mask")
        points = all_points if self.use_points_or_centers else self.clustering(all_points)
        self.set_rps()
        return self.mask_generation(self.tar_img_np, points, box, all_points, self.ref_masks_pool, C, negative_points,
                                    target_mask=target_mask)


class MatcherProposals:
    """``mask_generator`` object of MARS (/root/reference/mars/MARS.py:46-51) that produces the proposals the way the
    reference's stored ``{fold}_{idx}.pt`` files were made: one ``Matcher.predict`` and its unfiltered masks."""

    def __init__(self, matcher: "Matcher"):
        self.matcher = matcher

    def generate(self, support_images, support_masks, query_image) -> torch.Tensor:
        m = self.matcher
        m.clear()
        m.set_reference(support_images, support_masks)
        m.set_target(query_image)
        m.predict()
        return m.get_unfiltered_generated_masks().to(m.device)


def build_matcher_oss(args, state_dict=None, sam_state_dict=None):
    """Builder with the reference's argument names (lines 1297-1372), DINOv2 encoder only; the mask generator is
    built when ``args.sam_size`` is given (``sam_state_dict`` injects weights for synthetic runs)."""
    from ..utils.backbone_loader import BackboneLoader
    kw = dict(img_size=args.input_size, patch_size=14, init_values=1e-5, ffn_layer="mlp", block_chunks=0,
              num_register_tokens=getattr(args, "num_regs", 4), qkv_bias=True, proj_bias=True, ffn_bias=True,
              state_dict=state_dict, dtype=getattr(args, "dtype", torch.float32))
    enc, tr = BackboneLoader.load_backbone(backbone_name="dinov2", backbone_size=getattr(args, "dino_backbone", "vit_large"),
                                           device=args.device, backbone_weights_path=getattr(args, "dino_weights", None),
                                           encoder_kwargs=kw)
    cfg = {"emd": getattr(args, "emd_filter", 0.0), "purity": getattr(args, "purity_filter", 0.0),
           "coverage": getattr(args, "coverage_filter", 0.0), "score_filter": getattr(args, "use_score_filter", False),
           "score": getattr(args, "deep_score_filter", 0.33), "score_norm": getattr(args, "deep_score_norm_filter", 0.1),
           "topk_scores_threshold": getattr(args, "topk_scores_threshold", 0.0)}
    generator = None
    if getattr(args, "sam_size", None) is not None:
        from ..segment_anything import SamAutomaticMaskGenerator, sam_model_registry
        sam = sam_model_registry[args.sam_size](checkpoint=getattr(args, "sam_weights", None), device=args.device,
                                                state_dict=sam_state_dict, dtype=getattr(args, "sam_dtype", torch.float32))
        generator = SamAutomaticMaskGenerator(
            sam, points_per_side=getattr(args, "points_per_side", 64), points_per_batch=64,
            pred_iou_thresh=getattr(args, "pred_iou_thresh", 0.88), stability_score_thresh=getattr(args, "stability_score_thresh", 0.95),
            stability_score_offset=1.0, sel_stability_score_thresh=getattr(args, "sel_stability_score_thresh", 0.0),
            sel_pred_iou_thresh=getattr(args, "iou_filter", 0.0), box_nms_thresh=getattr(args, "box_nms_thresh", 1.0),
            sel_output_layer=getattr(args, "output_layer", 3), output_layer=getattr(args, "dense_multimask_output", 0),
            dense_pred=bool(getattr(args, "use_dense_mask", 0)), multimask_output=getattr(args, "dense_multimask_output", 0) > 0,
            sel_multimask_output=getattr(args, "multimask_output", 1) > 0)
    return Matcher(encoder=enc, encoder_transforms=tr, use_encoder_registers=getattr(args, "num_regs", 4) > 0,
                   input_size=args.input_size, generator=generator, num_centers=getattr(args, "num_centers", 8),
                   use_box=bool(getattr(args, "use_box", False)), use_points_or_centers=bool(getattr(args, "use_points_or_centers", True)),
                   sample_range=tuple(getattr(args, "sample_range", (4, 6))), max_sample_iterations=getattr(args, "max_sample_iterations", 30),
                   alpha=getattr(args, "alpha", 1.0), beta=getattr(args, "beta", 0.0), exp=getattr(args, "exp", 0.0),
                   score_filter_cfg=cfg, num_merging_mask=getattr(args, "num_merging_mask", 10),
                   use_negative_priors_from_discarded=bool(getattr(args, "use_negative_priors_from_discarded", False)),
                   use_negative_priors_from_cost=bool(getattr(args, "use_negative_priors_from_cost", False)),
                   merge_prompt_types=bool(getattr(args, "merge_prompt_types", False)), device=args.device)
