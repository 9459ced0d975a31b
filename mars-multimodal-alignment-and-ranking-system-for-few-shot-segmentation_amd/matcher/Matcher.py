"""The parts of the reference Matcher that run on the ranking path's kernels.

Interface of /root/reference/matcher/Matcher.py:
``Matcher.set_reference`` (128-185), ``set_target`` (187-205), ``extract_img_feats`` (251-302),
the similarity / cost part of ``patch_level_matching`` (437-445), the per-mask scores of
``RobustPromptSampler.get_mask_scores`` (1152-1210) and their combination (698-720), ``clear``.

Prompt sampling (bidirectional assignment, clustering, SAM prompt batches) is not part of this
path: ``predict`` needs a mask generator and matched points from the caller and says so.
The per-mask loop of the reference becomes one pooling launch, one point-count launch and N
transport problems on host threads.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .. import ops
from ..mars.components.VisualVisualAlignmentModule import IMAGENET_MEAN, IMAGENET_STD, _pad64


class RobustPromptSampler:
    """Scores of proposed masks (lines 1140-1210); ``get_mask_scores`` takes all masks at once."""

    def __init__(self, encoder_feat_size: int, sample_range=(4, 6), max_iterations: int = 30, emd_threads: int = 0):
        self.encoder_feat_size = int(encoder_feat_size)
        self.sample_range = sample_range
        self.max_iterations = max_iterations
        self.emd_threads = emd_threads or int(os.environ.get("MARS_EMD_THREADS", "0")) or min(16, os.cpu_count() or 1)

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


class Matcher:
    def __init__(self, encoder, encoder_transforms=None, use_encoder_registers: bool = False, generator=None, input_size=518,
                 num_centers: int = 8, use_box: bool = False, use_points_or_centers: bool = True, sample_range=(4, 6),
                 max_sample_iterations: int = 30, alpha: float = 1.0, beta: float = 0.0, exp: float = 0.0,
                 score_filter_cfg: Optional[dict] = None, num_merging_mask: int = 10, device="cuda:0", **unused):
        self.encoder = encoder
        self.encoder_transform = encoder_transforms     # API parity: the normalisation runs inside the patch extraction
        self.use_encoder_registers = use_encoder_registers
        self.generator = generator
        self.input_size = input_size if isinstance(input_size, tuple) else (input_size, input_size)
        self.num_centers, self.use_box, self.use_points_or_centers = num_centers, use_box, use_points_or_centers
        self.sample_range, self.max_sample_iterations = sample_range, max_sample_iterations
        self.alpha, self.beta, self.exp = alpha, beta, exp
        self.score_filter_cfg = score_filter_cfg if score_filter_cfg is not None else {"emd": 0.0, "purity": 0.0, "coverage": 0.0}
        self.num_merging_mask = num_merging_mask
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

    def predict(self, target_mask=None):
        raise NotImplementedError(
            "Matcher.predict needs the prompt-sampling stage (assignment matching, clustering, SAM prompt batches), which "
            "is outside this path; use extract_img_feats / similarity / score_masks with masks and matched points of your own")


def build_matcher_oss(args, state_dict=None):
    """Builder with the reference's argument names (lines 1297-1372), DINOv2 encoder only."""
    from ..utils.backbone_loader import BackboneLoader
    kw = dict(img_size=args.input_size, patch_size=14, init_values=1e-5, ffn_layer="mlp", block_chunks=0,
              num_register_tokens=getattr(args, "num_regs", 4), qkv_bias=True, proj_bias=True, ffn_bias=True,
              state_dict=state_dict, dtype=getattr(args, "dtype", torch.float32))
    enc, tr = BackboneLoader.load_backbone(backbone_name="dinov2", backbone_size=getattr(args, "dino_backbone", "vit_large"),
                                           device=args.device, backbone_weights_path=getattr(args, "dino_weights", None),
                                           encoder_kwargs=kw)
    cfg = {"emd": getattr(args, "emd_filter", 0.0), "purity": getattr(args, "purity_filter", 0.0),
           "coverage": getattr(args, "coverage_filter", 0.0)}
    return Matcher(encoder=enc, encoder_transforms=tr, use_encoder_registers=getattr(args, "num_regs", 4) > 0,
                   input_size=args.input_size, alpha=getattr(args, "alpha", 1.0), beta=getattr(args, "beta", 0.0),
                   exp=getattr(args, "exp", 0.0), score_filter_cfg=cfg, device=args.device)
