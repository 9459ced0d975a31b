"""Shared fixtures: one tiny end-to-end configuration that both the oracle and the HIP
path can run in seconds, built from the same seeded state dicts."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def tiny_config(pkg, size: int = 126, nshot: int = 1, nprop: int = 6, seed: int = 11,
                dino="vit_tiny2", clip="tiny/16", aclip="tiny/14"):
    synth = pkg.synth
    dcfg = synth.dino_cfg(dino, img_size=size)
    ccfg = synth.CLIP_ARCHS[clip]
    acfg = synth.CLIP_ARCHS[aclip]
    sds = dict(dino=synth.dinov2_state_dict(dcfg, seed), clip=synth.clip_state_dict(ccfg, seed),
               alphaclip=synth.clip_state_dict(acfg, seed, alpha=True))
    oracle_models = dict(
        dino_sd=sds["dino"], dino_kw=dict(depth=dcfg.depth, heads=dcfg.num_heads, patch=14, num_regs=dcfg.num_regs, ffn=dcfg.ffn),
        clip_sd=sds["clip"], clip_kw=dict(layers=ccfg.vision_layers, heads=ccfg.vision_heads, patch=ccfg.vision_patch_size,
                                          text_layers=ccfg.transformer_layers, text_heads=ccfg.transformer_heads),
        aclip_sd=sds["alphaclip"], aclip_kw=dict(layers=acfg.vision_layers, heads=acfg.vision_heads, patch=acfg.vision_patch_size,
                                                 resolution=acfg.image_resolution, text_layers=acfg.transformer_layers,
                                                 text_heads=acfg.transformer_heads))
    tokens = dict(fg=synth.tokenize(["a photo of object."]), bg=synth.tokenize(["a photo without object."]),
                  alphaclip=synth.tokenize(["a object."]))
    args = types.SimpleNamespace(
        input_size=size, num_regs=dcfg.num_regs, dino_backbone=dino, vta_backbone=clip, alphaclip_backbone=aclip,
        vva_refinement_box_threshold=0.8, last_n_attn_for_vva_refinement=24, vta_refinement_box_threshold=0.4,
        last_n_attn_for_vta_refinement=8, static_threshold=0.55, dynamic_threshold=0.95, alpha_coverage=0.85,
        device="cuda:0", dtype=torch.float32, class_name="object", class_description="")
    return dict(sds=sds, oracle_models=oracle_models, tokens=tokens, args=args, size=size, nshot=nshot, nprop=nprop,
                dcfg=dcfg, ccfg=ccfg, acfg=acfg)


def maxdiff(a, b) -> float:
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max())


def run_tiny_episode_parity(pkg, device="cuda:0", seed: int = 3, dtype=torch.float32, tol: float = 1e-4) -> dict:
    """One tiny episode through the HIP path and through the oracle; returns the deviations and
    raises if the fp32 tolerance (north-star: 1e-4) is exceeded."""
    from oracle import pipeline as opipe
    cfg = tiny_config(pkg)
    cfg["args"].device = device
    cfg["args"].dtype = dtype
    ep = pkg.synth.make_episode(seed, size=cfg["size"], nshot=cfg["nshot"], nprop=cfg["nprop"])
    from importlib import import_module
    MARS = import_module(pkg.__name__ + ".mars.MARS")
    mars = MARS.build_MARS_fss(cfg["args"], state_dicts=cfg["sds"])
    dev = torch.device(device)
    pred = mars.predict(ep["support_imgs"].to(dev), ep["support_masks"].to(dev), ep["query_img"].to(dev),
                        ep["mask_proposals"].to(dev))
    got = mars.filtering_merging_component.scores()
    ref = opipe.mars_predict(cfg["oracle_models"], ep, cfg["tokens"])
    rep = dict(
        vva=maxdiff(mars.vva, ref["vva"]["vva"]), vta=maxdiff(mars.vta, ref["vta_scaled"]),
        emd=maxdiff(got["emd"], ref["scores"]["emd"]), alphaclip=maxdiff(got["aclip"], ref["scores"]["alphaclip"]),
        score=maxdiff(got["score"], ref["scores"]["score"]),
        mask_mismatch=int((pred.cpu() != ref["mask"]).sum()),
        order_equal=bool((got["order"].numpy() == ref["scores"]["order"]).all()),
    )
    if dtype == torch.float32:
        for k in ("vva", "vta", "emd", "alphaclip", "score"):
            assert rep[k] <= tol, (k, rep)
    return rep
