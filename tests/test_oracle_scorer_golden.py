"""The oracle's scorer against vectors the reference's own lines produced
(tests/golden/make_golden_scorer.py ran FilteringMergingModule._score_proposals / _merge_masks,
VisualVisualAlignmentModule.compute, PriorInformationRefinementModule.compute and MARS.predict from
/root/reference and stored their outputs in tests/golden/scorer_vectors.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline as opipe
from tests.helpers import GOLDEN, load_package

V = np.load(os.path.join(GOLDEN, "scorer_vectors.npz"))
T = lambda k: torch.from_numpy(np.asarray(V[k]))            # noqa: E731


@pytest.mark.parametrize("k", range(int(V["merge_cases"])))
def test_merge_masks(k):
    masks, score = T(f"merge{k}_masks").float(), V[f"merge{k}_scores"]
    st, dy = V[f"merge{k}_thresholds"]
    order = np.asarray(sorted(range(len(score)), key=lambda i: score[i], reverse=True))
    merged, _ = opipe.fmm_merge(masks, score, order, static_threshold=float(st), dynamic_threshold=float(dy))
    assert torch.equal(merged, T(f"merge{k}_merged").float())


@pytest.mark.parametrize("name", [str(c) for c in V["score_cases"]])
def test_score_proposals(name):
    masks = T(f"score_{name}_masks").float()
    G = V[f"score_{name}_vva"].shape[0]
    alpha, st, dy = V[f"score_{name}_thresholds"]
    aclip = (T(f"score_{name}_feats") @ T(f"score_{name}_text").t())[:, 0].numpy()
    o = opipe.combine_scores(masks, G, T(f"score_{name}_vva"), T(f"score_{name}_vta"), V[f"score_{name}_emd"], aclip, alpha=float(alpha))
    assert np.array_equal(o["order"], V[f"score_{name}_order"])
    assert np.abs(o["score"] - V[f"score_{name}_score"]).max() == 0.0          # same numpy expressions: bit-equal
    merged, _ = opipe.fmm_merge(masks, o["score"], o["order"], static_threshold=float(st), dynamic_threshold=float(dy))
    assert torch.equal(merged, T(f"score_{name}_merged").float())


@pytest.mark.parametrize("name", [str(c) for c in V["pir_cases"]])
def test_prior_refinement(name):
    G, regs, L, last_n, heads, four_d = [int(x) for x in V[f"pir_{name}_params"]]
    prior, thr = T(f"pir_{name}_prior"), float(V[f"pir_{name}_threshold"])
    maps = list(T(f"pir_{name}_attn"))
    boxes = opipe.scoremap_boxes(prior.numpy(), thr)
    assert sorted(map(tuple, boxes.tolist())) == sorted(map(tuple, V[f"pir_{name}_boxes"].tolist()))
    A = opipe.mean_attention(maps, last_n, regs)
    out = opipe.pir_refine(prior, A, thr)
    assert float((out - T(f"pir_{name}_refined")).abs().max()) <= 1e-7


@pytest.mark.parametrize("name", [str(c) for c in V["vva_cases"]])
def test_visual_visual_alignment(name):
    pkg = load_package(require_native=False)
    arch, img = str(V["vva_arch"][0]), int(V["vva_arch"][1])
    cfg = pkg.synth.dino_cfg(arch, img_size=img)
    sd = pkg.synth.dinov2_state_dict(cfg, seed=1)
    kw = dict(depth=cfg.depth, heads=cfg.num_heads, patch=14, num_regs=cfg.num_regs, ffn="mlp")
    o = opipe.vva_compute(sd, kw, T(f"vva_{name}_support"), T(f"vva_{name}_masks").float(), T(f"vva_{name}_query"), last_n=cfg.depth)
    assert float((o["similarity"] - T(f"vva_{name}_similarity")).abs().max()) <= 1e-6
    assert float((o["cost"] - T(f"vva_{name}_cost")).abs().max()) <= 1e-6
    assert float((o["vva_raw"] - T(f"vva_{name}_prior")).abs().max()) <= 1e-5
    assert float((o["mean_attn"] - T(f"vva_{name}_mean_attn")).abs().max()) <= 1e-6
    # the fixture's refinement was a pass-through: its output is the min-max of the prior (line 109)
    assert float((opipe.minmax(o["vva_raw"]) - T(f"vva_{name}_out")).abs().max()) <= 1e-5


@pytest.mark.parametrize("name", [str(c) for c in V["mars_cases"]])
def test_mars_glue(name):
    vta_raw, vva = T(f"mars_{name}_vta_raw"), T(f"mars_{name}_vva")
    G = vva.shape[0]
    vta = opipe.minmax(torch.nn.functional.interpolate(vta_raw[None, None], (G, G), mode="nearest")[0, 0])
    assert torch.equal(vta, T(f"mars_{name}_vta_scaled"))
    cname, desc = [str(x) for x in V[f"mars_{name}_label"]]
    assert opipe.alphaclip_prompt(cname, desc) == [str(x) for x in V[f"mars_{name}_prompt"]]
    masks = T(f"mars_{name}_masks").float()
    aclip = (T(f"mars_{name}_feats") @ T(f"mars_{name}_text").t())[:, 0].numpy()
    o = opipe.combine_scores(masks, G, vva, vta, V[f"mars_{name}_emd"], aclip)
    assert np.array_equal(o["order"], V[f"mars_{name}_order"])
    assert np.abs(o["score"] - V[f"mars_{name}_score"]).max() == 0.0
    merged, _ = opipe.fmm_merge(masks, o["score"], o["order"])
    assert torch.equal(merged, T(f"mars_{name}_pred").float())
