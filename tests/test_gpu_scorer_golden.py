"""The HIP scorer against vectors the reference's own lines produced (tests/golden/scorer_vectors.npz,
written by tests/golden/make_golden_scorer.py from /root/reference's FilteringMergingModule,
VisualVisualAlignmentModule, PriorInformationRefinementModule and MARS).  Model-dependent inputs
(EMD values, AlphaCLIP features) are the fixture's presets, injected through the same three hooks the
fixture overrode in the reference class."""
import importlib
import os
import types

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
V = np.load(os.path.join(GOLDEN, "scorer_vectors.npz"))
T = lambda k: torch.from_numpy(np.asarray(V[k]))            # noqa: E731


def preset_fmm(pkg, dev, prefix, alpha, st, dy):
    FMM = importlib.import_module(pkg.__name__ + ".mars.components.FilteringMergingModule").FilteringMergingModule
    emd, feats, text = T(f"{prefix}_emd"), T(f"{prefix}_feats").to(dev), T(f"{prefix}_text").to(dev)

    class PresetFMM(FMM):
        def _compute_emd_scores(self, cost_matrix, fg_host, bits_host):
            assert bits_host.shape[0] == emd.numel() and int(fg_host.sum()) > 0
            return emd

        def _compute_alphaclip_text_feats(self, text_list):
            self.seen_text = list(text_list)
            return text

        def _compute_alphaclip_vis_feats(self, image, masks, w=None):
            return feats

    model = types.SimpleNamespace(input_resolution=28, cfg=types.SimpleNamespace(embed_dim=feats.shape[1]))
    return PresetFMM(model, None, None, alpha=alpha, static_threshold=st, dynamic_threshold=dy, device=dev, tokenizer=lambda t: t)


@pytest.mark.parametrize("k", range(int(V["merge_cases"])))
def test_merge_masks(pkg, cuda, k):
    """_merge_masks :209-221.  The keep flags follow from the fixture's scores on the host (the kernel that
    decides them is driven by the score fixtures below, both threshold branches and ties included); the OR
    over the kept proposals is the merge kernel's."""
    masks, score = T(f"merge{k}_masks").float().to(cuda), V[f"merge{k}_scores"]
    st, dy = V[f"merge{k}_thresholds"]
    top = score.max()
    lb = dy * top if top < st else st
    keep = torch.from_numpy((score >= lb).astype(np.uint8)).to(cuda)
    merged = torch.empty(masks.shape[1:], dtype=torch.float32, device=cuda)
    pkg.ops.merge_masks(masks, keep, merged)
    assert torch.equal(merged.cpu(), T(f"merge{k}_merged").float())


@pytest.mark.parametrize("name", [str(c) for c in V["score_cases"]])
def test_score_proposals(pkg, cuda, name):
    """_score_proposals :59-140 + _merge_masks through compute() :35-57."""
    alpha, st, dy = [float(x) for x in V[f"score_{name}_thresholds"]]
    fmm = preset_fmm(pkg, cuda, f"score_{name}", alpha, st, dy)
    masks = T(f"score_{name}_masks").float().to(cuda)
    G = V[f"score_{name}_vva"].shape[0]
    S = masks.shape[-1]
    merged = fmm.compute(query_img=torch.zeros(1, 3, S, S, device=cuda), mask_proposals=masks,
                         support_mask=T(f"score_{name}_support").float().to(cuda), cost_matrix=T(f"score_{name}_cost"),
                         patch_features_spatial_dimension=G, vva=T(f"score_{name}_vva").to(cuda), vta=T(f"score_{name}_vta").to(cuda),
                         text=["a object."])
    got = fmm.scores()
    assert np.abs(got["score"].numpy() - V[f"score_{name}_score"]).max() <= 1e-6
    assert got["order"].tolist() == V[f"score_{name}_order"].tolist()
    assert torch.equal(merged.cpu(), T(f"score_{name}_merged").float())


def test_rank_puts_non_finite_scores_last(pkg, cuda):
    """A proposal without patches has no transport problem (NaN from the host solver): it must rank last, stay
    out of the merge and leave ``order`` a permutation."""
    N = 6
    dev = cuda
    emd = torch.tensor([0.7, float("nan"), 0.9, 0.8, float("nan"), 0.6], dtype=torch.float64, device=dev)
    aclip = torch.tensor([0.1, 0.2, 0.3, 0.25, 0.0, 0.15], device=dev)
    cnt = torch.tensor([5, 0, 7, 6, 0, 3], dtype=torch.int32, device=dev)
    svv = torch.tensor([2.0, 0.0, 3.0, 2.5, 0.0, 1.0], device=dev)
    svt = torch.tensor([1.0, 0.0, 4.0, 2.0, 0.0, 1.5], device=dev)
    ucnt = torch.tensor([12], dtype=torch.int32, device=dev)
    score = torch.zeros(N, dtype=torch.float64, device=dev)
    order = torch.full((N,), -1, dtype=torch.int32, device=dev)
    keep = torch.zeros(N, dtype=torch.uint8, device=dev)
    pkg.ops.rank(emd, aclip, cnt, svv, svt, ucnt, N, 0.85, 0.55, 0.95, score, order, keep)
    o = order.tolist()
    assert sorted(o) == list(range(N))
    assert o[-2:] == [1, 4] and o[0] == 2
    assert keep.tolist()[1] == 0 and keep.tolist()[4] == 0 and keep.tolist()[2] == 1
    s = score.cpu()
    assert torch.isnan(s[1]) and torch.isnan(s[4]) and torch.isfinite(s[[0, 2, 3, 5]]).all()


@pytest.mark.parametrize("name", [str(c) for c in V["pir_cases"]])
def test_prior_refinement(pkg, cuda, name):
    """PriorInformationRefinementModule.compute :20-89 (boxes as the fixture gave them to the reference by hand)."""
    PIR = importlib.import_module(pkg.__name__ + ".mars.components.PriorInformationRefinementModule").PriorInformationRefinementModule
    G, regs, L, last_n, heads, four_d = [int(x) for x in V[f"pir_{name}_params"]]
    mod = PIR(box_threshold=float(V[f"pir_{name}_threshold"]), last_n_attention_maps_for_refinement=last_n, device=cuda, num_regs=regs)
    maps = [m.to(cuda) for m in T(f"pir_{name}_attn")]
    out = mod.compute(T(f"pir_{name}_prior").to(cuda), maps)
    ref = T(f"pir_{name}_refined")
    assert float((out.cpu() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))
    # the box mask the kernel derived is the union of the hand-given boxes
    B = torch.zeros(G, G)
    for x0, y0, x1, y1 in V[f"pir_{name}_boxes"].tolist():
        B[y0:y1, x0:x1] = 1
    assert torch.equal(mod._buffers(G * G)["Bm"].cpu().view(G, G), B)


@pytest.mark.parametrize("name", [str(c) for c in V["vva_cases"]])
def test_visual_visual_alignment(pkg, cuda, name):
    """VisualVisualAlignmentModule.compute :42-111 with the refinement replaced by a pass-through, as in the fixture."""
    arch, img = str(V["vva_arch"][0]), int(V["vva_arch"][1])
    cfg = pkg.synth.dino_cfg(arch, img_size=img)
    args = types.SimpleNamespace(input_size=img, num_regs=cfg.num_regs, dino_backbone=arch, vva_refinement_box_threshold=0.8,
                                 last_n_attn_for_vva_refinement=cfg.depth, device=cuda, dtype=torch.float32)
    build = importlib.import_module(pkg.__name__ + ".mars.components.VisualVisualAlignmentModule").build_visual_visual_alignment_component
    mod = build(args, state_dict=pkg.synth.dinov2_state_dict(cfg, seed=1))
    mod.pir.refine = lambda prior, acc: prior
    out = mod.compute(T(f"vva_{name}_support").to(cuda), T(f"vva_{name}_masks").float().to(cuda), T(f"vva_{name}_query").to(cuda))
    P = (img // 14) ** 2
    assert float((mod.similarity_matrix - T(f"vva_{name}_similarity")).abs().max()) <= 1e-5
    assert float((mod.cost_matrix - T(f"vva_{name}_cost")).abs().max()) <= 1e-5
    assert float((mod.mean_attn[:P, :P].cpu() - T(f"vva_{name}_mean_attn")).abs().max()) <= 1e-6
    assert float((mod.vva_raw.cpu() - T(f"vva_{name}_prior")).abs().max()) <= 1e-4
    assert float((out.cpu() - T(f"vva_{name}_out")).abs().max()) <= 1e-4


@pytest.mark.parametrize("name", [str(c) for c in V["mars_cases"]])
def test_mars_glue(pkg, cuda, name):
    """MARS.predict :59-102 around preset alignment maps: nearest resize + min-max of the text prior,
    the AlphaCLIP prompt, and what reaches the scorer."""
    MARS = importlib.import_module(pkg.__name__ + ".mars.MARS").MARS
    cname, desc = [str(x) for x in V[f"mars_{name}_label"]]
    vva = T(f"mars_{name}_vva").to(cuda)
    G = vva.shape[0]
    fmm = preset_fmm(pkg, cuda, f"mars_{name}", 0.85, 0.55, 0.95)

    class Retriever:
        def get_conceptual_information(self, support_images, support_masks):
            return cname, desc

    class PresetVTA:
        def compute(self, query_image, fg_label, bg_labels):
            self.seen = (fg_label, list(bg_labels))
            return T(f"mars_{name}_vta_raw").to(cuda)

    class PresetVVA:
        cost_matrix = T(f"mars_{name}_cost")
        model_embedding_spatial_dimensions = G

        def compute(self, support_imgs, support_masks, query_img):
            return vva

        def clear(self):
            pass

    vta_c = PresetVTA()
    m = MARS(Retriever(), vta_c, PresetVVA(), fmm)
    support = T(f"mars_{name}_support").float().to(cuda)
    first = m.predict(support[:, :, None].repeat(1, 1, 3, 1, 1), support, T(f"mars_{name}_query").to(cuda),
                      T(f"mars_{name}_masks").float().to(cuda))
    assert vta_c.seen == (cname, [])
    assert fmm.seen_text == [str(x) for x in V[f"mars_{name}_prompt"]]
    assert float((m.vta.cpu() - T(f"mars_{name}_vta_scaled")).abs().max()) <= 1e-6
    got = fmm.scores()
    assert np.abs(got["score"].numpy() - V[f"mars_{name}_score"]).max() <= 1e-6
    assert got["order"].tolist() == V[f"mars_{name}_order"].tolist()
    assert torch.equal(first.cpu(), T(f"mars_{name}_pred").float())
    # a prediction the caller holds survives the next call (the reference returns a fresh tensor each time)
    held = first.clone()
    m.predict(support[:, :, None].repeat(1, 1, 3, 1, 1), support, T(f"mars_{name}_query").to(cuda),
              T(f"mars_{name}_masks").float().to(cuda).flip(0).contiguous())
    assert torch.equal(first, held)
