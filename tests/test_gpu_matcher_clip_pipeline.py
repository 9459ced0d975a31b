"""Matcher scoring pieces and CLIPPipeline.process_masks on the GPU against their CPU restatement
(oracle/matcher.py; parity unpinned: cv2 / POT / torchvision are absent, see the oracle header)."""
import importlib

import numpy as np
import pytest
import torch

from tests.helpers import maxdiff, tiny_config
from oracle import matcher as omatch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _points(ep, n=40, seed=0):
    rng = np.random.default_rng(seed)
    S = ep["query_img"].shape[-1]
    ys, xs = np.nonzero(ep["query_mask"][0].numpy() > 0) if "query_mask" in ep else (np.array([S // 2]), np.array([S // 2]))
    pick = rng.integers(0, len(ys), size=n // 2)
    inside = np.stack([xs[pick], ys[pick]], axis=1).astype(np.float64) + rng.random((n // 2, 2))
    anywhere = rng.random((n - n // 2, 2)) * (S + 6) - 3          # a few outside the image: they are clipped
    return np.concatenate([inside, anywhere], axis=0)


@pytest.mark.parametrize("nshot", [1, 2])
def test_matcher_features_similarity_and_mask_scores(pkg, cuda, nshot):
    cfg = tiny_config(pkg, nshot=nshot, nprop=7)
    cfg["args"].device = "cuda:0"
    ep = pkg.synth.make_episode(31 + nshot, size=cfg["size"], nshot=nshot, nprop=7)
    M = importlib.import_module(pkg.__name__ + ".matcher.Matcher")
    cfg["args"].alpha, cfg["args"].beta, cfg["args"].exp = 0.8, 0.3, 0.5
    m = M.build_matcher_oss(cfg["args"], state_dict=cfg["sds"]["dino"])
    m.set_reference(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda))
    m.set_target(ep["query_img"].to(cuda))
    G = cfg["size"] // 14
    om = cfg["oracle_models"]
    pool_ref = omatch.set_reference_pool(ep["support_masks"], 14)
    assert torch.equal(m.ref_masks_pool.cpu(), pool_ref)
    ref_f, tar_f = m.extract_img_feats()
    o_ref, o_tar = omatch.extract_img_feats(om["dino_sd"], om["dino_kw"], ep["support_imgs"][0], ep["query_img"])
    assert maxdiff(ref_f, o_ref) < TOL and maxdiff(tar_f, o_tar) < TOL
    S, C = m.similarity()
    oS, oC = omatch.similarity_and_cost(o_ref, o_tar)
    assert maxdiff(S, oS) < TOL and maxdiff(C, oC) < TOL
    masks = ep["mask_proposals"].clone()
    masks[3] = 0                                                   # an empty proposal: selects every patch
    pts = _points(ep)
    got = m.score_masks(masks[:, None].to(cuda), pts)
    want = omatch.matcher_scores(masks[:, None].numpy(), pts, oC, pool_ref, G, 0.8, 0.3, 0.5)
    for k in ("purity", "coverage", "emd", "scores"):
        assert maxdiff(got[k], want[k]) < TOL, k
    assert torch.equal(m.get_unfiltered_generated_masks().cpu(), (masks > 0).float())
    with pytest.raises(RuntimeError):                               # no mask generator was configured
        m.predict()
    # an all-zero reference mask gets the centred square (lines 150-153)
    m.clear()
    m.set_reference(ep["support_imgs"].to(cuda), torch.zeros_like(ep["support_masks"]).to(cuda))
    assert torch.equal(m.ref_masks_pool.cpu(), omatch.set_reference_pool(torch.zeros_like(ep["support_masks"]), 14))


def test_points_in_masks_and_cosine_table_kernels(pkg, cuda):
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(5, 37, 41, generator=g) > 0.6).float().to(cuda)
    pts = (torch.rand(300, 2, generator=g) * torch.tensor([47.0, 43.0]) - 3.0)
    cnt = torch.zeros(5, dtype=torch.int32, device=cuda)
    pkg.ops.points_in_masks(masks, pts.to(cuda).contiguous(), cnt)
    x = pts[:, 0].long().clamp(0, 40)
    y = pts[:, 1].long().clamp(0, 36)
    want = torch.stack([(masks[k].cpu()[y, x] != 0).sum() for k in range(5)]).int()
    assert torch.equal(cnt.cpu(), want)
    f = torch.randn(9, 48, generator=g).to(cuda)
    t = torch.randn(4, 48, generator=g).to(cuda)
    out = torch.zeros(9, 4, device=cuda)
    pkg.ops.cosine_table(f, t, out, scale=100.0)
    ref = 100.0 * torch.nn.functional.normalize(f.double(), dim=1) @ torch.nn.functional.normalize(t.double(), dim=1).t()
    assert float((out.double() - ref).abs().max()) < 1e-4


class _FixedGenerator:
    def __init__(self, masks):
        self.masks = masks
        self.resets = 0

    def generate(self, image_np):
        assert image_np.ndim == 3 and image_np.shape[2] == 3
        return [{"segmentation": m} for m in self.masks]

    def reset_stored_features(self):
        self.resets += 1


def test_clip_pipeline_process_masks(pkg, cuda):
    cfg = tiny_config(pkg, nprop=5)
    ep = pkg.synth.make_episode(41, size=cfg["size"], nshot=1, nprop=5)
    loader = importlib.import_module(pkg.__name__ + ".utils.backbone_loader")
    model, transforms = loader.BackboneLoader.load_backbone(
        backbone_name="alphaclip", backbone_size="tiny/14", device="cuda:0", backbone_weights_path=None,
        encoder_kwargs={"device": "cuda:0", "state_dict": cfg["sds"]["alphaclip"], "dtype": torch.float32})
    CP = importlib.import_module(pkg.__name__ + ".mars.ClipPipeline")
    seg = [m.numpy() > 0 for m in ep["mask_proposals"][:4]]
    gen = _FixedGenerator(seg)
    pipe = CP.CLIPPipeline(model, transforms, generator=gen, input_size=cfg["size"], device="cuda:0")
    pipe.set_reference(ep["support_imgs"], ep["support_masks"])
    pipe.set_target(ep["query_img"])
    pipe.set_candidate_mask([ep["mask_proposals"][4].numpy() > 0])
    prompts = ["a object.", "a photo of object.", "a photo without object."]
    masks, sims = pipe.process_masks(prompts)
    assert len(masks) == 5 and masks[-1]["type"] == "matcher" and sorted(sims) == [0, 1, 2, 3, 4]
    om = cfg["oracle_models"]
    want = omatch.clip_pipeline_process_masks(om["aclip_sd"], om["aclip_kw"], ep["query_img"][0],
                                              seg + [ep["mask_proposals"][4].numpy() > 0], pkg.synth.tokenize(prompts))
    for i in range(5):
        assert sims[i].shape == (1, 3)
        assert maxdiff(sims[i], want[i]) < 100 * TOL, i          # the table is 100 x cosine
    with pytest.raises(ValueError):
        pipe.predict()                                            # no class name and no VLM
    pipe.reset_clip_pipeline()
    assert gen.resets >= 1 and pipe.target_img is None


@pytest.mark.parametrize("variant", ["plain", "negatives", "score_filter", "box", "cascade"])
def test_matcher_predict_end_to_end(pkg, cuda, variant):
    """Matcher.predict (assignment matching, prompt sets, SAM prompt batches, metric filters, merge) against the
    oracle's restatement under equal seeds.  Parity unpinned except for the assignment step (scipy)."""
    import random
    from oracle import sam as osam
    cfg = tiny_config(pkg, nshot=1)
    a = cfg["args"]
    a.device, a.sam_size = "cuda:0", "tiny"
    a.alpha, a.beta, a.exp = 0.8, 0.3, 0.5
    a.sample_range, a.max_sample_iterations, a.num_merging_mask = (1, 3), 6, 4
    a.emd_filter, a.purity_filter, a.coverage_filter = 0.0, 0.02, 0.3
    if variant == "negatives":
        a.use_negative_priors_from_discarded, a.use_negative_priors_from_cost, a.merge_prompt_types = True, True, True
    if variant == "score_filter":
        a.use_score_filter, a.deep_score_filter, a.deep_score_norm_filter = True, 0.9, 0.8
    a.use_box = variant == "box"
    scfg = pkg.synth.SAM_ARCHS["tiny"]
    sam_sd = pkg.synth.sam_state_dict(scfg, seed=5)
    ep = pkg.synth.make_episode(41, size=cfg["size"], nshot=1, nprop=2)
    M = importlib.import_module(pkg.__name__ + ".matcher.Matcher")
    m = M.build_matcher_oss(a, state_dict=cfg["sds"]["dino"], sam_state_dict=sam_sd)
    m.set_reference(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda))
    m.set_target(ep["query_img"].to(cuda))
    G, size = cfg["size"] // 14, cfg["size"]
    # matched points against the oracle (scipy assignment on the oracle's similarity)
    om = cfg["oracle_models"]
    o_ref, o_tar = omatch.extract_img_feats(om["dino_sd"], om["dino_kw"], ep["support_imgs"][0], ep["query_img"])
    oS, _ = omatch.similarity_and_cost(o_ref, o_tar)
    pool_ref = omatch.set_reference_pool(ep["support_masks"], 14)
    neg = variant == "negatives"
    want_pts, want_neg, oC, _, _ = omatch.patch_level_matching(oS, pool_ref, G, 14, (size, size), neg, neg)
    got_pts, got_neg, got_box, _, _, _, _ = m.patch_level_matching()
    assert np.array_equal(got_pts, want_pts) and len(got_pts) >= 1
    want_box = omatch.points_box(want_pts, (size, size)) if variant == "box" else None
    assert (got_box is None) == (want_box is None) and (want_box is None or np.array_equal(got_box, want_box))
    if neg:
        assert len(got_neg) == len(want_neg) == 2
        for x, y in zip(got_neg, want_neg):
            assert (x is None and y is None) or np.array_equal(x, y)
    else:
        assert np.array_equal(got_neg, want_neg)
    # the cascaded step of the reference: low-resolution logits of an earlier prediction as an extra prompt
    target_mask = None
    if variant == "cascade":
        gs = 4 * (scfg.img_size // scfg.patch_size)
        target_mask = torch.randn(1, gs, gs, generator=torch.Generator().manual_seed(2)) * 3
    random.seed(7)
    torch.manual_seed(7)
    pred, score = m.predict(target_mask=target_mask)
    random.seed(7)
    torch.manual_seed(7)
    gen_kw = dict(points_per_batch=64, dense_pred=False, sel_pred_iou_thresh=0.0, sel_stability_score_thresh=0.0, box_nms_thresh=1.0,
                  sel_output_layer=3, sel_multimask_output=True)
    cfgf = {"emd": 0.0, "purity": 0.02, "coverage": 0.3, "score_filter": variant == "score_filter", "score": 0.9, "score_norm": 0.8}
    merged, final, unfiltered, merged_from = omatch.mask_generation(
        sam_sd, scfg.oracle_kw(), gen_kw, m.tar_img_np, want_pts, want_pts, pool_ref, oC, G, want_neg, use_neg=neg,
        merge_prompt_types=neg, sample_range=(1, 3), max_iterations=6, alpha=0.8, beta=0.3, exp=0.5, score_filter_cfg=cfgf,
        num_merging_mask=4, box=want_box, target_mask=target_mask)
    assert pred.shape == merged.shape == (1, size, size)
    assert m.get_unfiltered_generated_masks().shape == unfiltered.shape
    assert float((m.get_unfiltered_generated_masks() != unfiltered).float().mean()) < 2e-3      # logits at the threshold
    assert m.get_masks_to_merge().shape == merged_from.shape
    assert float((pred.cpu() != merged).float().mean()) < 2e-3
    assert abs(float(score) - final) < 1e-3


def test_mars_on_matcher_proposals(pkg, cuda):
    """The reference's whole workflow in one process: Matcher proposals -> MARS ranking; the ranking agrees with the
    CPU oracle of the ranking path fed the same proposals."""
    import random
    from oracle import pipeline as opipe
    cfg = tiny_config(pkg, nshot=1)
    a = cfg["args"]
    a.device, a.sam_size = "cuda:0", "tiny"
    a.sample_range, a.max_sample_iterations = (1, 2), 6
    sam_sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["tiny"], seed=5)
    ep = pkg.synth.make_episode(43, size=cfg["size"], nshot=1, nprop=2)
    M = importlib.import_module(pkg.__name__ + ".matcher.Matcher")
    matcher = M.build_matcher_oss(a, state_dict=cfg["sds"]["dino"], sam_state_dict=sam_sd)
    MARS = importlib.import_module(pkg.__name__ + ".mars.MARS")
    mars = MARS.build_MARS_fss(a, state_dicts=cfg["sds"])
    mars.mask_generator = M.MatcherProposals(matcher)
    random.seed(3)
    pred = mars.predict(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda), ep["query_img"].to(cuda))
    props = matcher.get_unfiltered_generated_masks()
    assert props.dim() == 3 and props.shape[0] >= 2 and set(props.unique().tolist()) <= {0.0, 1.0}
    # distinct proposals only: exact duplicates tie in the min-max scaling of the ranking
    _, first = np.unique(props.flatten(1).cpu().numpy(), axis=0, return_index=True)
    keep = torch.as_tensor(np.sort(first))
    mars.mask_generator = None
    pred = mars.predict(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda), ep["query_img"].to(cuda), props[keep].to(cuda))
    ref = opipe.mars_predict(cfg["oracle_models"], dict(ep, mask_proposals=props[keep].cpu()), cfg["tokens"])
    got = mars.filtering_merging_component.scores()
    assert maxdiff(got["score"], ref["scores"]["score"]) < TOL
    assert int((pred.cpu() != ref["mask"]).sum()) == 0
