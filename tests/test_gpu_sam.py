"""Segment-Anything proposal stage on the HIP kernels against oracle/sam.py and the committed
vectors made from the reference's segment_anything.modeling modules (tests/golden/make_golden.py).

fp32 tolerances: the GEMMs accumulate in another order than the CPU's, so logits agree to ~1e-4
and a mask bit may flip only where |logit| is within that distance of the threshold.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
TOL = 2e-4


def maxdiff(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max())


@pytest.fixture(scope="module")
def pkg():
    return load_package(require_native=True)


@pytest.fixture(scope="module", params=["tiny", "tiny80"])
def setup(request, pkg):
    from oracle import sam as osam
    arch = request.param
    cfg = pkg.synth.SAM_ARCHS[arch]
    sd = pkg.synth.sam_state_dict(cfg, seed=5)
    sam = pkg.segment_anything.sam_model_registry[arch](state_dict=sd, device="cuda", dtype=torch.float32)
    return arch, cfg, sd, sam, osam


def test_encoder_matches_reference_vectors(setup, pkg):
    arch, cfg, sd, sam, osam = setup
    pred = pkg.segment_anything.SamPredictor(sam)
    pred.set_image(GOLD[f"sam_{arch}_image"])
    assert pred.input_size == osam.preprocess_shape(*GOLD[f"sam_{arch}_image"].shape[:2], cfg.img_size)
    assert maxdiff(pred.features, GOLD[f"sam_{arch}_features"]) < TOL
    # module-style call on an already normalised, padded batch
    state = osam.set_image(sd, cfg.oracle_kw(), GOLD[f"sam_{arch}_image"])
    img = torch.as_tensor(osam.apply_image(GOLD[f"sam_{arch}_image"], cfg.img_size)).permute(2, 0, 1)[None].float()
    out = sam.image_encoder(osam.preprocess(img, cfg.img_size).cuda())
    assert maxdiff(out, state["features"]) < TOL


def test_prompt_encoder(setup):
    arch, cfg, sd, sam, osam = setup
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(5, 2, 2, generator=g) * cfg.img_size
    lab = torch.tensor([[1, 0], [1, 1], [0, -1], [1, -1], [0, 0]])
    ref = osam.embed_points(sd, pts, lab.float(), (cfg.img_size, cfg.img_size))
    sparse, dense = sam.prompt_encoder(points=(pts.cuda(), lab.cuda()), boxes=None, masks=None)
    assert maxdiff(sparse, ref) < 2e-5
    G = cfg.img_size // cfg.patch_size
    assert maxdiff(sam.prompt_encoder.get_dense_pe(), osam.dense_pe(sd, (G, G))) < 2e-5
    assert dense.shape == (5, cfg.out_chans, G, G)
    assert maxdiff(dense[2, :, 1, 1], sd["prompt_encoder.no_mask_embed.weight"][0]) == 0.0


def test_decoder_and_postprocess_match_reference_vectors(setup, pkg):
    arch, cfg, sd, sam, osam = setup
    pred = pkg.segment_anything.SamPredictor(sam)
    # start from the reference's embedding so the decoder is checked on its own
    pred.set_image(GOLD[f"sam_{arch}_image"], stored_features=torch.as_tensor(GOLD[f"sam_{arch}_features"]))
    pts = torch.as_tensor(GOLD[f"sam_{arch}_points"]).float()
    masks, iou, low = pred.predict_torch(pts[:, None, :].cuda(), torch.ones(len(pts), 1, dtype=torch.int32).cuda(), return_logits=True)
    assert maxdiff(low, GOLD[f"sam_{arch}_low_res"]) < TOL
    assert maxdiff(iou, GOLD[f"sam_{arch}_iou"]) < TOL
    assert maxdiff(masks, GOLD[f"sam_{arch}_masks"]) < TOL
    # module-style decoder call, single-mask output
    G = cfg.img_size // cfg.patch_size
    feats = torch.as_tensor(GOLD[f"sam_{arch}_features"]).cuda()
    sparse, dense = sam.prompt_encoder(points=(pts[:, None, :].cuda(), torch.ones(len(pts), 1).cuda()), boxes=None, masks=None)
    low1, iou1 = sam.mask_decoder(image_embeddings=feats, image_pe=sam.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sparse,
                                  dense_prompt_embeddings=dense, multimask_output=False)
    state = dict(features=torch.as_tensor(GOLD[f"sam_{arch}_features"]))
    ref_sparse = osam.embed_points(sd, pts[:, None, :], torch.ones(len(pts), 1), (cfg.img_size, cfg.img_size))
    ref_dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(len(pts), -1, G, G)
    rlow, riou = osam.mask_decoder(sd, state["features"], osam.dense_pe(sd, (G, G)), ref_sparse, ref_dense, cfg.oracle_kw(), False)
    assert low1.shape == rlow.shape and maxdiff(low1, rlow) < TOL and maxdiff(iou1, riou) < TOL
    # end to end (own encoder): binary masks differ only where the logit is at the threshold
    pred.set_image(GOLD[f"sam_{arch}_image"])
    bm, _, _ = pred.predict_torch(pts[:, None, :].cuda(), torch.ones(len(pts), 1, dtype=torch.int32).cuda())
    ref = torch.as_tensor(GOLD[f"sam_{arch}_masks"])
    flips = (bm.cpu() != (ref > 0))
    assert bool((ref[flips].abs() < 5 * TOL).all())


def test_preprocess_and_transforms(setup, pkg):
    arch, cfg, sd, sam, osam = setup
    img = GOLD[f"sam_{arch}_image"]
    t = torch.as_tensor(osam.apply_image(img, cfg.img_size)).permute(2, 0, 1)[None].float()
    assert maxdiff(sam.preprocess(t.cuda()), osam.preprocess(t, cfg.img_size)) < 1e-6
    tr = pkg.segment_anything.predictor.ResizeLongestSide(cfg.img_size)
    c = np.array([[3.0, 5.0], [60.5, 20.25]])
    assert np.array_equal(tr.apply_coords(c, img.shape[:2]), osam.apply_coords(c, img.shape[:2], cfg.img_size))
    g = torch.Generator().manual_seed(4)
    low = torch.randn(3, 2, 4 * (cfg.img_size // cfg.patch_size), 4 * (cfg.img_size // cfg.patch_size), generator=g)
    insz = osam.preprocess_shape(*img.shape[:2], cfg.img_size)
    for orig in (img.shape[:2], (57, 41)):
        got = sam.postprocess_masks(low.cuda(), insz, orig)
        assert maxdiff(got, osam.postprocess_masks(low, cfg.img_size, insz, orig)) < 2e-5


def test_mask_statistics_and_nms(pkg):
    from oracle import sam as osam
    ops = pkg.ops
    g = torch.Generator().manual_seed(8)
    m = torch.randn(7, 33, 50, generator=g) * 2
    m[3] = -5.0                                                       # empty mask
    m[4, 10:12, 7:30] = 9.0
    stats = torch.empty(7, 6, dtype=torch.int32, device="cuda")
    ops.mask_stats(m.cuda(), 0.0, 1.0, stats)
    st = stats.cpu()
    assert torch.equal(st[:, 0], (m > 1.0).sum((1, 2)).int()) and torch.equal(st[:, 1], (m > -1.0).sum((1, 2)).int())
    assert torch.equal(st[:, 2:6].long(), osam.mask_to_box(m > 0.0))
    b = torch.empty_like(m, device="cuda")
    ops.threshold_masks(m.cuda(), 0.0, b)
    assert torch.equal(b.cpu() > 0, m > 0)
    boxes = torch.rand(200, 4, generator=g) * 50
    boxes[:, 2:] += boxes[:, :2]
    scores = torch.rand(200, generator=g)
    scores[10] = scores[20]                                          # a tie: stable order
    for thr in (0.3, 0.7):
        assert torch.equal(ops.box_nms_host(boxes, scores, thr), osam.box_nms(boxes, scores, thr))


def test_generator_matches_oracle(pkg):
    from oracle import sam as osam
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS["tiny"]
    sd = pkg.synth.sam_state_dict(cfg, seed=5)
    sam = sa.sam_model_registry["tiny"](state_dict=sd, device="cuda")
    img = GOLD["sam_tiny_image"]
    kw = dict(points_per_side=4, points_per_batch=5, pred_iou_thresh=-10.0, stability_score_thresh=0.0, box_nms_thresh=0.7)
    recs = sa.SamAutomaticMaskGenerator(sam, **kw).generate(img)
    ref = osam.generate(sd, cfg.oracle_kw(), img, **kw)
    assert len(recs) == len(ref) >= 1
    for a, b in zip(recs, ref):
        assert a["point_coords"] == b["point_coords"] and abs(a["predicted_iou"] - b["predicted_iou"]) < TOL
        assert float((a["segmentation"] != b["segmentation"]).mean()) < 2e-3
        assert np.abs(np.array(a["bbox"]) - np.array(b["bbox"])).max() <= 1 and abs(a["area"] - b["area"]) <= 4
        assert maxdiff(a["low_res_masks"], b["low_res_masks"]) < TOL
        assert abs(a["stability_score"] - b["stability_score"]) < 5e-3 and a["crop_box"] == [0.0, 0.0, float(img.shape[1]), float(img.shape[0])]
    # filters: an IoU threshold nothing reaches gives no record
    assert sa.SamAutomaticMaskGenerator(sam, points_per_side=4, points_per_batch=16, pred_iou_thresh=10.0).generate(img) == []
    # stability threshold at the oracle's median keeps the same points
    kw2 = dict(kw, stability_score_thresh=float(np.median([r["stability_score"] for r in ref])) + 1e-3)
    a2, b2 = sa.SamAutomaticMaskGenerator(sam, **kw2).generate(img), osam.generate(sd, cfg.oracle_kw(), img, **kw2)
    assert [r["point_coords"] for r in a2] == [r["point_coords"] for r in b2]


def test_proposals_feed_the_ranking_pipeline(pkg):
    sa = pkg.segment_anything
    sam = sa.sam_model_registry["tiny"](state_dict=pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["tiny"], seed=5), device="cuda")
    amg = sa.SamAutomaticMaskGenerator(sam, points_per_side=4, points_per_batch=16, pred_iou_thresh=-10.0, stability_score_thresh=0.0)
    gen = sa.ProposalGenerator(amg, n_proposals=6)
    q = torch.as_tensor(GOLD["sam_tiny_image"]).permute(2, 0, 1)[None].float().div(255).cuda()
    props = gen.generate(None, None, q)
    assert props.shape == (6,) + tuple(q.shape[-2:]) and props.dtype == torch.float32 and props.is_cuda
    assert set(props.unique().tolist()) <= {0.0, 1.0}


def test_episode_with_generated_proposals(pkg):
    """MARS(mask_generator=...) end to end: proposals come from the generator, the ranking path then
    agrees with the CPU oracle of the ranking path fed the same proposals."""
    import importlib
    from oracle import pipeline as opipe
    from tests.helpers import tiny_config
    sa = pkg.segment_anything
    cfg = tiny_config(pkg)
    ep = pkg.synth.make_episode(11, size=cfg["size"], nshot=cfg["nshot"], nprop=cfg["nprop"])
    sam = sa.sam_model_registry["tiny"](state_dict=pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["tiny"], seed=5), device="cuda")
    # no suppression: the ranking needs distinct proposals (min-max scaling of exactly tied scores is 0/0)
    amg = sa.SamAutomaticMaskGenerator(sam, points_per_side=4, points_per_batch=16, pred_iou_thresh=-10.0, stability_score_thresh=0.0,
                                       box_nms_thresh=1.0)
    gen = sa.ProposalGenerator(amg, n_proposals=cfg["nprop"])
    MARS = importlib.import_module(pkg.__name__ + ".mars.MARS")
    mars = MARS.build_MARS_fss(cfg["args"], state_dicts=cfg["sds"])
    mars.mask_generator = gen
    dev = "cuda"
    pred = mars.predict(ep["support_imgs"].to(dev), ep["support_masks"].to(dev), ep["query_img"].to(dev))
    props = gen.generate(None, None, ep["query_img"].to(dev))
    assert len({int(v) for v in props.sum((1, 2)).tolist()}) == cfg["nprop"]
    ep2 = dict(ep, mask_proposals=props.cpu())
    ref = opipe.mars_predict(cfg["oracle_models"], ep2, cfg["tokens"])
    got = mars.filtering_merging_component.scores()
    assert maxdiff(got["score"], ref["scores"]["score"]) < 1e-4
    assert got["order"].tolist() == ref["scores"]["order"].tolist()
    assert int((pred.cpu() != ref["mask"]).sum()) == 0


def test_bf16_run_agrees_with_fp32(pkg):
    """bf16 GEMM operands cannot meet the fp32 tolerance; the run is judged on mask overlap and IoU-head agreement."""
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS["tiny"]
    sd = pkg.synth.sam_state_dict(cfg, seed=5)
    img = GOLD["sam_tiny_image"]
    pts = torch.as_tensor(GOLD["sam_tiny_points"]).float()[:, None, :].cuda()
    lab = torch.ones(pts.shape[0], 1, dtype=torch.int32).cuda()
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        pred = sa.SamPredictor(sa.sam_model_registry["tiny"](state_dict=sd, device="cuda", dtype=dt))
        pred.set_image(img)
        out[dt] = pred.predict_torch(pts, lab, return_logits=True)
    m32, i32, _ = out[torch.float32]
    m16, i16, _ = out[torch.bfloat16]
    assert float((i32 - i16).abs().max()) < 0.05
    scale = float(m32.abs().max())
    assert float((m32 - m16).abs().max()) < 0.05 * scale
    a, b = m32 > 0, m16 > 0
    inter, union = (a & b).flatten(1).sum(1).float(), (a | b).flatten(1).sum(1).float().clamp(min=1)
    assert float((inter / union)[union > 50].min()) > 0.9


def test_box_prompts_match_reference_vectors(setup, pkg):
    arch, cfg, sd, sam, osam = setup
    boxes = torch.as_tensor(GOLD[f"sam_{arch}_boxes"]).cuda()
    sparse, _ = sam.prompt_encoder(points=None, boxes=boxes, masks=None)
    assert maxdiff(sparse, GOLD[f"sam_{arch}_box_sparse"]) < 2e-5
    pred = pkg.segment_anything.SamPredictor(sam)
    pred.set_image(GOLD[f"sam_{arch}_image"], stored_features=torch.as_tensor(GOLD[f"sam_{arch}_features"]))
    pts = torch.as_tensor(GOLD[f"sam_{arch}_box_points"]).cuda()
    lab = torch.tensor([[1, 0]] * 3, dtype=torch.int32).cuda()
    _, iou, low = pred.predict_torch(pts, lab, boxes=boxes, return_logits=True)
    assert maxdiff(low, GOLD[f"sam_{arch}_box_low_res"]) < TOL and maxdiff(iou, GOLD[f"sam_{arch}_box_iou"]) < TOL
    # box alone: two corner tokens and no padding token
    _, iou_b, low_b = pred.predict_torch(None, None, boxes=boxes, return_logits=True)
    state = dict(features=torch.as_tensor(GOLD[f"sam_{arch}_features"]), input_size=pred.input_size, original_size=pred.original_size)
    G = cfg.img_size // cfg.patch_size
    sp = osam.embed_boxes(sd, boxes.cpu(), (cfg.img_size, cfg.img_size))
    dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(3, -1, G, G)
    rlow, riou = osam.mask_decoder(sd, state["features"], osam.dense_pe(sd, (G, G)), sp, dense, cfg.oracle_kw(), True)
    assert maxdiff(low_b, rlow) < TOL and maxdiff(iou_b, riou) < TOL
    tr = pkg.segment_anything.predictor.ResizeLongestSide(cfg.img_size)
    b = np.array([[3.0, 4.0, 50.0, 61.5]])
    hw = GOLD[f"sam_{arch}_image"].shape[:2]
    assert np.allclose(tr.apply_boxes(b, hw), osam.apply_coords(b.reshape(-1, 2, 2), hw, cfg.img_size).reshape(-1, 4))


def test_mask_prompts_match_reference_vectors(setup, pkg):
    arch, cfg, sd, sam, osam = setup
    G = cfg.img_size // cfg.patch_size
    mask_in = torch.as_tensor(GOLD[f"sam_{arch}_low_res"])[:3, :1].cuda()
    pts = torch.as_tensor(GOLD[f"sam_{arch}_box_points"]).cuda()
    lab = torch.tensor([[1, 0]] * 3, dtype=torch.int32).cuda()
    _, dense = sam.prompt_encoder(points=(pts, lab), boxes=None, masks=mask_in)
    assert dense.shape == (3, cfg.out_chans, G, G) and maxdiff(dense, GOLD[f"sam_{arch}_mask_dense"]) < 2e-5
    pred = pkg.segment_anything.SamPredictor(sam)
    pred.set_image(GOLD[f"sam_{arch}_image"], stored_features=torch.as_tensor(GOLD[f"sam_{arch}_features"]))
    _, iou, low = pred.predict_torch(pts, lab, mask_input=mask_in, return_logits=True)
    assert maxdiff(low, GOLD[f"sam_{arch}_mask_low_res"]) < TOL and maxdiff(iou, GOLD[f"sam_{arch}_mask_iou"]) < TOL
    # one mask for the whole batch, and the module-style decoder call with a dense embedding
    _, iou1, low1 = pred.predict_torch(pts, lab, mask_input=mask_in[:1], return_logits=True)
    assert maxdiff(low1[0], low[0]) < 1e-6
    feats = torch.as_tensor(GOLD[f"sam_{arch}_features"]).cuda()
    sparse, _ = sam.prompt_encoder(points=(pts, lab), boxes=None, masks=None)
    lowm, ioum = sam.mask_decoder(image_embeddings=feats, image_pe=sam.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sparse,
                                  dense_prompt_embeddings=dense, multimask_output=True)
    assert maxdiff(lowm, GOLD[f"sam_{arch}_mask_low_res"]) < TOL and maxdiff(ioum, GOLD[f"sam_{arch}_mask_iou"]) < TOL


def test_real_geometry_against_oracle(pkg):
    """1024-pixel input, 64 x 64 tokens, 14 x 14 windows (grid padded to 70), 256-channel decoder, 8 decoder heads:
    the shapes of the released models on a narrow two-block tower the CPU oracle finishes in seconds."""
    from oracle import sam as osam
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS["geom"]
    sd = pkg.synth.sam_state_dict(cfg, seed=9)
    sam = sa.sam_model_registry["geom"](state_dict=sd, device="cuda")
    img = (pkg.synth.rand(4, "geom_img", (300, 400, 3)) * 255).to(torch.uint8).numpy()
    pred = sa.SamPredictor(sam)
    pred.set_image(img)
    state = osam.set_image(sd, cfg.oracle_kw(), img)
    assert pred.input_size == state["input_size"] == (768, 1024)
    assert maxdiff(pred.features, state["features"]) < TOL
    pts = torch.tensor([[[100.0, 200.0]], [[512.0, 384.0]], [[900.5, 50.25]]])
    lab = torch.ones(3, 1)
    masks, iou, low = pred.predict_torch(pts.cuda(), lab.int().cuda(), return_logits=True)
    rm, ri, rl = osam.predict_torch(sd, cfg.oracle_kw(), state, pts, lab)
    assert maxdiff(low, rl) < 5 * TOL and maxdiff(iou, ri) < 5 * TOL and maxdiff(masks, rm) < 5 * TOL


def test_generator_with_crop_layers_and_rle(pkg):
    """One extra crop layer (four overlapping crops, their own embeddings and point grids, edge filter, NMS between
    crops) and run-length output, against the oracle."""
    from oracle import sam as osam
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS["tiny"]
    sd = pkg.synth.sam_state_dict(cfg, seed=5)
    sam = sa.sam_model_registry["tiny"](state_dict=sd, device="cuda")
    # Masks of random weights cover their whole crop at threshold 0, and the crop-edge filter then drops every mask
    # of the inner crops; a raised threshold leaves small blobs, so two crops contribute.
    sam.mask_threshold = 0.5
    img = np.random.default_rng(3).integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    kw = dict(points_per_side=4, points_per_batch=7, pred_iou_thresh=-10.0, stability_score_thresh=0.0, box_nms_thresh=0.9,
              crop_n_layers=1, crop_nms_thresh=0.95, crop_overlap_ratio=0.3, crop_n_points_downscale_factor=2,
              output_mode="uncompressed_rle")
    recs = sa.SamAutomaticMaskGenerator(sam, **kw).generate(img)
    ref = osam.generate(sd, cfg.oracle_kw(), img, mask_threshold=0.5, **kw)
    assert len(recs) == len(ref) >= 2 and len({tuple(r["crop_box"]) for r in ref}) >= 2       # more than one crop contributes
    amg = sa.amg
    for a, b in zip(recs, ref):
        assert a["point_coords"] == b["point_coords"] and a["crop_box"] == b["crop_box"]
        assert abs(a["predicted_iou"] - b["predicted_iou"]) < TOL
        ma, mb = amg.rle_to_mask(a["segmentation"]), osam.rle_to_mask(b["segmentation"])
        assert ma.shape == img.shape[:2] and float((ma != mb).mean()) < 2e-3
        assert a["area"] == amg.area_from_rle(a["segmentation"])
        assert np.abs(np.array(a["bbox"]) - np.array(b["bbox"])).max() <= 1


def test_generator_small_regions_and_coco_rle(pkg):
    """min_mask_region_area (holes and islands below the area removed, boxes recomputed, NMS preferring untouched masks)
    and the compressed COCO run-length output, against the same post-processing applied to the plain run's records."""
    from scipy import ndimage
    sa = pkg.segment_anything
    amg = sa.amg
    cfg = pkg.synth.SAM_ARCHS["tiny"]
    sd = pkg.synth.sam_state_dict(cfg, seed=5)
    sam = sa.sam_model_registry["tiny"](state_dict=sd, device="cuda")
    sam.mask_threshold = 0.5
    img = np.random.default_rng(3).integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    kw = dict(points_per_side=4, points_per_batch=7, pred_iou_thresh=-10.0, stability_score_thresh=0.0, box_nms_thresh=0.9)
    plain = sa.SamAutomaticMaskGenerator(sam, **kw).generate(img)
    area = 60
    recs = sa.SamAutomaticMaskGenerator(sam, min_mask_region_area=area, output_mode="coco_rle", **kw).generate(img)
    assert 1 <= len(recs) <= len(plain)
    # the reference's procedure on the plain records (automatic_mask_generator.py:559-607)
    new, scores = [], []
    for r in plain:
        m1, c1 = amg.remove_small_regions(r["segmentation"], area, "holes")
        m2, c2 = amg.remove_small_regions(m1, area, "islands")
        new.append(m2)
        scores.append(float(not (c1 or c2)))
    assert 0.0 in scores, "the case must exercise a changed mask"
    boxes = torch.from_numpy(amg.batched_mask_to_box(np.stack(new)))
    keep = pkg.ops.box_nms_host(boxes, torch.tensor(scores), 0.9).tolist()
    assert len(keep) == len(recs)
    eight = np.ones((3, 3), int)
    for a, i in zip(recs, keep):
        m = amg.rle_to_mask(amg.coco_decode_rle(a["segmentation"]))
        want = new[i] if scores[i] == 0.0 else plain[i]["segmentation"]
        assert np.array_equal(m, want) and a["area"] == int(want.sum()) and a["point_coords"] == plain[i]["point_coords"]
        if scores[i] == 0.0:
            x0, y0, x1, y1 = boxes[i].tolist()
            assert a["bbox"] == [x0, y0, x1 - x0, y1 - y0]
            lab, n = ndimage.label(m, structure=eight)
            sizes = np.bincount(lab.reshape(-1))[1:]
            assert n == 1 or sizes.min() >= area               # no island below the area is left (one may stay if all were)
        else:
            assert a["bbox"] == plain[i]["bbox"]


@pytest.mark.parametrize("nb,gh,gw,H,hd,hdp", [(1, 64, 64, 2, 80, 128), (1, 64, 64, 3, 64, 64), (3, 14, 14, 2, 64, 64),
                                                  (2, 14, 14, 2, 80, 128), (1, 10, 10, 1, 64, 64), (2, 5, 13, 2, 48, 64),
                                                  (3, 16, 16, 2, 80, 128), (1, 9, 9, 1, 96, 128), (2, 8, 8, 1, 80, 128), (1, 13, 5, 3, 80, 128),
                                                  (1, 64, 64, 2, 80, 80), (2, 14, 14, 3, 80, 80), (3, 16, 16, 2, 80, 80), (1, 9, 9, 1, 80, 80),
                                                  (2, 8, 8, 1, 80, 80), (1, 13, 5, 3, 80, 80), (1, 20, 20, 2, 80, 80),
                                                  (1, 4, 40, 2, 80, 80), (1, 2, 100, 1, 64, 64), (1, 100, 1, 1, 64, 64), (2, 7, 25, 2, 64, 64)])
def test_fused_attention_with_relpos_bias(pkg, nb, gh, gw, H, hd, hdp):
    """marship_sam_attn against softmax(scale q.k + rel_h + rel_w) v in f64 on the same bf16 operands: global
    blocks of a 1024-pixel image (one key row per tile), windows (every key resident; the bias as a matrix product when
    a query's two bias rows have at most 32 entries together, looked up per score otherwise), ragged last tiles, padded
    head dims, grids one token wide."""
    ops = pkg.ops
    T = gh * gw
    g = torch.Generator().manual_seed(nb * 1000 + T + hdp)
    qkv = torch.zeros(nb * T, 3 * H * hdp, dtype=torch.bfloat16)
    val = torch.randn(nb * T, 3, H, hd, generator=g)
    val[:, 0] *= 2.0                                                   # scores a few units wide
    qkv.view(nb * T, 3, H, hdp)[..., :hd] = val.to(torch.bfloat16)
    rel_h = torch.randn(nb * H, T, gh, generator=g) * 1.5
    rel_w = torch.randn(nb * H, T, gw, generator=g) * 1.5
    scale = hd ** -0.5
    q, k, v = (qkv.view(nb, T, 3, H, hdp)[:, :, i].permute(0, 2, 1, 3).double() for i in range(3))      # [nb, H, T, hdp]
    kh, kw = torch.arange(T) // gw, torch.arange(T) % gw
    bias = rel_h.double().view(nb, H, T, gh)[..., kh] + rel_w.double().view(nb, H, T, gw)[..., kw]
    P = torch.softmax(scale * q @ k.transpose(-1, -2) + bias, dim=-1)
    ref = (P @ v).permute(0, 2, 1, 3).reshape(nb * T, H * hdp)
    out = torch.full((nb * T, H * hdp), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.sam_attn(qkv.cuda(), out, rel_h.cuda(), rel_w.cuda(), nb, T, H, hdp, gh, gw, scale)
    assert bool(torch.isfinite(out.float()).all())
    err = (out.double().cpu() - ref).abs()
    # the probabilities are rounded to bf16 before P.V and the result once more
    assert float(err.max()) < 0.02 * float(ref.abs().max()), float(err.max())
    if hdp > hd:
        assert float(out.view(nb * T, H, hdp)[..., hd:].abs().max()) == 0.0


def test_row_maps_of_relpos_and_attention_equal_materialised_windows(pkg):
    """The relative-position terms and the attention of windows read through the row map of window_partition (padding
    tokens = one row of padding values) and written back in image order == the same kernels on materialised windows
    followed by gather_rows, bit for bit; rows of the output that no entry names keep their contents."""
    ops = pkg.ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(12)
    G, ws, H, hd = 20, 14, 2, 80
    nw = -(-G // ws)
    part = torch.full((nw * nw * ws * ws,), -1, dtype=torch.int32)
    unpart = torch.zeros(G * G, dtype=torch.int32)
    for wy in range(nw):
        for wx in range(nw):
            for iy in range(ws):
                for ix in range(ws):
                    y, x = wy * ws + iy, wx * ws + ix
                    if y < G and x < G:
                        r = ((wy * nw + wx) * ws + iy) * ws + ix
                        part[r] = y * G + x
                        unpart[y * G + x] = r
    part, unpart = part.to(dev), unpart.to(dev)
    T, nb = ws * ws, nw * nw
    img = torch.randn(G * G, 3 * H * hd, generator=g).to(torch.bfloat16).to(dev)          # qkv of the image's rows
    pad = torch.randn(3 * H * hd, generator=g).to(torch.bfloat16).to(dev)
    win = torch.empty(nb * T, 3 * H * hd, dtype=torch.bfloat16, device=dev)              # the windows, materialised
    ops.gather_rows(img, part, win)
    win[part < 0] = pad
    Rh, Rw = torch.randn(ws, ws, hd, generator=g).to(dev), torch.randn(ws, ws, hd, generator=g).to(dev)
    rel = [[torch.empty(nb * H * T, ws, device=dev) for _ in range(2)] for _ in range(2)]
    ops.relpos_bias(win, nb, T, H, hd, hd, ws, Rh, Rw, rel[0][0], rel[0][1])
    ops.relpos_bias(img, nb, T, H, hd, hd, ws, Rh, Rw, rel[1][0], rel[1][1], in_rows=part, pad_row=pad)
    assert torch.equal(rel[0][0], rel[1][0]) and torch.equal(rel[0][1], rel[1][1])
    ao = torch.empty(nb * T, H * hd, dtype=torch.bfloat16, device=dev)
    ops.sam_attn(win, ao, rel[0][0], rel[0][1], nb, T, H, hd, ws, ws, hd ** -0.5)
    want = torch.empty(G * G, H * hd, dtype=torch.bfloat16, device=dev)
    ops.gather_rows(ao, unpart, want)
    got = torch.full_like(want, 7.0)
    ops.sam_attn(img, got, rel[0][0], rel[0][1], nb, T, H, hd, ws, ws, hd ** -0.5, in_rows=part, pad_row=pad, out_rows=part)
    assert torch.equal(got, want)
    only_out = torch.full_like(want, 7.0)
    ops.sam_attn(win, only_out, rel[0][0], rel[0][1], nb, T, H, hd, ws, ws, hd ** -0.5, out_rows=part)
    assert torch.equal(only_out, want)


def test_encoder_without_materialised_windows_equals_the_gathers(pkg, monkeypatch):
    """14 x 14 windows on a 20 x 20 token grid (border windows padded): the encoder that reads the windows through row
    maps == the encoder that gathers them, bit for bit, on two images through the same work buffers."""
    sa = pkg.segment_anything
    sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["win14"], seed=5)
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, size=(200, 320, 3), dtype=np.uint8), rng.integers(0, 256, size=(320, 240, 3), dtype=np.uint8)]
    feats = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MARS_SAM_FUSED_PARTITION", flag)
        sam = sa.sam_model_registry["win14"](state_dict=sd, device="cuda", dtype=torch.bfloat16)
        assert (sam.image_encoder.blocks[0].qkv_pad is not None) and sam.image_encoder.fused_partition == (flag == "1")
        pred = sa.SamPredictor(sam)
        out = []
        for img in imgs:
            pred.set_image(img)
            out.append(pred.features.float().clone())
        feats.append(out)
    assert torch.equal(feats[0][0], feats[1][0]) and torch.equal(feats[0][1], feats[1][1])
    assert not torch.equal(feats[0][0], feats[0][1])


@pytest.mark.parametrize("arch", ["tiny", "tiny80"])
def test_encoder_fused_attention_agrees_with_three_launch_form(pkg, arch, monkeypatch):
    """bf16 encoder with the fused attention against the same encoder with scores, softmax and P.V as three
    launches: both are one bf16 rounding per operand away from the f32 encoder."""
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS[arch]
    sd = pkg.synth.sam_state_dict(cfg, seed=5)
    img = GOLD[f"sam_{arch}_image"]
    ref = torch.as_tensor(GOLD[f"sam_{arch}_features"]).cuda()
    feats = {}
    for fused in (True, False):
        monkeypatch.setenv("MARS_FLASH", "1" if fused else "0")       # (80-dim heads are stored unpadded for the fused kernel only)
        sam = sa.sam_model_registry[arch](state_dict=sd, device="cuda", dtype=torch.bfloat16)
        assert sam.image_encoder.fused_attention == fused
        assert sam.image_encoder.hdp == ((80 if fused else 128) if arch == "tiny80" else 64)
        pred = sa.SamPredictor(sam)
        pred.set_image(img)
        feats[fused] = pred.features.float().clone()
    e_fused, e_three = maxdiff(feats[True], ref), maxdiff(feats[False], ref)
    assert e_fused < 1.5 * e_three + 1e-3 * float(ref.abs().max()), (e_fused, e_three)


def test_fused_upscaling_agrees_with_stagewise_form(pkg, monkeypatch):
    """At the shipped decoder width in bf16 the upscaling runs as two fused kernels: same logits (to bf16 rounding of
    the intermediate they share) and the same IoU predictions as the product-per-stage form, on the real token geometry."""
    sa = pkg.segment_anything
    sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["geom"], seed=3)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    pts = np.stack([rng.uniform(0, 400, 6), rng.uniform(0, 300, 6)], -1)[:, None, :]
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MARS_SAM_FUSED_UPSCALE", flag)
        sam = sa.sam_model_registry["geom"](state_dict=sd, device="cuda", dtype=torch.bfloat16)
        assert sam.mask_decoder.fused_up == (flag == "1")
        pred = sa.SamPredictor(sam)
        pred.set_image(img)
        coords = torch.as_tensor(pred.transform.apply_coords(pts, img.shape[:2]), dtype=torch.float32, device="cuda")
        masks, iou, low = pred.predict_torch(coords, torch.ones(6, 1, dtype=torch.int32, device="cuda"), multimask_output=True, return_logits=True)
        res.append((masks.float().clone(), iou.float().clone(), low.float().clone()))
    (m1, i1, l1), (m0, i0, l0) = res
    assert torch.equal(i1, i0)
    scale = float(l0.abs().max())
    assert float((l1 - l0).abs().max()) <= 2e-3 * scale
    assert float(((m1 > 0) != (m0 > 0)).float().mean()) < 1e-3


@pytest.mark.parametrize("columns", ["1", "0", "flipped"])
@pytest.mark.parametrize("L,S,inp,orig", [(256, 1024, (768, 1024), (480, 640)), (32, 128, (128, 96), (333, 250)), (64, 256, (171, 256), (100, 150)),
                                           (256, 1024, (1024, 683), (1500, 1000)), (30, 120, (120, 90), (77, 55))])
def test_on_the_fly_postprocess_equals_the_stored_form(pkg, monkeypatch, L, S, inp, orig, columns):
    """Statistics and 0 / 1 masks evaluated from the low-resolution logits inside the kernels == postprocess_masks
    (two stored passes) followed by mask_stats / threshold_masks, bit for bit; a strided view of four planes a prompt.
    The statistics in both forms (a thread walking its columns down the plane where the shape allows it / a band of
    output rows at a time), and on a y table in reverse order (output rows bottom up: the walk starts over at every row)."""
    ops, rs = pkg.ops, pkg.resample
    dev = torch.device("cuda")
    oh, ow = orig
    monkeypatch.setenv("MARS_RESAMPLE_COLUMNS", "0" if columns == "0" else "1")
    th_h, tw_h = rs.sam_postprocess_table(L, S, inp[0], oh), rs.sam_postprocess_table(L, S, inp[1], ow)
    if columns == "flipped":
        th_h = tuple(np.ascontiguousarray(a[::-1]) for a in th_h)
    up = lambda t: tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in t)
    th, tw = up(th_h), up(tw_h)
    band, rows = rs.band_plan(th_h, L, ow)
    B, nm = 5, 3
    g = torch.Generator().manual_seed(L + oh)
    base = (torch.randn(B, 4, L // 8, L // 8, generator=g) * 6).to(dev)
    low4 = F.interpolate(base, size=(L, L), mode="bilinear") + torch.randn(B, 4, L, L, generator=g).to(dev) * 0.3
    low4[1, 2] = -5.0                                                      # an empty mask
    low = low4[:, 1:]                                                      # a view: planes 1..3 of every prompt
    tmp = torch.empty(B * nm, L, ow, device=dev); full = torch.empty(B * nm, oh, ow, device=dev)
    lc = low.contiguous()
    ops.resample_axis(lc, B * nm, L, L, 1, tw, tmp)
    ops.resample_axis(tmp, B * nm, L, ow, 0, th, full)
    want = torch.empty(B * nm, 6, dtype=torch.int32, device=dev)
    ops.mask_stats(full, 0.0, 1.0, want)
    got = torch.full_like(want, -7)
    ops.resample_mask_stats(low, tw, th, band, rows, 0.0, 1.0, got)
    assert torch.equal(got, want)
    got2, bits_all = torch.full_like(want, -7), torch.full((B * nm, oh, ow), 9.0, device=dev)
    ops.resample_mask_stats(low, tw, th, band, rows, 0.0, 1.0, got2, bits=bits_all)          # statistics and masks in one pass
    assert torch.equal(got2, want) and torch.equal(bits_all, (full > 0.0).float())
    assert int(want[4, 0]) == 0 and want[4, 2:].tolist() == [0, 0, 0, 0]   # (plane 4 = prompt 1, mask 2 of the view)
    sel = torch.tensor([7, 0, 14, 4, 3], dtype=torch.int32, device=dev)
    bits = torch.full((5, oh, ow), 9.0, device=dev)
    ops.resample_threshold(low, sel, tw, th, band, rows, 0.0, bits)
    assert torch.equal(bits, (full[sel.long()] > 0.0).float())
    allb = torch.empty(B * nm, oh, ow, device=dev)
    ops.resample_threshold(low, None, tw, th, band, rows, 0.0, allb)
    assert torch.equal(allb, (full > 0.0).float())


def test_generator_on_the_fly_postprocess_gives_the_same_records(pkg, monkeypatch):
    sa = pkg.segment_anything
    sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["tiny"], seed=5)
    img = GOLD["sam_tiny_image"]
    recs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MARS_SAM_FUSED_POST", flag)
        sam = sa.sam_model_registry["tiny"](state_dict=sd, device="cuda")
        amg = sa.SamAutomaticMaskGenerator(sam, points_per_side=6, points_per_batch=7, pred_iou_thresh=-10.0, stability_score_thresh=0.02)
        calls = []
        orig = pkg.ops.resample_mask_stats
        monkeypatch.setattr(pkg.ops, "resample_mask_stats", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        recs.append(amg.generate(img))
        monkeypatch.setattr(pkg.ops, "resample_mask_stats", orig)
        assert bool(calls) == (flag == "1")
    a, b = recs
    assert len(a) == len(b) and len(a) > 0
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["segmentation"], rb["segmentation"]) and ra["bbox"] == rb["bbox"] and ra["area"] == rb["area"]
        assert ra["predicted_iou"] == rb["predicted_iou"] and ra["stability_score"] == rb["stability_score"]


def test_fused_image_to_token_block_agrees_with_separate_launches(pkg, monkeypatch):
    """bf16 decoder at the shipped width: image -> token attention, norm4 and the next layer's operands as one kernel
    against projection + attention + projection + normalisation pass, on the real token geometry."""
    sa = pkg.segment_anything
    sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["geom"], seed=3)
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    pts = np.stack([rng.uniform(0, 400, 6), rng.uniform(0, 300, 6)], -1)[:, None, :]
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MARS_SAM_FUSED_I2T", flag)
        sam = sa.sam_model_registry["geom"](state_dict=sd, device="cuda", dtype=torch.bfloat16)
        assert all(("i2t_fused" in L) == (flag == "1") for L in sam.mask_decoder.layers)
        pred = sa.SamPredictor(sam)
        pred.set_image(img)
        coords = torch.as_tensor(pred.transform.apply_coords(pts, img.shape[:2]), dtype=torch.float32, device="cuda")
        masks, iou, low = pred.predict_torch(coords, torch.ones(6, 1, dtype=torch.int32, device="cuda"), multimask_output=True, return_logits=True)
        res.append((masks.float().clone(), iou.float().clone(), low.float().clone()))
    (m1, i1, l1), (m0, i0, l0) = res
    scale = float(l0.abs().max())
    assert float((i1 - i0).abs().max()) <= 2e-2
    assert float((l1 - l0).abs().max()) <= 3e-2 * scale
    assert float(((m1 > 0) != (m0 > 0)).float().mean()) < 5e-3


def test_fused_token_to_image_block_agrees_with_separate_launches(pkg, monkeypatch):
    """bf16 decoder at the shipped width: token -> image attention with the key / value projections inside the kernel against
    projection + attention launches (the switch that sends small batches to the separate launches is lowered for the test)."""
    sa = pkg.segment_anything
    sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["geom"], seed=3)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    pts = np.stack([rng.uniform(0, 400, 9), rng.uniform(0, 300, 9)], -1)[:, None, :]
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MARS_SAM_FUSED_T2I", flag)
        monkeypatch.setenv("MARS_SAM_FUSED_T2I_MIN", "1")
        sam = sa.sam_model_registry["geom"](state_dict=sd, device="cuda", dtype=torch.bfloat16)
        assert hasattr(sam.mask_decoder.final, "wk_runs") == (flag == "1")
        calls = []
        orig = pkg.ops.sam_t2i_block
        monkeypatch.setattr(pkg.ops, "sam_t2i_block", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        pred = sa.SamPredictor(sam)
        pred.set_image(img)
        coords = torch.as_tensor(pred.transform.apply_coords(pts, img.shape[:2]), dtype=torch.float32, device="cuda")
        masks, iou, low = pred.predict_torch(coords, torch.ones(9, 1, dtype=torch.int32, device="cuda"), multimask_output=True, return_logits=True)
        monkeypatch.setattr(pkg.ops, "sam_t2i_block", orig)
        assert len(calls) == (len(sam.mask_decoder.layers) + 1 if flag == "1" else 0)  # every layer and the final attention
        res.append((masks.float().clone(), iou.float().clone(), low.float().clone()))
    (m1, i1, l1), (m0, i0, l0) = res
    scale = float(l0.abs().max())
    assert float((i1 - i0).abs().max()) <= 2e-2
    assert float((l1 - l0).abs().max()) <= 3e-2 * scale
    assert float(((m1 > 0) != (m0 > 0)).float().mean()) < 5e-3


def test_fused_bf16_decoder_at_the_shipped_width_against_f32(pkg, monkeypatch):
    """The whole bf16 mask decoder with every fused kernel on (256 channels, 64 x 64 token grid, the lane-order stream
    included) against the f32 decoder of the same model: logits within a few bf16 roundings, masks almost everywhere equal
    (measured: relative logit error 0.009, IoU predictions 0.0065 apart, 0.3 % of the mask pixels differ)."""
    sa = pkg.segment_anything
    sd = pkg.synth.sam_state_dict(pkg.synth.SAM_ARCHS["geom"], seed=3)
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    pts = np.stack([rng.uniform(0, 400, 12), rng.uniform(0, 300, 12)], -1)[:, None, :]
    monkeypatch.setenv("MARS_SAM_FUSED_T2I_MIN", "1")
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        sam = sa.sam_model_registry["geom"](state_dict=sd, device="cuda", dtype=dt)
        pred = sa.SamPredictor(sam)
        pred.set_image(img)
        feats32 = res["feats"] if dt == torch.bfloat16 else pred.features_tokens.clone()
        res["feats"] = feats32
        pred.features_tokens = feats32.clone()                      # same image embedding for both decoders: this test is about the decoder
        coords = torch.as_tensor(pred.transform.apply_coords(pts, img.shape[:2]), dtype=torch.float32, device="cuda")
        masks, iou, low = pred.predict_torch(coords, torch.ones(12, 1, dtype=torch.int32, device="cuda"), multimask_output=True, return_logits=True)
        res[dt] = (masks.float().clone(), iou.float().clone(), low.float().clone())
    (m32, i32, l32), (m16, i16, l16) = res[torch.float32], res[torch.bfloat16]
    rel = float((l16 - l32).norm() / l32.norm())
    print("bf16 vs f32 decoder: relative logit error", rel, "iou", float((i16 - i32).abs().max()), "mask bits", float(((m16 > 0) != (m32 > 0)).float().mean()))
    assert rel < 0.03, rel
    assert float((i16 - i32).abs().max()) < 0.05
    assert float(((m16 > 0) != (m32 > 0)).float().mean()) < 0.01
