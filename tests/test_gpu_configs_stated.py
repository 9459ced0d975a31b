"""Each benchmark configuration once, at the size and precision it is stated with.  The CPU oracle needs minutes to
hours per episode at these sizes, so the checks are size-independent properties (invariance under proposal order
and subsetting, the merged mask as the union of the kept proposals, sortedness, count identities), the exact
transport solver on a few proposals, and the meters of the configuration's own benchmark.

  configs[2]  COCO-20i 5-shot, DINOv2-L, 64 proposals, bf16 towers
  configs[3]  LVIS-92i 1-shot, DINOv2-g (SwiGLU, 40 blocks), 128 proposals, bf16 towers, meter over 1203 classes
  configs[4]  FSS-1000 1-shot, DINOv2-L, 256 proposals with fp8 operands in the AlphaCLIP attention, meter over 1000 classes
  configs[0], [1], [3] proposal stage: SAM ViT-B at full width against the CPU oracle, SAM ViT-H through properties
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tests.helpers import maxdiff  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def _build(pkg, dino="vit_large", dtype=torch.bfloat16, fp8=False, sds=None):
    MARS = importlib.import_module(pkg.__name__ + ".mars.MARS")
    synth = pkg.synth
    if sds is None:
        sds = dict(dino=synth.dinov2_state_dict(synth.dino_cfg(dino, bench.SIZE), 0),
                   clip=synth.clip_state_dict(synth.CLIP_ARCHS["ViT-B/16"], 0),
                   alphaclip=synth.clip_state_dict(synth.CLIP_ARCHS["ViT-L/14@336px"], 0, alpha=True))
    args = bench.build_args(torch.device("cuda:0"), dtype)
    args.dino_backbone = dino
    args.fp8_attention = fp8
    return MARS.build_MARS_fss(args, state_dicts=sds), sds


def _episode(pkg, seed, nshot, nprop):
    ep = pkg.synth.make_episode(seed, size=bench.SIZE, nshot=nshot, nprop=nprop, fold=0)
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ep.items()}


def _run(mars, ep, props):
    pred = mars.predict(ep["support_imgs"], ep["support_masks"], ep["query_img"], props)
    return pred, {k: v.clone() for k, v in mars.filtering_merging_component.scores().items()}


def _common_properties(pred, sc, props):
    n = props.shape[0]
    assert sc["score"].numel() == n and bool(torch.isfinite(sc["score"]).all())
    assert sorted(sc["order"].tolist()) == list(range(n))
    ordered = sc["score"][sc["order"].long()]
    assert bool((ordered[:-1] >= ordered[1:]).all())
    assert float(sc["emd"].min()) >= 0.0 and float(sc["emd"].max()) <= 1.0 + 1e-9
    assert float(sc["aclip"].abs().max()) <= 1.0 + 1e-3
    kept = props[sc["keep"].bool().cuda()]
    assert kept.shape[0] >= 1 and torch.equal(pred, (kept.sum(0) > 0).float())


def _meter(pkg, benchmark, class_ids, pred, ep, class_id):
    """One update of the benchmark's meter and the identities its counts must satisfy."""
    Evaluator = importlib.import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
    AverageMeter = importlib.import_module(pkg.__name__ + ".mars.utils.logger").AverageMeter
    Evaluator.initialize()
    meter = AverageMeter(types.SimpleNamespace(benchmark=benchmark, class_ids=class_ids), device="cuda:0")
    ai, au = Evaluator.classify_prediction(pred[None].clone(), {"query_mask": ep["query_mask"]})
    meter.update(ai, au, torch.tensor([class_id]))
    gt = ep["query_mask"][0]
    inter, union = float(((pred > 0) & (gt > 0)).sum()), float(((pred > 0) | (gt > 0)).sum())
    assert float(meter.intersection_buf[1, class_id]) == inter and float(meter.union_buf[1, class_id]) == union
    assert float(meter.intersection_buf.sum()) + float(meter.union_buf.sum()) == 2.0 * pred.numel()
    miou, fb, _ = meter.compute_iou()
    assert abs(float(miou) - 100.0 * inter / max(union, 1.0) / len(class_ids)) < 1e-3       # one class filled, the rest 0
    assert meter.intersection_buf.shape == (2, meter.nclass)
    return meter


def test_config2_five_shot_64_proposals_bf16(pkg, cuda):
    """configs[2]: five supports (similarity [6845, 1369]), 64 proposals, bf16 towers; EMD of three proposals
    against the exact CPU solver on the same similarity rows."""
    mars, _ = _build(pkg)
    ep = _episode(pkg, 2001, 5, 64)
    pred, sc = _run(mars, ep, ep["mask_proposals"])
    _common_properties(pred, sc, ep["mask_proposals"])
    vva = mars.visual_visual_alignment_component
    S = vva.similarity_matrix
    G = vva.model_embedding_spatial_dimensions
    assert tuple(S.shape) == (5 * G * G, G * G)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1))
    pred_p, sc_p = _run(mars, ep, ep["mask_proposals"][perm.cuda()].contiguous())
    assert torch.equal(pred, pred_p)
    for k in ("emd", "aclip", "cnt", "svv", "svt", "score"):
        assert torch.equal(sc[k][perm], sc_p[k]), k
    from oracle.emd import emd2_uniform
    C = ((1 - S) / 2).double().numpy()
    sb = torch.empty(5 * G * G, dtype=torch.uint8, device="cuda")
    pkg.ops.pool_masks(ep["support_masks"][0].contiguous().float(), G, sb)
    rows = sb.cpu().bool().numpy()
    small = torch.argsort(sc["cnt"])[:3].tolist()                      # the three smallest proposals: the solver's time grows with columns
    bits = torch.empty(3, G * G, dtype=torch.uint8, device="cuda")
    pkg.ops.pool_masks(ep["mask_proposals"][small].contiguous(), G, bits)
    for j, i in enumerate(small):
        cols = bits[j].cpu().bool().numpy()
        want = 1.0 - emd2_uniform(C[rows][:, cols])
        assert abs(float(sc["emd"][i]) - want) < 1e-6, (i, float(sc["emd"][i]), want)
    _meter(pkg, "coco", [4 * i for i in range(20)], pred, ep, int(ep["class_id"]))


def test_config3_lvis_dinov2_giant_128_proposals_bf16(pkg, cuda):
    """configs[3]: DINOv2-g/14 reg4 (SwiGLU feed-forward, 40 blocks, 24 heads) in bf16, 128 proposals, and the LVIS
    meter (1203 classes)."""
    torch.set_num_threads(16)
    mars, _ = _build(pkg, dino="vit_giant2")
    ep = _episode(pkg, 3001, 1, 128)
    pred, sc = _run(mars, ep, ep["mask_proposals"])
    _common_properties(pred, sc, ep["mask_proposals"])
    idx = torch.arange(40, 104)
    _, sub = _run(mars, ep, ep["mask_proposals"][idx.cuda()].contiguous())
    for k in ("emd", "aclip", "cnt", "svv", "svt"):
        assert torch.equal(sc[k][idx], sub[k]), k                      # a proposal's quantities do not depend on its batch
    assert mars.visual_visual_alignment_component.model.embed_dim == 1536
    m = _meter(pkg, "lvis", list(range(0, 1203, 10)), pred, ep, 1200)
    assert m.nclass == 1203


def test_config4_fss_256_proposals_with_fp8_attention(pkg, cuda):
    """configs[4]: 256 proposals AND the fp8 attention path in one run (the AlphaCLIP tower's q, k, v and
    probabilities as e4m3), with the FSS-1000 meter (1000 classes).  Against the bf16 run of the same episode:
    the feature directions and the ranking stay close, everything that does not pass through the attention is
    bitwise unchanged."""
    from scipy.stats import kendalltau
    mars, sds = _build(pkg)
    ep = _episode(pkg, 4001, 1, 256)
    pred16, sc16 = _run(mars, ep, ep["mask_proposals"])
    mars8, _ = _build(pkg, fp8=True, sds=sds)
    assert mars8.filtering_merging_component.alpha_clip_model.vtower.fp8_attention
    pred8, sc8 = _run(mars8, ep, ep["mask_proposals"])
    _common_properties(pred8, sc8, ep["mask_proposals"])
    cos = torch.nn.functional.cosine_similarity(sc16["feats"].float(), sc8["feats"].float(), dim=1)
    tau = kendalltau(sc16["score"].double().numpy(), sc8["score"].double().numpy()).statistic
    print(f"256 proposals, fp8 attention vs bf16: min feature cosine {float(cos.min()):.4f}, Kendall tau {tau:.3f}")
    assert float(cos.min()) > 0.98 and tau > 0.8
    for k in ("emd", "cnt", "svv", "svt"):
        assert torch.equal(sc8[k], sc16[k]), k
    idx = torch.arange(96, 160)                                        # straddles the AlphaCLIP chunk boundary at 128
    _, sub = _run(mars8, ep, ep["mask_proposals"][idx.cuda()].contiguous())
    for k in ("emd", "aclip", "cnt"):
        assert torch.equal(sc8[k][idx], sub[k]), k
    m = _meter(pkg, "fss", list(range(760, 1000)), pred8, ep, 999)
    assert m.nclass == 1000


def test_sam_vit_b_full_width_against_oracle(pkg, cuda):
    """SAM ViT-B (768 wide, 12 blocks, global attention at 2/5/8/11, 1024-pixel input) in fp32 against the CPU
    restatement: image embedding, then one batch of point prompts through prompt encoder and mask decoder."""
    from oracle import sam as osam
    torch.set_num_threads(16)
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS["vit_b"]
    sd = pkg.synth.sam_state_dict(cfg, seed=9)
    sam = sa.sam_model_registry["vit_b"](state_dict=sd, device="cuda")
    img = (pkg.synth.rand(4, "vitb_img", (518, 518, 3)) * 255).to(torch.uint8).numpy()
    pred = sa.SamPredictor(sam)
    pred.set_image(img)
    state = osam.set_image(sd, cfg.oracle_kw(), img)
    scale = float(state["features"].abs().max())
    d = maxdiff(pred.features, state["features"])
    print(f"SAM ViT-B image embedding: max |gpu - oracle| = {d:.2e} (|oracle| max {scale:.2e})")
    assert d < 2e-4 * max(1.0, scale)
    pts = torch.tensor([[[100.0, 200.0]], [[259.0, 259.0]], [[500.5, 50.25]], [[17.0, 480.0]]])
    lab = torch.ones(4, 1)
    masks, iou, low = pred.predict_torch(pts.cuda(), lab.int().cuda(), return_logits=True)
    rm, ri, rl = osam.predict_torch(sd, cfg.oracle_kw(), state, pts, lab)
    s = max(1.0, float(rl.abs().max()))
    assert maxdiff(low, rl) < 5e-4 * s and maxdiff(iou, ri) < 5e-4 and maxdiff(masks, rm) < 5e-4 * s


def test_sam_vit_h_full_width_properties(pkg, cuda):
    """SAM ViT-H (1280 wide, 32 blocks) in bf16, the proposal generator of configs[1] and [3]: the records are
    consistent (area = mask pixels, box bounds the mask, scores above their thresholds, suppression leaves no pair
    above the IoU threshold), the bf16 embedding agrees with an fp32 run, and the proposals feed the ranking."""
    sa = pkg.segment_anything
    cfg = pkg.synth.SAM_ARCHS["vit_h"]
    sd = pkg.synth.sam_state_dict(cfg, seed=9)
    img = (pkg.synth.rand(5, "vith_img", (518, 518, 3)) * 255).to(torch.uint8).numpy()
    sam16 = sa.sam_model_registry["vit_h"](state_dict=sd, device="cuda", dtype=torch.bfloat16)
    p16 = sa.SamPredictor(sam16)
    p16.set_image(img)
    f16 = p16.features.float().clone()
    assert bool(torch.isfinite(f16).all())
    sam32 = sa.sam_model_registry["vit_h"](state_dict=sd, device="cuda", dtype=torch.float32)
    p32 = sa.SamPredictor(sam32)
    p32.set_image(img)
    f32 = p32.features.float()
    cos = torch.nn.functional.cosine_similarity(f16.flatten(), f32.flatten(), dim=0)
    print(f"SAM ViT-H embedding, bf16 vs fp32: cosine {float(cos):.5f}")
    assert float(cos) > 0.99
    del sam32, p32
    kw = dict(points_per_side=8, points_per_batch=64, pred_iou_thresh=-10.0, stability_score_thresh=0.0, box_nms_thresh=0.7)
    recs = sa.SamAutomaticMaskGenerator(sam16, **kw).generate(img)
    assert 1 <= len(recs) <= 64 * 3
    boxes = []
    for r in recs:
        seg = r["segmentation"]
        assert seg.shape == (518, 518) and seg.dtype == bool and int(seg.sum()) == r["area"] > 0
        ys, xs = np.nonzero(seg)
        x, y, w, h = r["bbox"]
        assert x == xs.min() and y == ys.min() and x + w == xs.max() and y + h == ys.max()
        assert r["predicted_iou"] >= -10.0 and r["stability_score"] >= 0.0 and np.isfinite(r["predicted_iou"])
        boxes.append([x, y, x + w, y + h])
    b = torch.tensor(boxes, dtype=torch.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt, rb = torch.max(b[:, None, :2], b[None, :, :2]), torch.min(b[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=1e-9)
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.7 + 1e-6                              # what box NMS at 0.7 must leave
    gen = sa.ProposalGenerator(sa.SamAutomaticMaskGenerator(sam16, **kw), n_proposals=16)
    q = torch.as_tensor(img).permute(2, 0, 1)[None].float().div(255).cuda()
    props = gen.generate(None, None, q)
    assert props.shape == (16, 518, 518) and set(props.unique().tolist()) <= {0.0, 1.0}
