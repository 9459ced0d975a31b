"""The oracle restatement against the vectors the REFERENCE modules produced
(tests/golden/reference_vectors.npz, written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_package, maxdiff
from oracle import pipeline as opipe
from oracle import vit as ovit

pkg = load_package(require_native=False)
synth = pkg.synth
G = np.load(os.path.join(GOLDEN, "reference_vectors.npz"))


def close(a, b, tol=1e-5):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_dinov2_forward_and_attention():
    for arch, img in (("vit_tiny2", 70), ("vit_tiny3", 84)):
        cfg = synth.dino_cfg(arch, img_size=img)
        sd = synth.dinov2_state_dict(cfg, seed=1)
        x = synth.randn(7, f"dino_in_{arch}", (2, 3, img, img), 1.0)
        kw = dict(depth=cfg.depth, heads=cfg.num_heads, patch=14, num_regs=cfg.num_regs, ffn="mlp")
        close(ovit.dino_forward(sd, x, **kw), G[f"dino_{arch}_x_prenorm"])
        _, attn = ovit.dino_forward(sd, x[:1], want_attn=True, **kw)
        close(attn[-1], G[f"dino_{arch}_attn_last"])
        close(opipe.mean_attention(attn, cfg.depth, cfg.num_regs), G[f"dino_{arch}_mean_attn"])


def test_dinov2_interpolated_positional_table():
    cfg = synth.dino_cfg("vit_tiny2", img_size=70)
    sd = synth.dinov2_state_dict(cfg, seed=1)
    x = synth.randn(7, "dino_in_interp", (1, 3, 98, 84), 1.0)
    got = ovit.dino_forward(sd, x, depth=cfg.depth, heads=cfg.num_heads, patch=14, num_regs=cfg.num_regs, ffn="mlp")
    assert float((got - torch.from_numpy(G["dino_vit_tiny2_interp_x_prenorm"])).abs().max()) < 1e-5


def test_dinov2_swiglu():
    cfg = synth.DinoCfg(embed_dim=128, depth=2, num_heads=2, img_size=70, ffn="swiglu")
    sd = synth.dinov2_state_dict(cfg, seed=2)
    x = synth.randn(7, "dino_in_swiglu", (1, 3, 70, 70), 1.0)
    close(ovit.dino_forward(sd, x, depth=2, heads=2, patch=14, num_regs=4, ffn="swiglu"), G["dino_swiglu_x_prenorm"])


def test_clip_image_text_gradcam():
    cfg = synth.CLIP_ARCHS["tiny/16"]
    sd = synth.clip_state_dict(cfg, seed=3)
    x = synth.randn(7, "clip_in", (1, 3, 80, 80), 1.0)
    feats, attn = ovit.clip_encode_image(sd, x, layers=cfg.vision_layers, heads=cfg.vision_heads, patch=cfg.vision_patch_size)
    close(feats, G["clip_feats"])
    toks = synth.tokenize(["a photo of object.", "a photo without object.", "a object."])
    close(ovit.clip_encode_text(sd, toks, layers=cfg.transformer_layers, heads=cfg.transformer_heads), G["clip_text"])
    text2 = opipe.vta_text_feats(sd, dict(text_layers=cfg.transformer_layers, text_heads=cfg.transformer_heads), toks[0:1], toks[1:2])
    cam, attn_last, logits = ovit.gradcam(sd, feats, text2, layers=cfg.vision_layers, heads=cfg.vision_heads)
    close(cam, G["clip_cam"], tol=2e-5)
    close(logits, G["clip_logits"])
    close(torch.stack([a[0] for a in attn] + [attn_last[0]]).mean(0), G["clip_attn_mean"])


def test_alphaclip_visual_and_text():
    cfg = synth.CLIP_ARCHS["tiny/14"]
    sd = synth.clip_state_dict(cfg, seed=4, alpha=True)
    R = cfg.image_resolution
    img = synth.randn(7, "aclip_img", (1, 3, R, R), 1.0).repeat(3, 1, 1, 1)
    alpha = synth.randn(7, "aclip_alpha", (3, 1, R, R), 1.5)
    close(ovit.alphaclip_visual(sd, img, alpha, layers=cfg.vision_layers, heads=cfg.vision_heads, patch=cfg.vision_patch_size),
          G["aclip_visual"])
    close(ovit.clip_encode_text(sd, synth.tokenize(["a object."]), layers=cfg.transformer_layers, heads=cfg.transformer_heads),
          G["aclip_text"])


def test_evaluator_areas():
    pred, gt, ign = torch.from_numpy(G["eval_pred"]), torch.from_numpy(G["eval_gt"]), torch.from_numpy(G["eval_ignore"])
    i, u = opipe.classify_prediction(pred[0], gt[0])
    close(i, G["eval_inter"][:, 0]); close(u, G["eval_union"][:, 0])
    i, u = opipe.classify_prediction(pred[0], gt[0], ign[0])
    close(i, G["eval_inter_ign"][:, 0]); close(u, G["eval_union_ign"][:, 0])


def test_prompt_tokens_match_reference_tokenizer():
    with open(os.path.join(GOLDEN, "prompt_tokens.json")) as f:
        table = json.load(f)
    for text, ids in synth.PROMPT_TOKENS.items():
        assert table[text] == ids
        row = synth.tokenize([text])[0]
        assert row[: len(ids)].tolist() == ids and int(row[len(ids):].abs().sum()) == 0
        assert int(row.argmax()) == len(ids) - 1   # EOT is the highest id: encode_text pools there


@pytest.mark.parametrize("arch", ["tiny", "tiny80"])
def test_sam_encoder_decoder_postprocess(arch):
    """oracle/sam.py against the outputs of the reference's segment_anything.modeling modules."""
    from oracle import sam as osam
    cfg = synth.SAM_ARCHS[arch]
    sd = synth.sam_state_dict(cfg, seed=5)
    kw = cfg.oracle_kw()
    img = G[f"sam_{arch}_image"]
    state = osam.set_image(sd, kw, img)
    assert maxdiff(state["features"], G[f"sam_{arch}_features"]) < 2e-5
    pts = torch.as_tensor(G[f"sam_{arch}_points"])
    masks, iou, low = osam.predict_torch(sd, kw, state, pts[:, None, :], torch.ones(len(pts), 1))
    assert maxdiff(low, G[f"sam_{arch}_low_res"]) < 5e-5
    assert maxdiff(iou, G[f"sam_{arch}_iou"]) < 5e-5
    assert maxdiff(masks, G[f"sam_{arch}_masks"]) < 5e-5
    # box prompts alone (corner tokens) and two points plus a box through the decoder
    boxes = torch.as_tensor(G[f"sam_{arch}_boxes"])
    assert maxdiff(osam.embed_boxes(sd, boxes, (cfg.img_size, cfg.img_size)), G[f"sam_{arch}_box_sparse"]) < 2e-5
    _, iou_b, low_b = osam.predict_torch(sd, kw, state, torch.as_tensor(G[f"sam_{arch}_box_points"]), torch.tensor([[1, 0]] * 3), boxes=boxes)
    assert maxdiff(low_b, G[f"sam_{arch}_box_low_res"]) < 5e-5 and maxdiff(iou_b, G[f"sam_{arch}_box_iou"]) < 5e-5
    # a mask prompt: the first prediction's low-resolution logits fed back with two points
    mask_in = torch.as_tensor(G[f"sam_{arch}_low_res"])[:3, :1]
    assert maxdiff(osam.embed_masks(sd, mask_in), G[f"sam_{arch}_mask_dense"]) < 2e-5
    _, iou_m, low_m = osam.predict_torch(sd, kw, state, torch.as_tensor(G[f"sam_{arch}_box_points"]), torch.tensor([[1, 0]] * 3), mask_input=mask_in)
    assert maxdiff(low_m, G[f"sam_{arch}_mask_low_res"]) < 5e-5 and maxdiff(iou_m, G[f"sam_{arch}_mask_iou"]) < 5e-5


def test_sam_generator_structure():
    """End-to-end generator of the oracle: record fields, NMS keeps non-overlapping boxes, thresholds bite."""
    from oracle import sam as osam
    cfg = synth.SAM_ARCHS["tiny"]
    sd = synth.sam_state_dict(cfg, seed=5)
    img = G["sam_tiny_image"]
    recs = osam.generate(sd, cfg.oracle_kw(), img, points_per_side=4, points_per_batch=5, pred_iou_thresh=-10.0,
                         stability_score_thresh=0.0, box_nms_thresh=0.7)
    assert len(recs) >= 1
    for r in recs:
        assert r["segmentation"].shape == img.shape[:2] and r["segmentation"].dtype == bool
        assert r["area"] == int(r["segmentation"].sum()) and len(r["bbox"]) == 4 and r["crop_box"] == [0, 0, img.shape[1], img.shape[0]]
    none = osam.generate(sd, cfg.oracle_kw(), img, points_per_side=4, points_per_batch=16, pred_iou_thresh=10.0)
    assert none == []
    # NMS restatement on a hand-made case
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 10, 10], [20, 20, 30, 30], [0, 0, 9.9, 9.9]], dtype=torch.float32)
    keep = osam.box_nms(boxes, torch.tensor([0.9, 0.95, 0.5, 0.1]), 0.7)
    assert keep.tolist() == [1, 2]
