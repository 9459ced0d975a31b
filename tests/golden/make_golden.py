"""Generate the golden vectors under tests/golden/ by running the REFERENCE modules.

Run in the build container only (it reads /root/reference, which does not exist on
the GPU box):

    python tests/golden/make_golden.py

What is imported from the reference, unmodified:
  * dinov2.models.vision_transformer (DinoVisionTransformer)
  * clip/model.py and alpha_clip/model.py, loaded by file path because the package
    __init__ files pull torchvision/ftfy; alpha_clip needs a name-only ``loralib``
    stub (dereferenced only when lora_adapt=True)
  * pytorch_grad_cam (GradCAM) with name-only stubs for ttach / cv2 / torchvision,
    none of which is called on this path
  * mars.utils.evaluation.Evaluator
  * clip/simple_tokenizer.py (ftfy stubbed as identity; prompts are ASCII)

Seeded synthetic state dicts (package ``synth``) are loaded into those modules, the
modules are run on seeded inputs, and their outputs are written as .npz fixtures.
The script also asserts that the oracle restatement agrees with the reference to
1e-5 before writing, so a stale oracle cannot produce a fixture.
"""
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from __graft_entry__ import load_package  # noqa: E402

pkg = load_package(require_native=False)
synth = pkg.synth
from oracle import vit as ovit  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def maxdiff(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def check(tag, ref, ours, tol=1e-5):
    d = maxdiff(ref, ours)
    scale = float(torch.as_tensor(ref).abs().max())
    print(f"  {tag:34s} max|ref-oracle| = {d:.3e}  (|ref|max {scale:.3e})")
    assert d <= tol * max(1.0, scale), (tag, d)


def golden_dino(out):
    import dinov2.models.vision_transformer as vits
    for arch, img in (("vit_tiny2", 70), ("vit_tiny3", 84)):
        cfg = synth.dino_cfg(arch, img_size=img)
        sd = synth.dinov2_state_dict(cfg, seed=1)
        model = vits.DinoVisionTransformer(
            img_size=img, patch_size=14, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
            mlp_ratio=4, init_values=1e-5, ffn_layer="mlp", block_chunks=0, num_register_tokens=cfg.num_regs,
            block_fn=vits.partial(vits.Block, attn_class=vits.MemEffAttention))
        missing = model.load_state_dict(sd, strict=True)
        model.eval()
        x = synth.randn(7, f"dino_in_{arch}", (2, 3, img, img), 1.0)
        with torch.no_grad():
            ref_x = model.forward_features(x)["x_prenorm"]
            ref_attn = model.get_last_self_attention(x[:1])
        kw = dict(depth=cfg.depth, heads=cfg.num_heads, patch=14, num_regs=cfg.num_regs, ffn="mlp")
        ours_x = ovit.dino_forward(sd, x, **kw)
        _, ours_attn = ovit.dino_forward(sd, x[:1], want_attn=True, **kw)
        print(f"[dinov2 {arch} @{img}]")
        check("x_prenorm", ref_x, ours_x)
        for i, (r, o) in enumerate(zip(ref_attn, ours_attn)):
            check(f"attn[{i}]", r, o)
        ref_mean = opipe.mean_attention(list(ref_attn), cfg.depth, cfg.num_regs)
        out[f"dino_{arch}_x_prenorm"] = ref_x.numpy()
        out[f"dino_{arch}_mean_attn"] = ref_mean.numpy()
        out[f"dino_{arch}_attn_last"] = ref_attn[-1].numpy()
        if arch == "vit_tiny2":
            # an input of another size than the constructor's: interpolate_pos_encoding (:188-221), non-square
            xi = synth.randn(7, "dino_in_interp", (1, 3, 98, 84), 1.0)
            with torch.no_grad():
                ref_i = model.forward_features(xi)["x_prenorm"]
            check("x_prenorm, 98 x 84 input", ref_i, ovit.dino_forward(sd, xi, **kw))
            out["dino_vit_tiny2_interp_x_prenorm"] = ref_i.numpy()
    # SwiGLU variant (vit_giant2 family) at toy width
    cfg = synth.DinoCfg(embed_dim=128, depth=2, num_heads=2, img_size=70, ffn="swiglu")
    sd = synth.dinov2_state_dict(cfg, seed=2)
    model = vits.DinoVisionTransformer(
        img_size=70, patch_size=14, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, init_values=1e-5,
        ffn_layer="swiglufused", block_chunks=0, num_register_tokens=4,
        block_fn=vits.partial(vits.Block, attn_class=vits.MemEffAttention))
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = synth.randn(7, "dino_in_swiglu", (1, 3, 70, 70), 1.0)
    with torch.no_grad():
        ref_x = model.forward_features(x)["x_prenorm"]
    ours_x = ovit.dino_forward(sd, x, depth=2, heads=2, patch=14, num_regs=4, ffn="swiglu")
    print("[dinov2 swiglu toy]")
    check("x_prenorm", ref_x, ours_x)
    out["dino_swiglu_x_prenorm"] = ref_x.numpy()


def golden_clip(out):
    clip_model = _load_by_path("ref_clip_model", os.path.join(REF, "clip/model.py"))
    cfg = synth.CLIP_ARCHS["tiny/16"]
    sd = synth.clip_state_dict(cfg, seed=3)
    model = clip_model.CLIP(cfg.embed_dim, cfg.image_resolution, cfg.vision_layers, cfg.vision_width,
                            cfg.vision_patch_size, cfg.context_length, cfg.vocab_size, cfg.transformer_width,
                            cfg.transformer_heads, cfg.transformer_layers)
    model.load_state_dict(sd, strict=True)
    model.eval().float()
    H = 80  # 5x5 grid: exercises the positional-table upsample (native grid 4x4)
    x = synth.randn(7, "clip_in", (1, 3, H, H), 1.0)
    with torch.no_grad():
        ref_feats, ref_attn = model.encode_image(x, H, H)
    ours_feats, ours_attn = ovit.clip_encode_image(sd, x, layers=cfg.vision_layers, heads=cfg.vision_heads,
                                                   patch=cfg.vision_patch_size)
    print("[clip tiny/16 @80]")
    check("encode_image feats", ref_feats.permute(1, 0, 2), ours_feats)
    for i, (r, o) in enumerate(zip(ref_attn, ours_attn)):
        check(f"attn[{i}]", r, o)
    toks = synth.tokenize(["a photo of object.", "a photo without object.", "a object."])
    with torch.no_grad():
        ref_text = model.encode_text(toks)
    ours_text = ovit.clip_encode_text(sd, toks, layers=cfg.transformer_layers, heads=cfg.transformer_heads)
    check("encode_text", ref_text, ours_text)

    # GradCAM through the reference pytorch_grad_cam
    _stub("ttach")
    _stub("cv2", __version__="4.10.0", COLORMAP_JET=2)
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms", Compose=object, Normalize=object, ToTensor=object)
    tv.transforms = tvt
    from pytorch_grad_cam import GradCAM

    def reshape_transform(tensor, height=28, width=28):  # SoftmaxGradCAM.py:32-41 semantics
        tensor = tensor.permute(1, 0, 2)
        height = width = int((tensor.size(1) - 1) ** 0.5)
        result = tensor[:, 1:, :].reshape(tensor.size(0), height, width, tensor.size(2))
        return result.transpose(2, 3).transpose(1, 2)

    class Target:
        def __call__(self, o):
            return o[0] if o.dim() == 1 else o[:, 0]

    text2 = opipe.vta_text_feats(sd, dict(text_layers=cfg.transformer_layers, text_heads=cfg.transformer_heads),
                                 toks[0:1], toks[1:2])
    cam = GradCAM(model=model, target_layers=[model.visual.transformer.resblocks[-1].ln_1],
                  reshape_transform=reshape_transform)
    ref_cam, ref_logits, ref_attn_last = cam(input_tensor=[ref_feats, text2, H, H], targets=[Target()],
                                             target_size=None)
    ours_cam, ours_attn_last, ours_logits = ovit.gradcam(sd, ours_feats, text2, layers=cfg.vision_layers,
                                                         heads=cfg.vision_heads)
    check("gradcam cam", torch.from_numpy(ref_cam[0]), ours_cam, tol=2e-5)
    check("gradcam logits", ref_logits.detach(), ours_logits)
    check("gradcam last attn", ref_attn_last.detach(), ours_attn_last)
    out["clip_feats"] = ref_feats.permute(1, 0, 2).numpy()
    out["clip_attn_mean"] = torch.stack([a[0] for a in ref_attn] + [ref_attn_last.detach()[0]]).mean(0).numpy()
    out["clip_text"] = ref_text.numpy()
    out["clip_cam"] = ref_cam[0]
    out["clip_logits"] = ref_logits.detach().numpy()


def golden_alphaclip(out):
    _stub("loralib")
    amodel = _load_by_path("ref_alpha_clip_model", os.path.join(REF, "alpha_clip/model.py"))
    cfg = synth.CLIP_ARCHS["tiny/14"]
    sd = synth.clip_state_dict(cfg, seed=4, alpha=True)
    model = amodel.CLIP(cfg.embed_dim, cfg.image_resolution, cfg.vision_layers, cfg.vision_width,
                        cfg.vision_patch_size, cfg.context_length, cfg.vocab_size, cfg.transformer_width,
                        cfg.transformer_heads, cfg.transformer_layers)
    model.load_state_dict(sd, strict=True)
    model.eval().float()
    R = cfg.image_resolution
    img = synth.randn(7, "aclip_img", (1, 3, R, R), 1.0).repeat(3, 1, 1, 1)
    alpha = synth.randn(7, "aclip_alpha", (3, 1, R, R), 1.5)
    with torch.no_grad():
        ref = model.visual(img, alpha)
        toks = synth.tokenize(["a object."])
        ref_text = model.encode_text(toks)
    ours = ovit.alphaclip_visual(sd, img, alpha, layers=cfg.vision_layers, heads=cfg.vision_heads,
                                 patch=cfg.vision_patch_size)
    ours_text = ovit.clip_encode_text(sd, toks, layers=cfg.transformer_layers, heads=cfg.transformer_heads)
    print("[alphaclip tiny/14]")
    check("visual(image, alpha)", ref, ours)
    check("encode_text", ref_text, ours_text)
    out["aclip_visual"] = ref.numpy()
    out["aclip_text"] = ref_text.numpy()


def golden_evaluator(out):
    from mars.utils.evaluation import Evaluator
    Evaluator.initialize()
    g = torch.Generator().manual_seed(5)
    pred = (torch.rand(1, 64, 64, generator=g) > 0.5).float()
    gt = (torch.rand(1, 64, 64, generator=g) > 0.6).float()
    ai, au = Evaluator.classify_prediction(pred.clone(), {"query_mask": gt})
    oi, ou = opipe.classify_prediction(pred[0], gt[0])
    print("[evaluator]")
    check("area_inter", ai[:, 0], oi)
    check("area_union", au[:, 0], ou)
    ign = ((torch.rand(1, 64, 64, generator=g) > 0.9) & (gt == 0)).float()
    ai2, au2 = Evaluator.classify_prediction(pred.clone(), {"query_mask": gt.clone(), "query_ignore_idx": ign.clone()})
    oi2, ou2 = opipe.classify_prediction(pred[0], gt[0], ign[0])
    check("area_inter (ignore)", ai2[:, 0], oi2)
    check("area_union (ignore)", au2[:, 0], ou2)
    out["eval_pred"] = pred.numpy()
    out["eval_gt"] = gt.numpy()
    out["eval_ignore"] = ign.numpy()
    out["eval_inter"] = ai.numpy()
    out["eval_union"] = au.numpy()
    out["eval_inter_ign"] = ai2.numpy()
    out["eval_union_ign"] = au2.numpy()


def golden_sam(out):
    """Segment Anything: the reference's own modules (segment_anything.modeling, utils.amg) on seeded
    weights against oracle/sam.py."""
    from functools import partial
    # the package __init__ pulls the predictor (torchvision, absent); a bare package entry lets the
    # torch-only sub-packages `modeling` and `utils.amg` import on their own
    sa = types.ModuleType("segment_anything")
    sa.__path__ = [os.path.join(REF, "segment_anything")]
    sys.modules["segment_anything"] = sa
    from segment_anything.modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer
    from segment_anything.utils import amg
    from oracle import sam as osam
    for arch in ("tiny", "tiny80"):
        cfg = synth.SAM_ARCHS[arch]
        sd = synth.sam_state_dict(cfg, seed=5)
        G = cfg.img_size // cfg.patch_size
        model = Sam(
            image_encoder=ImageEncoderViT(
                depth=cfg.depth, embed_dim=cfg.embed_dim, img_size=cfg.img_size, mlp_ratio=4,
                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=cfg.num_heads, patch_size=cfg.patch_size,
                qkv_bias=True, use_rel_pos=True, global_attn_indexes=cfg.global_attn_indexes, window_size=cfg.window_size,
                out_chans=cfg.out_chans),
            prompt_encoder=PromptEncoder(embed_dim=cfg.out_chans, image_embedding_size=(G, G),
                                         input_image_size=(cfg.img_size, cfg.img_size), mask_in_chans=cfg.mask_in_chans),
            mask_decoder=MaskDecoder(
                num_multimask_outputs=3,
                transformer=TwoWayTransformer(depth=cfg.dec_depth, embedding_dim=cfg.out_chans, mlp_dim=cfg.dec_mlp_dim,
                                              num_heads=cfg.dec_heads),
                transformer_dim=cfg.out_chans, iou_head_depth=cfg.iou_head_depth, iou_head_hidden_dim=cfg.iou_head_hidden),
            pixel_mean=list(osam.PIXEL_MEAN), pixel_std=list(osam.PIXEL_STD))
        model.load_state_dict(sd, strict=True)        # every parameter of the reference model is covered
        model.eval()
        kw = cfg.oracle_kw()
        oh, ow = (100, 126) if arch == "tiny" else (96, 70)
        img = (synth.rand(3, f"sam_img_{arch}", (oh, ow, 3)) * 255).to(torch.uint8).numpy()
        inp = osam.apply_image(img, cfg.img_size)
        t = torch.as_tensor(inp).permute(2, 0, 1).contiguous()[None].float()
        print(f"[sam {arch}] input {inp.shape}")
        with torch.no_grad():
            ref_pre = model.preprocess(t)
            ref_feat = model.image_encoder(ref_pre)
        check("preprocess", ref_pre, osam.preprocess(t, cfg.img_size))
        check("image_encoder", ref_feat, osam.image_encoder(sd, ref_pre, kw), tol=2e-5)
        pts = torch.as_tensor(osam.apply_coords(osam.build_point_grid(3) * np.array([[ow, oh]]), (oh, ow), cfg.img_size))
        labels = torch.ones(len(pts), 1, dtype=torch.int)
        with torch.no_grad():
            ref_sparse, ref_dense = model.prompt_encoder(points=(pts[:, None, :], labels), boxes=None, masks=None)
            ref_pe = model.prompt_encoder.get_dense_pe()
            ref_low, ref_iou = model.mask_decoder(image_embeddings=ref_feat, image_pe=ref_pe, sparse_prompt_embeddings=ref_sparse,
                                                  dense_prompt_embeddings=ref_dense, multimask_output=True)
            ref_masks = model.postprocess_masks(ref_low, tuple(t.shape[-2:]), (oh, ow))
        check("sparse embeddings", ref_sparse, osam.embed_points(sd, pts[:, None, :].float(), labels.float(), (cfg.img_size, cfg.img_size)))
        check("dense pe", ref_pe, osam.dense_pe(sd, (G, G)))
        state = dict(features=ref_feat, input_size=tuple(t.shape[-2:]), original_size=(oh, ow))
        o_masks, o_iou, o_low = osam.predict_torch(sd, kw, state, pts[:, None, :], labels)
        check("low_res_masks", ref_low, o_low, tol=2e-5)
        check("iou_predictions", ref_iou, o_iou, tol=2e-5)
        check("masks (logits)", ref_masks, o_masks, tol=2e-5)
        # box prompts, alone and together with two points (no padding token then)
        boxes = torch.tensor([[5.0, 9.0, 60.0, 70.0], [20.5, 3.0, 90.0, 50.25], [0.0, 0.0, 30.0, 30.0]])
        pts2 = pts[:3, None, :].repeat(1, 2, 1).float() + torch.tensor([[0.0, 0.0], [7.0, -3.0]])
        lab2 = torch.tensor([[1, 0]] * 3, dtype=torch.int)
        with torch.no_grad():
            sp_b, de_b = model.prompt_encoder(points=None, boxes=boxes, masks=None)
            sp_pb, _ = model.prompt_encoder(points=(pts2, lab2), boxes=boxes, masks=None)
            low_pb, iou_pb = model.mask_decoder(image_embeddings=ref_feat, image_pe=ref_pe, sparse_prompt_embeddings=sp_pb,
                                                dense_prompt_embeddings=de_b, multimask_output=True)
        check("box embeddings", sp_b, osam.embed_boxes(sd, boxes, (cfg.img_size, cfg.img_size)))
        _, o_iou_pb, o_low_pb = osam.predict_torch(sd, kw, state, pts2, lab2, boxes=boxes)
        check("points + box low_res_masks", low_pb, o_low_pb, tol=2e-5)
        check("points + box iou", iou_pb, o_iou_pb, tol=2e-5)
        out[f"sam_{arch}_boxes"] = boxes.numpy()
        out[f"sam_{arch}_box_points"] = pts2.numpy()
        out[f"sam_{arch}_box_sparse"] = sp_b.numpy()
        out[f"sam_{arch}_box_low_res"] = low_pb.numpy()
        out[f"sam_{arch}_box_iou"] = iou_pb.numpy()
        # a mask prompt (the low-resolution logits of the first prediction fed back) with two points
        mask_in = ref_low[:3, :1]
        with torch.no_grad():
            sp_m, de_m = model.prompt_encoder(points=(pts2, lab2), boxes=None, masks=mask_in)
            low_m, iou_m = model.mask_decoder(image_embeddings=ref_feat, image_pe=ref_pe, sparse_prompt_embeddings=sp_m,
                                              dense_prompt_embeddings=de_m, multimask_output=True)
        check("mask embedding", de_m, osam.embed_masks(sd, mask_in), tol=2e-5)
        _, o_iou_m, o_low_m = osam.predict_torch(sd, kw, state, pts2, lab2, mask_input=mask_in)
        check("points + mask low_res_masks", low_m, o_low_m, tol=2e-5)
        check("points + mask iou", iou_m, o_iou_m, tol=2e-5)
        out[f"sam_{arch}_mask_dense"] = de_m.numpy()
        out[f"sam_{arch}_mask_low_res"] = low_m.numpy()
        out[f"sam_{arch}_mask_iou"] = iou_m.numpy()
        # amg helpers
        m = ref_masks.flatten(0, 1)
        check("stability score", amg.calculate_stability_score(m, 0.0, 1.0), osam.stability_score(m, 0.0, 1.0))
        check("mask boxes", amg.batched_mask_to_box(m > 0), osam.mask_to_box(m > 0))
        check("point grid", amg.build_point_grid(5), osam.build_point_grid(5))
        for size, layers, ratio in (((oh, ow), 2, 512 / 1500), ((480, 640), 1, 0.25), ((333, 77), 3, 0.4)):
            rb, rl, _ = amg.generate_crop_boxes(size, layers, ratio)
            ob, ol = osam.generate_crop_boxes(size, layers, ratio)
            assert rb == ob and rl == ol, (size, layers)
        bx = torch.tensor([[0, 0, 30, 30], [25, 3, 60, 40], [5, 28, 20, 62], [40, 40, 50, 50]], dtype=torch.float)
        for cb in ([10, 20, 73, 84], [0, 0, 63, 64]):
            assert torch.equal(amg.is_box_near_crop_edge(bx, cb, [0, 0, ow, oh]), osam.is_box_near_crop_edge(bx, cb, [0, 0, ow, oh]))
        rles = amg.mask_to_rle_pytorch(m[:4] > 0)
        assert rles == osam.mask_to_rle(m[:4] > 0)
        assert all(np.array_equal(amg.rle_to_mask(r), osam.rle_to_mask(r)) and np.array_equal(osam.rle_to_mask(r), (m[i] > 0).numpy())
                   for i, r in enumerate(rles))
        print("  crop boxes / crop-edge filter / run-length codes: equal to utils.amg")
        out[f"sam_{arch}_crop_boxes"] = np.asarray(amg.generate_crop_boxes((oh, ow), 2, 512 / 1500)[0], dtype=np.int64)
        out[f"sam_{arch}_rle0_counts"] = np.asarray(rles[0]["counts"], dtype=np.int64)
        out[f"sam_{arch}_image"] = img
        out[f"sam_{arch}_features"] = ref_feat.numpy()
        out[f"sam_{arch}_points"] = pts.numpy()
        out[f"sam_{arch}_low_res"] = ref_low.numpy()
        out[f"sam_{arch}_iou"] = ref_iou.numpy()
        out[f"sam_{arch}_masks"] = ref_masks.numpy().astype(np.float32)


def golden_tokens():
    _stub("ftfy", fix_text=lambda s: s)
    tk = _load_by_path("ref_simple_tokenizer", os.path.join(REF, "clip/simple_tokenizer.py"))
    tok = tk.SimpleTokenizer()
    table = {}
    for name in ("object", "dog", "person", "car"):
        for t in (f"a {name}.", f"a photo of {name}.", f"a photo without {name}."):
            table[t] = [tok.encoder["<|startoftext|>"]] + tok.encode(t) + [tok.encoder["<|endoftext|>"]]
    with open(os.path.join(HERE, "prompt_tokens.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    for t, ids in table.items():
        if t in synth.PROMPT_TOKENS:
            assert synth.PROMPT_TOKENS[t] == ids, (t, ids)
    print(f"[tokens] {len(table)} prompts written")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    golden_dino(out)
    golden_clip(out)
    golden_alphaclip(out)
    golden_evaluator(out)
    golden_sam(out)
    golden_tokens()
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"),
          f"{os.path.getsize(os.path.join(HERE, 'reference_vectors.npz')) / 1024:.1f} KiB")
