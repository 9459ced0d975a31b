"""Encoders on the HIP path against (a) the vectors the reference modules produced
(tests/golden/reference_vectors.npz) and (b) the CPU oracle on the same seeded weights.
fp32 tolerance 1e-4 (the north-star bound); bf16 runs are checked for closeness only."""
import importlib
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, maxdiff
from oracle import pipeline as opipe
from oracle import vit as ovit

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "reference_vectors.npz"))
TOL = 1e-4


def _models(pkg):
    return importlib.import_module(pkg.__name__ + ".models")


@pytest.mark.parametrize("arch,img", [("vit_tiny2", 70), ("vit_tiny3", 84)])
def test_dinov2_fp32_matches_reference_vectors(pkg, cuda, arch, img):
    synth = pkg.synth
    cfg = synth.dino_cfg(arch, img_size=img)
    sd = synth.dinov2_state_dict(cfg, seed=1)
    m = _models(pkg).DinoVisionTransformer(cfg, sd, device=cuda, dtype=torch.float32)
    x = synth.randn(7, f"dino_in_{arch}", (2, 3, img, img), 1.0).to(cuda)
    out = m.forward_features(x)
    assert maxdiff(out["x_prenorm"], G[f"dino_{arch}_x_prenorm"]) < TOL
    attn = m.get_last_self_attention(x[:1])
    assert len(attn) == cfg.depth and attn[0].shape == (1, cfg.num_heads, 1 + 4 + (img // 14) ** 2, 1 + 4 + (img // 14) ** 2)
    assert maxdiff(attn[-1], G[f"dino_{arch}_attn_last"]) < TOL
    # fused mean of the attention maps (never materialised)
    P = (img // 14) ** 2
    acc = torch.zeros(P, (P + 63) // 64 * 64, device=cuda)
    m.forward_tokens(x[:1], mean_attn=(acc, 24))
    assert maxdiff(acc[:, :P], G[f"dino_{arch}_mean_attn"]) < TOL


def test_dinov2_input_of_another_size_than_the_constructor(pkg, cuda):
    """interpolate_pos_encoding (vision_transformer.py:188-221): a 98 x 84 input to a tower built for 70 x 70,
    against the reference module's output, plus the attention-mean side output on the non-square grid."""
    synth = pkg.synth
    cfg = synth.dino_cfg("vit_tiny2", img_size=70)
    sd = synth.dinov2_state_dict(cfg, seed=1)
    m = _models(pkg).DinoVisionTransformer(cfg, sd, device=cuda, dtype=torch.float32)
    x = synth.randn(7, "dino_in_interp", (1, 3, 98, 84), 1.0).to(cuda)
    out = m.forward_features(x)
    assert out["x_prenorm"].shape == (1, 1 + 4 + 7 * 6, cfg.embed_dim)
    assert maxdiff(out["x_prenorm"], G["dino_vit_tiny2_interp_x_prenorm"]) < TOL
    P = 42
    acc = torch.zeros(P, 64, device=cuda)
    m.forward_tokens(x, mean_attn=(acc, 24))
    _, attn = ovit.dino_forward(sd, x.cpu(), want_attn=True, depth=cfg.depth, heads=cfg.num_heads, patch=14, num_regs=4, ffn="mlp")
    assert maxdiff(acc[:, :P], opipe.mean_attention(attn, cfg.depth, 4)) < TOL
    # the constructor's own size still takes the table as loaded
    x0 = synth.randn(7, "dino_in_vit_tiny2", (2, 3, 70, 70), 1.0).to(cuda)
    assert maxdiff(m.forward_features(x0)["x_prenorm"], G["dino_vit_tiny2_x_prenorm"]) < TOL


def test_dinov2_swiglu_fp32(pkg, cuda):
    synth = pkg.synth
    cfg = synth.DinoCfg(embed_dim=128, depth=2, num_heads=2, img_size=70, ffn="swiglu")
    sd = synth.dinov2_state_dict(cfg, seed=2)
    m = _models(pkg).DinoVisionTransformer(cfg, sd, device=cuda, dtype=torch.float32)
    x = synth.randn(7, "dino_in_swiglu", (1, 3, 70, 70), 1.0).to(cuda)
    assert maxdiff(m.forward_features(x)["x_prenorm"], G["dino_swiglu_x_prenorm"]) < TOL


def test_dinov2_bf16_flash_and_unfused_agree(pkg, cuda, monkeypatch):
    synth = pkg.synth
    cfg = synth.dino_cfg("vit_tiny3", img_size=84)
    sd = synth.dinov2_state_dict(cfg, seed=1)
    x = synth.randn(7, "dino_in_vit_tiny3", (2, 3, 84, 84), 1.0).to(cuda)
    outs, accs = {}, {}
    P = 36
    for flash in ("1", "0"):
        monkeypatch.setenv("MARS_FLASH", flash)
        m = _models(pkg).DinoVisionTransformer(cfg, sd, device=cuda, dtype=torch.bfloat16)
        acc = torch.zeros(P, 64, device=cuda)
        outs[flash] = m.forward_tokens(x[:1], mean_attn=(acc, 24)).clone()
        accs[flash] = acc
    ref = G["dino_vit_tiny3_x_prenorm"][:1]
    scale = float(np.abs(ref).max())
    assert maxdiff(outs["1"], ref) < 0.05 * scale and maxdiff(outs["0"], ref) < 0.05 * scale
    assert maxdiff(outs["1"], outs["0"]) < 0.05 * scale
    assert maxdiff(accs["1"][:, :P], G["dino_vit_tiny3_mean_attn"]) < 5e-3
    assert maxdiff(accs["0"][:, :P], G["dino_vit_tiny3_mean_attn"]) < 5e-3


def test_clip_fp32_image_text_gradcam(pkg, cuda):
    synth = pkg.synth
    cfg = synth.CLIP_ARCHS["tiny/16"]
    sd = synth.clip_state_dict(cfg, seed=3)
    m = _models(pkg).CLIP(cfg, sd, device=cuda, dtype=torch.float32)
    x = synth.randn(7, "clip_in", (1, 3, 80, 80), 1.0).to(cuda)
    feats, attn = m.encode_image(x, 80, 80)
    assert feats.shape == (26, 1, cfg.vision_width) and len(attn) == cfg.vision_layers - 1
    assert maxdiff(feats.permute(1, 0, 2), G["clip_feats"]) < TOL
    toks = synth.tokenize(["a photo of object.", "a photo without object.", "a object."])
    text = m.encode_text(toks)
    assert maxdiff(text, G["clip_text"]) < TOL
    # text pair exactly as the oracle prepares it, then logits through the reference-named method
    t2 = opipe.vta_text_feats(sd, dict(text_layers=cfg.transformer_layers, text_heads=cfg.transformer_heads), toks[0:1], toks[1:2])
    logits, attn_last = m.forward_last_layer(feats, t2.to(cuda))
    assert maxdiff(logits, G["clip_logits"]) < TOL
    mean_all = torch.stack([a[0] for a in attn] + [attn_last[0]]).mean(0)
    assert maxdiff(mean_all, G["clip_attn_mean"]) < TOL
    X, _ = m.encode_image_tokens(x)
    cam, logits2, grad = m.gradcam(X, t2.to(cuda))
    assert maxdiff(logits2, G["clip_logits"]) < TOL
    assert maxdiff(cam, G["clip_cam"]) < 5e-4       # min-max of a small-magnitude map amplifies rounding
    # gradient itself against autograd on the oracle graph
    sdc = {k: v for k, v in sd.items()}
    of, _ = ovit.clip_encode_image(sdc, x.cpu(), layers=cfg.vision_layers, heads=cfg.vision_heads, patch=cfg.vision_patch_size)
    prefix = f"visual.transformer.resblocks.{cfg.vision_layers - 1}"
    ln1 = ovit._ln(of, sdc, prefix + ".ln_1", 1e-5).detach().requires_grad_(True)
    a, _ = ovit._mha(ln1, sdc[prefix + ".attn.in_proj_weight"], sdc[prefix + ".attn.in_proj_bias"],
                     sdc[prefix + ".attn.out_proj.weight"], sdc[prefix + ".attn.out_proj.bias"], cfg.vision_heads)
    x1 = of + a
    h = torch.nn.functional.linear(ovit._ln(x1, sdc, prefix + ".ln_2", 1e-5), sdc[prefix + ".mlp.c_fc.weight"], sdc[prefix + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    x2 = x1 + torch.nn.functional.linear(h, sdc[prefix + ".mlp.c_proj.weight"], sdc[prefix + ".mlp.c_proj.bias"])
    y = ovit._ln(x2, sdc, "visual.ln_post", 1e-5)
    f = y[:, 1:, :].mean(dim=1) @ sdc["visual.proj"]
    f = f / f.norm(dim=1, keepdim=True)
    lg = (sdc["logit_scale"].exp() * f @ (t2 / t2.norm(dim=1, keepdim=True)).t()).softmax(-1)
    (gref,) = torch.autograd.grad(lg[:, 0].sum(), ln1)
    assert maxdiff(grad, gref[0]) < 2e-4 * float(gref.abs().max()) + 1e-8


def test_alphaclip_fp32_and_bf16(pkg, cuda):
    synth = pkg.synth
    cfg = synth.CLIP_ARCHS["tiny/14"]
    sd = synth.clip_state_dict(cfg, seed=4, alpha=True)
    R = cfg.image_resolution
    img1 = synth.randn(7, "aclip_img", (1, 3, R, R), 1.0).to(cuda)
    alpha = synth.randn(7, "aclip_alpha", (3, 1, R, R), 1.5).to(cuda)
    m = _models(pkg).AlphaCLIP(cfg, sd, device=cuda, dtype=torch.float32)
    shared = m.visual(img1, alpha).clone()                       # one image broadcast over the alpha planes
    assert maxdiff(shared, G["aclip_visual"]) < TOL
    full = m.visual(img1.repeat(3, 1, 1, 1), alpha).clone()      # reference calling convention
    assert maxdiff(full, G["aclip_visual"]) < TOL
    assert maxdiff(m.encode_text(synth.tokenize(["a object."])), G["aclip_text"]) < TOL
    mb = _models(pkg).AlphaCLIP(cfg, sd, device=cuda, dtype=torch.bfloat16)
    fb = mb.visual(img1, alpha)
    cos = torch.nn.functional.cosine_similarity(fb.float().cpu(), torch.from_numpy(G["aclip_visual"]), dim=1)
    assert float(cos.min()) > 0.995


def test_alphaclip_folded_layernorms(pkg, cuda):
    """bf16 tower with its LayerNorms folded into the products around them against the same tower with
    stand-alone passes: both are one bf16 rounding per operand away from the f32 tower."""
    synth = pkg.synth
    cfg = synth.CLIP_ARCHS["tiny/14"]
    sd = synth.clip_state_dict(cfg, seed=4, alpha=True)
    R = cfg.image_resolution
    img1 = synth.randn(7, "aclip_img", (1, 3, R, R), 1.0).to(cuda)
    alpha = synth.randn(7, "aclip_alpha", (3, 1, R, R), 1.5).to(cuda)
    ref = torch.from_numpy(G["aclip_visual"]).to(cuda)
    M = _models(pkg)
    folded = M.AlphaCLIP(cfg, sd, device=cuda, dtype=torch.bfloat16, fold_ln=True)
    alone = M.AlphaCLIP(cfg, sd, device=cuda, dtype=torch.bfloat16, fold_ln=False)
    assert folded.fold_ln and folded.vtower.blocks[0].folded and not folded.vtower.blocks[-1].folded
    assert not alone.fold_ln and not alone.vtower.blocks[0].folded
    ff, fa = folded.visual(img1, alpha).clone(), alone.visual(img1, alpha).clone()
    e_fold, e_alone = maxdiff(ff, ref), maxdiff(fa, ref)
    assert e_fold < 1.5 * e_alone + 1e-3 * float(ref.abs().max()), (e_fold, e_alone)
    folded.first_token_last_block = False                      # the whole last block behind the folded run
    assert maxdiff(folded.visual(img1, alpha), ff) < 0.05 * max(1.0, float(ff.abs().max()))


def test_alphaclip_folded_layernorms_fall_back_on_large_row_means(pkg, cuda):
    """A residual stream whose rows sit far from zero (|mean| / std of about 30: a shifted ln_pre bias) is what the
    folded form is bad at; the folded runs report it, and from the next call on the tower runs its LayerNorms
    stand-alone: same features as a tower built that way.  A stream with ordinary rows keeps the folded form."""
    synth = pkg.synth
    cfg = synth.CLIP_ARCHS["tiny/14"]
    sd = synth.clip_state_dict(cfg, seed=4, alpha=True)
    R = cfg.image_resolution
    img1 = synth.randn(7, "aclip_img", (1, 3, R, R), 1.0).to(cuda)
    alpha = synth.randn(7, "aclip_alpha", (3, 1, R, R), 1.5).to(cuda)
    M = _models(pkg)
    calm = M.AlphaCLIP(cfg, sd, device=cuda, dtype=torch.bfloat16, fold_ln=True)
    for _ in range(3):
        calm.visual(img1, alpha)
        torch.cuda.synchronize()
    assert calm.fold_ln and calm.vtower.blocks[0].folded
    shifted = dict(sd)
    shifted["visual.ln_pre.bias"] = sd["visual.ln_pre.bias"] + 30.0
    folded = M.AlphaCLIP(cfg, shifted, device=cuda, dtype=torch.bfloat16, fold_ln=True)
    alone = M.AlphaCLIP(cfg, shifted, device=cuda, dtype=torch.bfloat16, fold_ln=False)
    want = alone.visual(img1, alpha).clone()
    first = folded.visual(img1, alpha).clone()                  # still folded: nothing has been reported yet
    torch.cuda.synchronize()
    assert folded.fold_ln
    with pytest.warns(UserWarning, match="stand-alone"):
        second = folded.visual(img1, alpha).clone()
    assert not folded.fold_ln and not folded.vtower.blocks[0].folded
    assert torch.equal(second, want)
    # what the report is about: the folded features of such a stream are further from the stand-alone ones than
    # the two forms are from each other on an ordinary stream
    ordinary = maxdiff(calm.visual(img1, alpha), M.AlphaCLIP(cfg, sd, device=cuda, dtype=torch.bfloat16, fold_ln=False).visual(img1, alpha))
    assert maxdiff(first, want) > ordinary
    # MARS_FOLD_LN_LIMIT <= 0 (here: the attribute it sets): nobody looks
    blind = M.AlphaCLIP(cfg, shifted, device=cuda, dtype=torch.bfloat16, fold_ln=True)
    blind.fold_limit = 0.0
    for _ in range(2):
        blind.visual(img1, alpha)
        torch.cuda.synchronize()
    assert blind.fold_ln


def test_backbone_loader_seam(pkg, cuda):
    BL = importlib.import_module(pkg.__name__ + ".utils.backbone_loader").BackboneLoader
    enc, tr = BL.load_backbone("dinov2", "vit_tiny2", None, "cuda:0", dict(img_size=70, patch_size=14, num_register_tokens=4, seed=1))
    assert enc.family == "vits_dino2" and enc.embed_dim == 128 and enc.patch_size == 14 and enc.num_register_tokens == 4
    x = pkg.synth.randn(7, "dino_in_vit_tiny2", (2, 3, 70, 70), 1.0).to(cuda)
    assert maxdiff(enc.forward_features(x)["x_prenorm"], G["dino_vit_tiny2_x_prenorm"]) < TOL
    with pytest.raises(AssertionError):
        BL.load_backbone("resnet101", "resnet101")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_alphaclip_first_token_last_block(pkg, cuda, dtype):
    """The last AlphaCLIP block formed for the class token only gives the features of the full block."""
    synth = pkg.synth
    cfg = synth.CLIP_ARCHS["tiny/14"]
    sd = synth.clip_state_dict(cfg, 3, alpha=True)
    m = pkg.models.AlphaCLIP(cfg, sd, device=cuda, dtype=dtype)
    g = torch.Generator().manual_seed(5)
    R = cfg.image_resolution
    img = torch.rand(1, 3, R, R, generator=g).to(cuda)
    alpha = torch.randn(5, 1, R, R, generator=g).to(cuda)
    m.first_token_last_block = True
    a = m.visual(img, alpha).clone()
    m.first_token_last_block = False
    b = m.visual(img, alpha).clone()
    tol = 2e-5 if dtype == torch.float32 else 0.05
    assert float((a - b).abs().max()) < tol * max(1.0, float(b.abs().max()))
