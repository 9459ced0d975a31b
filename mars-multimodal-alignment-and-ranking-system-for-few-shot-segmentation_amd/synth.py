"""Seeded synthetic weights, episodes and proposals.

No checkpoint or dataset is reachable offline, so every tensor the ranking path
consumes is regenerated from ``(seed, tensor name, shape)`` with the CPU torch
generator.  The same call yields bit-identical tensors in any process, which is
how the parity tests, the oracle and the bench agree on inputs without shipping
gigabytes.

State dicts use the key layout of the reference checkpoints so that a real file
drops in unchanged:
  * DINOv2   -- /root/reference/dinov2/models/vision_transformer.py:109-177
  * CLIP     -- /root/reference/clip/model.py:568-616 (OpenAI state dict)
  * AlphaCLIP-- /root/reference/alpha_clip/model.py:581-638 (after the
                ``in_proj_weight -> in_proj.weight`` rename of lines 618-628)

The weights are deliberately *non-degenerate*: layer-scale gammas are O(1) and
the linear layers are variance preserving, so attention is far from uniform and
every residual branch matters.  With the reference's default init (gamma 1e-5,
std 0.02) a broken kernel would hide behind near-identity blocks.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch


# ----------------------------------------------------------------------------
# deterministic tensors
# ----------------------------------------------------------------------------
def stable_seed(seed: int, name: str) -> int:
    return (zlib.crc32(name.encode("utf-8")) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def randn(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(stable_seed(seed, name))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std + mean


def rand(seed: int, name: str, shape) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(stable_seed(seed, name))
    return torch.rand(tuple(shape), generator=g, dtype=torch.float32)


# ----------------------------------------------------------------------------
# architectures
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class DinoCfg:
    embed_dim: int
    depth: int
    num_heads: int
    img_size: int = 518
    patch_size: int = 14
    num_regs: int = 4
    ffn: str = "mlp"          # "mlp" | "swiglu"
    mlp_ratio: float = 4.0

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def hidden(self) -> int:
        h = int(self.embed_dim * self.mlp_ratio)
        if self.ffn == "swiglu":  # /root/reference/dinov2/layers/swiglu_ffn.py:57
            h = (int(h * 2 / 3) + 7) // 8 * 8
        return h


# names follow /root/reference/dinov2/models/vision_transformer.py:381-437; "vit_tiny*"
# are test-only shapes that keep head_dim 64.
DINO_ARCHS = {
    "vit_tiny2": dict(embed_dim=128, depth=2, num_heads=2),
    "vit_tiny3": dict(embed_dim=192, depth=3, num_heads=3),
    "vit_small": dict(embed_dim=384, depth=12, num_heads=6),
    "vit_base": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_large": dict(embed_dim=1024, depth=24, num_heads=16),
    "vit_giant2": dict(embed_dim=1536, depth=40, num_heads=24, ffn="swiglu"),
}


def dino_cfg(arch: str, img_size: int = 518, patch_size: int = 14, num_regs: int = 4) -> DinoCfg:
    return DinoCfg(img_size=img_size, patch_size=patch_size, num_regs=num_regs, **DINO_ARCHS[arch])


@dataclass(frozen=True)
class ClipCfg:
    """Shapes of one CLIP model (vision ViT + text transformer)."""
    embed_dim: int            # joint space
    image_resolution: int     # native resolution (positional table size)
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int = 77
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size


CLIP_ARCHS = {
    # OpenAI shapes (/root/reference/clip/model.py:602-606)
    "ViT-B/16": ClipCfg(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-L/14": ClipCfg(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": ClipCfg(768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    # test-only shapes
    "tiny/16": ClipCfg(64, 64, 3, 128, 16, 77, 49408, 128, 2, 2),
    "tiny/14": ClipCfg(64, 56, 3, 128, 14, 77, 49408, 128, 2, 2),
}


# ----------------------------------------------------------------------------
# state dicts
# ----------------------------------------------------------------------------
def _linear(sd, seed, prefix, out_f, in_f, gain=1.0, bias_std=0.02):
    sd[prefix + ".weight"] = randn(seed, prefix + ".weight", (out_f, in_f), gain / math.sqrt(in_f))
    sd[prefix + ".bias"] = randn(seed, prefix + ".bias", (out_f,), bias_std)


def _norm(sd, seed, prefix, dim):
    sd[prefix + ".weight"] = randn(seed, prefix + ".weight", (dim,), 0.1, 1.0)
    sd[prefix + ".bias"] = randn(seed, prefix + ".bias", (dim,), 0.05)


def dinov2_state_dict(cfg: DinoCfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    D, P = cfg.embed_dim, cfg.grid * cfg.grid
    p = cfg.patch_size
    sd: Dict[str, torch.Tensor] = {}
    sd["cls_token"] = randn(seed, "cls_token", (1, 1, D), 0.5)
    sd["pos_embed"] = randn(seed, "pos_embed", (1, 1 + P, D), 0.3)
    if cfg.num_regs:
        sd["register_tokens"] = randn(seed, "register_tokens", (1, cfg.num_regs, D), 0.5)
    sd["mask_token"] = torch.zeros(1, D)
    sd["patch_embed.proj.weight"] = randn(seed, "patch_embed.proj.weight", (D, 3, p, p), 1.0 / math.sqrt(3 * p * p))
    sd["patch_embed.proj.bias"] = randn(seed, "patch_embed.proj.bias", (D,), 0.02)
    for i in range(cfg.depth):
        b = f"blocks.{i}"
        _norm(sd, seed, b + ".norm1", D)
        _linear(sd, seed, b + ".attn.qkv", 3 * D, D, gain=1.3)
        _linear(sd, seed, b + ".attn.proj", D, D)
        sd[b + ".ls1.gamma"] = randn(seed, b + ".ls1.gamma", (D,), 0.1, 0.6)
        _norm(sd, seed, b + ".norm2", D)
        if cfg.ffn == "mlp":
            _linear(sd, seed, b + ".mlp.fc1", cfg.hidden, D)
            _linear(sd, seed, b + ".mlp.fc2", D, cfg.hidden)
        else:
            _linear(sd, seed, b + ".mlp.w12", 2 * cfg.hidden, D)
            _linear(sd, seed, b + ".mlp.w3", D, cfg.hidden)
        sd[b + ".ls2.gamma"] = randn(seed, b + ".ls2.gamma", (D,), 0.1, 0.6)
    _norm(sd, seed, "norm", D)
    return sd


def _clip_resblock(sd, seed, prefix, width, fused_attr: bool):
    """One pre-LN CLIP block.  ``fused_attr`` selects nn.MultiheadAttention naming
    (``attn.in_proj_weight``) versus AlphaCLIP's ``attn.in_proj.weight``."""
    wname = ".attn.in_proj_weight" if fused_attr else ".attn.in_proj.weight"
    bname = ".attn.in_proj_bias" if fused_attr else ".attn.in_proj.bias"
    sd[prefix + wname] = randn(seed, prefix + wname, (3 * width, width), 1.3 / math.sqrt(width))
    sd[prefix + bname] = randn(seed, prefix + bname, (3 * width,), 0.02)
    _linear(sd, seed, prefix + ".attn.out_proj", width, width, gain=0.7)
    _norm(sd, seed, prefix + ".ln_1", width)
    _linear(sd, seed, prefix + ".mlp.c_fc", 4 * width, width)
    _linear(sd, seed, prefix + ".mlp.c_proj", width, 4 * width, gain=0.7)
    _norm(sd, seed, prefix + ".ln_2", width)


def clip_state_dict(cfg: ClipCfg, seed: int = 0, alpha: bool = False) -> Dict[str, torch.Tensor]:
    """Full CLIP state dict.  ``alpha=True`` gives the AlphaCLIP layout (extra
    ``visual.conv1_alpha.weight``; visual attention stored as ``in_proj.weight``)."""
    W, p = cfg.vision_width, cfg.vision_patch_size
    sd: Dict[str, torch.Tensor] = {}
    tag = "aclip." if alpha else "clip."
    s = stable_seed(seed, tag)  # decorrelate the two families
    sd["visual.conv1.weight"] = randn(s, "visual.conv1.weight", (W, 3, p, p), 1.0 / math.sqrt(3 * p * p))
    if alpha:
        sd["visual.conv1_alpha.weight"] = randn(s, "visual.conv1_alpha.weight", (W, 1, p, p), 1.0 / math.sqrt(p * p))
    sd["visual.class_embedding"] = randn(s, "visual.class_embedding", (W,), 0.5)
    sd["visual.positional_embedding"] = randn(s, "visual.positional_embedding", (cfg.grid * cfg.grid + 1, W), 0.3)
    _norm(sd, s, "visual.ln_pre", W)
    for i in range(cfg.vision_layers):
        _clip_resblock(sd, s, f"visual.transformer.resblocks.{i}", W, fused_attr=not alpha)
    _norm(sd, s, "visual.ln_post", W)
    sd["visual.proj"] = randn(s, "visual.proj", (W, cfg.embed_dim), 1.0 / math.sqrt(W))
    # text tower (always nn.MultiheadAttention naming)
    TW = cfg.transformer_width
    sd["token_embedding.weight"] = randn(s, "token_embedding.weight", (cfg.vocab_size, TW), 0.5)
    sd["positional_embedding"] = randn(s, "positional_embedding", (cfg.context_length, TW), 0.3)
    for i in range(cfg.transformer_layers):
        _clip_resblock(sd, s, f"transformer.resblocks.{i}", TW, fused_attr=True)
    _norm(sd, s, "ln_final", TW)
    sd["text_projection"] = randn(s, "text_projection", (TW, cfg.embed_dim), 1.0 / math.sqrt(TW))
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    return sd


# ----------------------------------------------------------------------------
# synthetic episodes
# ----------------------------------------------------------------------------
# Segment Anything (key layout of /root/reference/segment_anything/build_sam.py:47-106)
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class SamCfg:
    embed_dim: int
    depth: int
    num_heads: int
    global_attn_indexes: Tuple[int, ...]
    img_size: int = 1024
    patch_size: int = 16
    window_size: int = 14
    out_chans: int = 256          # prompt / decoder embedding dim
    dec_depth: int = 2
    dec_heads: int = 8
    dec_mlp_dim: int = 2048
    iou_head_hidden: int = 256
    iou_head_depth: int = 3
    mask_in_chans: int = 16

    def oracle_kw(self) -> dict:
        return dict(img_size=self.img_size, patch=self.patch_size, depth=self.depth, heads=self.num_heads,
                    window_size=self.window_size, global_attn_indexes=tuple(self.global_attn_indexes), use_rel_pos=True,
                    dec_depth=self.dec_depth, dec_heads=self.dec_heads, iou_head_depth=self.iou_head_depth)


SAM_ARCHS = {
    "vit_h": SamCfg(1280, 32, 16, (7, 15, 23, 31)),
    "vit_l": SamCfg(1024, 24, 16, (5, 11, 17, 23)),
    "vit_b": SamCfg(768, 12, 12, (2, 5, 8, 11)),
    # test-size models: a window that does not divide the token grid (padding), one global block
    "tiny": SamCfg(64, 3, 2, (1,), img_size=128, window_size=3, out_chans=64, dec_heads=2, dec_mlp_dim=128, iou_head_hidden=64),
    "tiny80": SamCfg(320, 2, 4, (1,), img_size=96, window_size=4, out_chans=64, dec_heads=4, dec_mlp_dim=96, iou_head_hidden=48),
    # the real geometry (1024 px, 64 x 64 tokens, 14 x 14 windows padded to 70, 256 decoder channels) on a narrow, shallow tower
    "geom": SamCfg(128, 2, 2, (1,), dec_mlp_dim=256),
    # 14 x 14 windows (the shipped window) on a 20 x 20 token grid: padded border windows at the size the windows' kernels take
    "win14": SamCfg(320, 2, 4, (1,), img_size=320, window_size=14, out_chans=64, dec_heads=4, dec_mlp_dim=96, iou_head_hidden=48),
}


def sam_state_dict(cfg: SamCfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    D, C, G = cfg.embed_dim, cfg.out_chans, cfg.img_size // cfg.patch_size
    hd = D // cfg.num_heads
    fan = 3 * cfg.patch_size ** 2
    sd["image_encoder.patch_embed.proj.weight"] = randn(seed, "sam.pe.w", (D, 3, cfg.patch_size, cfg.patch_size), std=1.0 / math.sqrt(fan))
    sd["image_encoder.patch_embed.proj.bias"] = randn(seed, "sam.pe.b", (D,), std=0.02)
    sd["image_encoder.pos_embed"] = randn(seed, "sam.pos", (1, G, G, D), std=0.3)
    for i in range(cfg.depth):
        p = f"image_encoder.blocks.{i}"
        size = G if i in cfg.global_attn_indexes else cfg.window_size
        _norm(sd, seed, p + ".norm1", D)
        _linear(sd, seed, p + ".attn.qkv", 3 * D, D, gain=1.5)
        _linear(sd, seed, p + ".attn.proj", D, D)
        sd[p + ".attn.rel_pos_h"] = randn(seed, p + ".rph", (2 * size - 1, hd), std=0.4 / math.sqrt(hd))
        sd[p + ".attn.rel_pos_w"] = randn(seed, p + ".rpw", (2 * size - 1, hd), std=0.4 / math.sqrt(hd))
        _norm(sd, seed, p + ".norm2", D)
        _linear(sd, seed, p + ".mlp.lin1", 4 * D, D)
        _linear(sd, seed, p + ".mlp.lin2", D, 4 * D)
    sd["image_encoder.neck.0.weight"] = randn(seed, "sam.neck0", (C, D, 1, 1), std=1.0 / math.sqrt(D))
    _norm(sd, seed, "image_encoder.neck.1", C)
    sd["image_encoder.neck.2.weight"] = randn(seed, "sam.neck2", (C, C, 3, 3), std=1.0 / math.sqrt(9 * C))
    _norm(sd, seed, "image_encoder.neck.3", C)
    # prompt encoder
    sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = randn(seed, "sam.pe_gauss", (2, C // 2), std=1.0)
    for i in range(4):
        sd[f"prompt_encoder.point_embeddings.{i}.weight"] = randn(seed, f"sam.pt{i}", (1, C), std=0.5)
    sd["prompt_encoder.not_a_point_embed.weight"] = randn(seed, "sam.nap", (1, C), std=0.5)
    sd["prompt_encoder.no_mask_embed.weight"] = randn(seed, "sam.nomask", (1, C), std=0.5)
    mc = cfg.mask_in_chans
    sd["prompt_encoder.mask_downscaling.0.weight"] = randn(seed, "sam.md0", (mc // 4, 1, 2, 2), std=0.5)
    sd["prompt_encoder.mask_downscaling.0.bias"] = randn(seed, "sam.md0b", (mc // 4,), std=0.02)
    _norm(sd, seed, "prompt_encoder.mask_downscaling.1", mc // 4)
    sd["prompt_encoder.mask_downscaling.3.weight"] = randn(seed, "sam.md3", (mc, mc // 4, 2, 2), std=0.5 / math.sqrt(mc))
    sd["prompt_encoder.mask_downscaling.3.bias"] = randn(seed, "sam.md3b", (mc,), std=0.02)
    _norm(sd, seed, "prompt_encoder.mask_downscaling.4", mc)
    sd["prompt_encoder.mask_downscaling.6.weight"] = randn(seed, "sam.md6", (C, mc, 1, 1), std=1.0 / math.sqrt(mc))
    sd["prompt_encoder.mask_downscaling.6.bias"] = randn(seed, "sam.md6b", (C,), std=0.02)
    # mask decoder
    sd["mask_decoder.iou_token.weight"] = randn(seed, "sam.iou_tok", (1, C), std=0.5)
    sd["mask_decoder.mask_tokens.weight"] = randn(seed, "sam.mask_tok", (4, C), std=0.5)

    def attn(p, down):
        inner = C // down
        _linear(sd, seed, p + ".q_proj", inner, C, gain=1.5)
        _linear(sd, seed, p + ".k_proj", inner, C, gain=1.5)
        _linear(sd, seed, p + ".v_proj", inner, C)
        _linear(sd, seed, p + ".out_proj", C, inner)
    for i in range(cfg.dec_depth):
        p = f"mask_decoder.transformer.layers.{i}"
        attn(p + ".self_attn", 1)
        attn(p + ".cross_attn_token_to_image", 2)
        attn(p + ".cross_attn_image_to_token", 2)
        for k in (1, 2, 3, 4):
            _norm(sd, seed, f"{p}.norm{k}", C)
        _linear(sd, seed, p + ".mlp.lin1", cfg.dec_mlp_dim, C)
        _linear(sd, seed, p + ".mlp.lin2", C, cfg.dec_mlp_dim)
    attn("mask_decoder.transformer.final_attn_token_to_image", 2)
    _norm(sd, seed, "mask_decoder.transformer.norm_final_attn", C)
    sd["mask_decoder.output_upscaling.0.weight"] = randn(seed, "sam.up0", (C, C // 4, 2, 2), std=1.0 / math.sqrt(C))
    sd["mask_decoder.output_upscaling.0.bias"] = randn(seed, "sam.up0b", (C // 4,), std=0.02)
    _norm(sd, seed, "mask_decoder.output_upscaling.1", C // 4)
    sd["mask_decoder.output_upscaling.3.weight"] = randn(seed, "sam.up3", (C // 4, C // 8, 2, 2), std=1.0 / math.sqrt(C // 4))
    sd["mask_decoder.output_upscaling.3.bias"] = randn(seed, "sam.up3b", (C // 8,), std=0.02)
    for i in range(4):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}.layers"
        _linear(sd, seed, p + ".0", C, C)
        _linear(sd, seed, p + ".1", C, C)
        _linear(sd, seed, p + ".2", C // 8, C)
    dims = [C] + [cfg.iou_head_hidden] * (cfg.iou_head_depth - 1) + [4]
    for j in range(cfg.iou_head_depth):
        _linear(sd, seed, f"mask_decoder.iou_prediction_head.layers.{j}", dims[j + 1], dims[j])
    return sd


# ----------------------------------------------------------------------------
def _box_blur(img: torch.Tensor, k: int = 9) -> torch.Tensor:
    pad = k // 2
    x = torch.nn.functional.pad(img[None], (pad, pad, pad, pad), mode="reflect")
    return torch.nn.functional.avg_pool2d(x, k, stride=1)[0]


def _ellipse(size: int, cy: float, cx: float, ry: float, rx: float) -> torch.Tensor:
    ys = torch.arange(size, dtype=torch.float32)[:, None]
    xs = torch.arange(size, dtype=torch.float32)[None, :]
    return ((((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2) <= 1.0).float()


def _rect(size: int, cy: float, cx: float, ry: float, rx: float) -> torch.Tensor:
    ys = torch.arange(size, dtype=torch.float32)[:, None]
    xs = torch.arange(size, dtype=torch.float32)[None, :]
    return (((ys - cy).abs() <= ry) & ((xs - cx).abs() <= rx)).float()


def _random_shapes(seed: int, name: str, size: int, count: int) -> List[Tuple[float, float, float, float]]:
    u = rand(seed, name, (count, 4))
    out = []
    for i in range(count):
        cy = float(0.2 + 0.6 * u[i, 0]) * size
        cx = float(0.2 + 0.6 * u[i, 1]) * size
        ry = float(0.08 + 0.22 * u[i, 2]) * size
        rx = float(0.08 + 0.22 * u[i, 3]) * size
        out.append((cy, cx, ry, rx))
    return out


def _object_mask(seed: int, name: str, size: int) -> Tuple[torch.Tensor, List[Tuple[float, float, float, float]]]:
    n = 1 + int(rand(seed, name + ".n", (1,))[0] * 3)  # 1..3 ellipses
    shapes = _random_shapes(seed, name + ".shapes", size, n)
    m = torch.zeros(size, size)
    for s in shapes:
        m = torch.maximum(m, _ellipse(size, *s))
    return m, shapes


def make_episode(seed: int, size: int = 518, nshot: int = 1, nprop: int = 64, fold: int = 0,
                 nclass: int = 80) -> Dict[str, torch.Tensor]:
    """One few-shot episode with the batch-dict keys of the reference loaders
    (/root/reference/matcher/data/coco.py:58-67) plus ``mask_proposals`` in the
    on-disk format main_MARS.py reads (/root/reference/main_MARS.py:62):
    float32 {0,1} [N, size, size]."""
    query_img = _box_blur(rand(seed, "query_img", (3, size, size)))
    support_imgs = torch.stack([_box_blur(rand(seed, f"support_img{s}", (3, size, size))) for s in range(nshot)])
    query_mask, gt_shapes = _object_mask(seed, "query_mask", size)
    support_masks = torch.stack([_object_mask(seed, f"support_mask{s}", size)[0] for s in range(nshot)])
    # paint object texture so the features carry some signal shared by support and query
    tex = _box_blur(rand(seed, "texture", (3, size, size)), 5)
    query_img = query_img * (1 - 0.5 * query_mask) + 0.5 * tex * query_mask
    for s in range(nshot):
        support_imgs[s] = support_imgs[s] * (1 - 0.5 * support_masks[s]) + 0.5 * tex * support_masks[s]
    # proposals: ceil(N/8) jittered copies of gt ellipses, the rest random ellipses/rectangles
    n_gt = (nprop + 7) // 8
    props = []
    jit = rand(seed, "prop.jitter", (n_gt, 4))
    for i in range(n_gt):
        cy, cx, ry, rx = gt_shapes[i % len(gt_shapes)]
        j = (jit[i] - 0.5) * 0.2
        props.append(_ellipse(size, cy + float(j[0]) * size * 0.2, cx + float(j[1]) * size * 0.2,
                              max(ry * (1 + float(j[2])), 8.0), max(rx * (1 + float(j[3])), 8.0)))
    shapes = _random_shapes(seed, "prop.shapes", size, nprop - n_gt)
    kinds = rand(seed, "prop.kind", (max(nprop - n_gt, 1),))
    for i, s in enumerate(shapes):
        props.append(_ellipse(size, *s) if kinds[i] < 0.5 else _rect(size, *s))
    mask_proposals = torch.stack(props[:nprop])
    class_id = fold + 4 * (seed % 20) if nclass == 80 else seed % nclass
    return {
        "query_img": query_img[None].contiguous(),
        "query_mask": query_mask[None].contiguous(),
        "support_imgs": support_imgs[None].contiguous(),
        "support_masks": support_masks[None].contiguous(),
        "class_id": torch.tensor([class_id]),
        "query_name": [f"synthetic_{seed}"],
        "mask_proposals": mask_proposals.contiguous(),
    }


# Token ids of the prompts the synthetic benchmark uses, produced once with the
# reference tokenizer (/root/reference/clip/clip.py:205-245; BPE vocabulary
# /root/reference/clip/bpe_simple_vocab_16e6.txt.gz).  49406 = <|startoftext|>,
# 49407 = <|endoftext|>.
PROMPT_TOKENS = {
    "a object.": [49406, 320, 14115, 269, 49407],
    "a photo of object.": [49406, 320, 1125, 539, 14115, 269, 49407],
    "a photo without object.": [49406, 320, 1125, 2193, 14115, 269, 49407],
}


def tokenize(texts, context_length: int = 77) -> torch.Tensor:
    """Look-up tokenizer for the fixed benchmark prompts; returns int32 [n, 77]
    exactly as the reference's ``tokenize`` does for those strings."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, t in enumerate(texts):
        if t not in PROMPT_TOKENS:
            raise KeyError(
                f"prompt {t!r} has no bundled token ids; pass pre-tokenised ids "
                f"(known prompts: {sorted(PROMPT_TOKENS)})")
        ids = PROMPT_TOKENS[t]
        out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out
