"""fp32 CPU restatement of the three ViT towers (test infrastructure only).

Every function takes a plain ``dict[str, Tensor]`` laid out like the reference
checkpoints and returns torch tensors.  Citations give the reference lines each
function follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------
# shared pieces
# ----------------------------------------------------------------------------
def _ln(x, sd: SD, prefix: str, eps: float):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _mha(x, w_in, b_in, w_out, b_out, heads: int, causal: bool = False):
    """Multi-head self attention on [B, T, D]; returns (out, probs[B, h, T, T]).
    Same arithmetic as /root/reference/dinov2/layers/attention.py:49-67 and
    /root/reference/alpha_clip/model.py:220-253: q is scaled by head_dim**-0.5
    before the product."""
    B, T, D = x.shape
    hd = D // heads
    qkv = F.linear(x, w_in, b_in).reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu_(1)
    p = s.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, w_out, b_out), p


# ----------------------------------------------------------------------------
# DINOv2
# ----------------------------------------------------------------------------
def dino_pos_embed(sd: SD, gh: int, gw: int, offset: float = 0.1, antialias: bool = False) -> torch.Tensor:
    """interpolate_pos_encoding, /root/reference/dinov2/models/vision_transformer.py:188-221: the table as
    it is for a square input of the constructor's size (:192-193), otherwise the patch part resized
    bicubically with the scale factors (g + 0.1) / M the reference passes (:204-209; ``interpolate_offset``
    defaults to 0.1, ``interpolate_antialias`` to False, :68-69)."""
    pos = sd["pos_embed"].float()
    N = pos.shape[1] - 1
    if gh * gw == N and gh == gw:
        return pos
    M = int(N ** 0.5)
    assert M * M == N
    dim = pos.shape[-1]
    kw = dict(scale_factor=(float(gh + offset) / M, float(gw + offset) / M)) if offset else dict(size=(gh, gw))
    patch = F.interpolate(pos[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=antialias, **kw)
    assert tuple(patch.shape[-2:]) == (gh, gw)
    return torch.cat([pos[:, :1], patch.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)


def dino_tokens(sd: SD, x: torch.Tensor, patch: int, num_regs: int) -> torch.Tensor:
    """/root/reference/dinov2/models/vision_transformer.py:223-243."""
    B = x.shape[0]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    gh, gw = t.shape[-2:]
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1) + dino_pos_embed(sd, gh, gw)
    if num_regs:
        t = torch.cat([t[:, :1], sd["register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
    return t


def dino_block(sd: SD, i: int, x: torch.Tensor, heads: int, ffn: str):
    """/root/reference/dinov2/layers/block.py:86-114 (eval branch); returns (x, probs)."""
    b = f"blocks.{i}"
    a, p = _mha(_ln(x, sd, b + ".norm1", 1e-6), sd[b + ".attn.qkv.weight"], sd[b + ".attn.qkv.bias"],
                sd[b + ".attn.proj.weight"], sd[b + ".attn.proj.bias"], heads)
    x = x + sd[b + ".ls1.gamma"] * a
    h = _ln(x, sd, b + ".norm2", 1e-6)
    if ffn == "mlp":  # /root/reference/dinov2/layers/mlp.py:35-41, exact GELU
        h = F.linear(F.gelu(F.linear(h, sd[b + ".mlp.fc1.weight"], sd[b + ".mlp.fc1.bias"])),
                     sd[b + ".mlp.fc2.weight"], sd[b + ".mlp.fc2.bias"])
    else:             # /root/reference/dinov2/layers/swiglu_ffn.py:29-33
        x12 = F.linear(h, sd[b + ".mlp.w12.weight"], sd[b + ".mlp.w12.bias"])
        x1, x2 = x12.chunk(2, dim=-1)
        h = F.linear(F.silu(x1) * x2, sd[b + ".mlp.w3.weight"], sd[b + ".mlp.w3.bias"])
    return x + sd[b + ".ls2.gamma"] * h, p


def dino_forward(sd: SD, x: torch.Tensor, *, depth: int, heads: int, patch: int = 14,
                 num_regs: int = 4, ffn: str = "mlp", want_attn: bool = False):
    """``forward_features(x)['x_prenorm']`` (vision_transformer.py:266-286) and, when
    asked, the per-block attention probabilities that ``get_last_self_attention``
    returns (:352-370).  The reference recomputes each block to get them; the
    values are identical to the ones of the ordinary forward."""
    t = dino_tokens(sd, x, patch, num_regs)
    attn: List[torch.Tensor] = []
    for i in range(depth):
        t, p = dino_block(sd, i, t, heads, ffn)
        if want_attn:
            attn.append(p)
    return (t, attn) if want_attn else t


# ----------------------------------------------------------------------------
# CLIP (CLIP-ES flavour used for the visual-text alignment)
# ----------------------------------------------------------------------------
def clip_pos_embed(sd: SD, grid_hw: Tuple[int, int]) -> torch.Tensor:
    """/root/reference/clip/model.py:99-115: bilinear (align_corners=False) resize
    of the patch part of the table, then a round trip through fp16."""
    emb = sd["visual.positional_embedding"]
    first, rest = emb[:1], emb[1:]
    n, d = rest.shape
    size = int(math.isqrt(n))
    rest = rest.permute(1, 0).reshape(1, d, size, size)
    rest = F.interpolate(rest, size=grid_hw, mode="bilinear", align_corners=False)
    rest = rest.reshape(d, -1).permute(1, 0)
    return torch.cat([first, rest], 0).half().float()


def _clip_block(sd: SD, prefix: str, x: torch.Tensor, heads: int, *, alpha_names: bool, causal: bool = False):
    """Pre-LN block with QuickGELU MLP.  /root/reference/clip/model.py:279-301
    (nn.MultiheadAttention) and /root/reference/alpha_clip/model.py:256-281."""
    wn, bn = (".attn.in_proj.weight", ".attn.in_proj.bias") if alpha_names else (".attn.in_proj_weight", ".attn.in_proj_bias")
    ln1 = _ln(x, sd, prefix + ".ln_1", 1e-5)
    a, p = _mha(ln1, sd[prefix + wn], sd[prefix + bn], sd[prefix + ".attn.out_proj.weight"],
                sd[prefix + ".attn.out_proj.bias"], heads, causal)
    x = x + a
    h = F.linear(_ln(x, sd, prefix + ".ln_2", 1e-5), sd[prefix + ".mlp.c_fc.weight"], sd[prefix + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    x = x + F.linear(h, sd[prefix + ".mlp.c_proj.weight"], sd[prefix + ".mlp.c_proj.bias"])
    return x, p, ln1


def clip_encode_image(sd: SD, img: torch.Tensor, *, layers: int, heads: int, patch: int):
    """``CLIP.encode_image(img, H, W)`` (/root/reference/clip/model.py:357-383,
    312-336): all blocks but the last; returns tokens [B, T, D] and the list of
    head-averaged attention maps [B, T, T] of those blocks."""
    B, _, H, W = img.shape
    x = F.conv2d(img, sd["visual.conv1.weight"], None, stride=patch)
    x = x.reshape(B, x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].expand(B, 1, -1), x], dim=1)
    x = x + clip_pos_embed(sd, (H // patch, W // patch))
    x = _ln(x, sd, "visual.ln_pre", 1e-5)
    attn = []
    for i in range(layers - 1):
        x, p, _ = _clip_block(sd, f"visual.transformer.resblocks.{i}", x, heads, alpha_names=False)
        attn.append(p.mean(dim=1))
    return x, attn


def clip_last_layer(sd: SD, x: torch.Tensor, text: torch.Tensor, *, layers: int, heads: int):
    """``CLIP.forward_last_layer`` (/root/reference/clip/model.py:501-524).  Returns
    softmaxed logits [B, n_text], the head-averaged attention of the last block and
    the ``ln_1`` output of that block (the GradCAM hook target)."""
    x, p, ln1 = _clip_block(sd, f"visual.transformer.resblocks.{layers - 1}", x, heads, alpha_names=False)
    y = _ln(x, sd, "visual.ln_post", 1e-5)
    f = y[:, 1:, :].mean(dim=1) @ sd["visual.proj"]
    f = f / f.norm(dim=1, keepdim=True)
    t = text / text.norm(dim=1, keepdim=True)
    logits = (sd["logit_scale"].exp() * f @ t.t()).softmax(dim=-1)
    return logits, p.mean(dim=1), ln1


def clip_encode_text(sd: SD, tokens: torch.Tensor, *, layers: int, heads: int) -> torch.Tensor:
    """``encode_text`` (/root/reference/clip/model.py:486-499 and
    /root/reference/alpha_clip/model.py:525-538): causal transformer over 77 tokens,
    feature of the highest token id (EOT) projected by ``text_projection``."""
    x = sd["token_embedding.weight"][tokens.long()] + sd["positional_embedding"]
    for i in range(layers):
        x, _, _ = _clip_block(sd, f"transformer.resblocks.{i}", x, heads, alpha_names=False, causal=True)
    x = _ln(x, sd, "ln_final", 1e-5)
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd["text_projection"]


def gradcam(sd: SD, x: torch.Tensor, text: torch.Tensor, *, layers: int, heads: int):
    """Softmax-GradCAM of the last block (/root/reference/pytorch_grad_cam/base_cam.py:62-118,
    126-164; grad_cam.py:16-23; /root/reference/mars/components/SoftmaxGradCAM.py:32-41):
    gradient of softmax logit 0 with respect to the ``ln_1`` output of the last block,
    channel weights = spatial mean of that gradient over the patch tokens, ReLU of the
    weighted activation sum, min-max twice.  Returns (cam [G, G], last attention
    [B, T, T], logits)."""
    prefix = f"visual.transformer.resblocks.{layers - 1}"
    x = x.detach()
    ln1 = _ln(x, sd, prefix + ".ln_1", 1e-5).detach().requires_grad_(True)
    a, p = _mha(ln1, sd[prefix + ".attn.in_proj_weight"], sd[prefix + ".attn.in_proj_bias"],
                sd[prefix + ".attn.out_proj.weight"], sd[prefix + ".attn.out_proj.bias"], heads)
    x1 = x + a
    h = F.linear(_ln(x1, sd, prefix + ".ln_2", 1e-5), sd[prefix + ".mlp.c_fc.weight"], sd[prefix + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    x2 = x1 + F.linear(h, sd[prefix + ".mlp.c_proj.weight"], sd[prefix + ".mlp.c_proj.bias"])
    y = _ln(x2, sd, "visual.ln_post", 1e-5)
    f = y[:, 1:, :].mean(dim=1) @ sd["visual.proj"]
    f = f / f.norm(dim=1, keepdim=True)
    t = text / text.norm(dim=1, keepdim=True)
    logits = (sd["logit_scale"].exp() * f @ t.t()).softmax(dim=-1)
    (grad,) = torch.autograd.grad(logits[:, 0].sum(), ln1)
    g = int(math.isqrt(x.shape[1] - 1))
    act = ln1.detach()[0, 1:, :]            # [P, D]
    w = grad[0, 1:, :].mean(dim=0)           # [D]
    cam = torch.relu((act * w).sum(dim=1)).reshape(g, g)
    cam = cam - cam.min()
    cam = cam / (1e-7 + cam.max())
    cam = cam - cam.min()                    # aggregate_multi_layers scales again
    cam = cam / (1e-7 + cam.max())
    return cam, p.mean(dim=1).detach(), logits.detach()


# ----------------------------------------------------------------------------
# AlphaCLIP
# ----------------------------------------------------------------------------
def alphaclip_visual(sd: SD, img: torch.Tensor, alpha: torch.Tensor, *, layers: int, heads: int, patch: int):
    """``model.visual(image, alpha)`` (/root/reference/alpha_clip/model.py:358-386):
    RGB and alpha patch embeddings summed, CLS + positional table, ln_pre, pre-LN
    blocks, ln_post of the CLS token, projection."""
    B = img.shape[0]
    x = F.conv2d(img, sd["visual.conv1.weight"], None, stride=patch)
    x = x + F.conv2d(alpha, sd["visual.conv1_alpha.weight"], None, stride=patch)
    x = x.reshape(B, x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].expand(B, 1, -1), x], dim=1)
    x = x + sd["visual.positional_embedding"]
    x = _ln(x, sd, "visual.ln_pre", 1e-5)
    for i in range(layers):
        x, _, _ = _clip_block(sd, f"visual.transformer.resblocks.{i}", x, heads, alpha_names=True)
    return _ln(x[:, 0, :], sd, "visual.ln_post", 1e-5) @ sd["visual.proj"]
