"""DINOv2 / CLIP / AlphaCLIP encoders on the HIP kernels, with the reference's method names.

The classes duck-type what ``BackboneLoader.load_backbone`` returns in the reference
(/root/reference/utils/backbone_loader.py:17-52):

* ``DinoVisionTransformer``: ``forward_features(x)['x_prenorm']``, ``get_last_self_attention(x)``,
  ``embed_dim``, ``patch_size``, ``num_register_tokens``, ``family``
  (/root/reference/dinov2/models/vision_transformer.py:266-286,352-370)
* ``CLIP``: ``encode_image(img, H, W)``, ``encode_text``, ``forward_last_layer``, ``visual.patch_size``, ``dtype``
  (/root/reference/clip/model.py:483-524)
* ``AlphaCLIP``: ``visual(image, alpha)``, ``encode_text`` (/root/reference/alpha_clip/model.py:358-386,525-538)

plus fused entry points the components use so that attention maps are averaged on the
fly instead of being materialised.  torch is used for allocation, parameter
preparation at load time and views; every per-call computation is a C-ABI kernel.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .hip import ACT_GELU, ACT_NONE, ACT_QUICKGELU
from .synth import ClipCfg, DinoCfg

SD = Dict[str, torch.Tensor]


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def flash_enabled() -> bool:
    return os.environ.get("MARS_FLASH", "1") != "0"


class Lin:
    """One nn.Linear: weight [N, Kpad] in the compute dtype (zero padded along K), bias f32."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], dtype, device):
        N, K = w.shape
        Kp = _pad_to(K, 64)
        wp = torch.zeros(N, Kp, dtype=torch.float32)
        wp[:, :K] = w.float()
        self.w = wp.to(device=device, dtype=dtype).contiguous()
        self.b = b.float().to(device).contiguous() if b is not None else None
        self.N, self.K, self.Kp = N, K, Kp


class Norm:
    def __init__(self, sd: SD, prefix: str, device, eps: float):
        self.w = sd[prefix + ".weight"].float().to(device).contiguous()
        self.b = sd[prefix + ".bias"].float().to(device).contiguous()
        self.eps = eps


class Block:
    pass


class Tower:
    """A stack of pre-LN transformer blocks with 64-wide heads over a f32 residual stream."""

    def __init__(self, D: int, heads: int, device, dtype):
        assert D == heads * 64, "the attention kernels are written for head_dim 64"
        self.D, self.H, self.device, self.dtype = D, heads, device, dtype
        self.blocks: List[Block] = []
        self._ws: Dict[Tuple[int, int, int], dict] = {}
        self.fp8_attention = False            # bf16 towers only: q, k, v and the probabilities as 8-bit floats
        self._qscale: Optional[torch.Tensor] = None

    # ---- workspaces (allocated once per shape; pads stay zero) ----
    def ws(self, B: int, T: int, hidden: int) -> dict:
        key = (B, T, hidden)
        w = self._ws.get(key)
        if w is None:
            dev, dtp, D, H = self.device, self.dtype, self.D, self.H
            M, Tp = B * T, _pad_to(T, 64)
            w = dict(
                xn=torch.zeros(M, D, dtype=dtp, device=dev),
                qkv=torch.zeros(M, 3 * D, dtype=dtp, device=dev),
                ao=torch.zeros(M, D, dtype=dtp, device=dev),
                hid=torch.zeros(M, hidden, dtype=dtp, device=dev),
                lse=torch.zeros(B * H, T, dtype=torch.float32, device=dev),
                Tp=Tp,
            )
            self._ws[key] = w
        return w

    def ws_unfused(self, w: dict, B: int, T: int):
        if "S" not in w:
            dev, dtp, H, Tp = self.device, self.dtype, self.H, w["Tp"]
            w["S"] = torch.zeros(B * H, T, Tp, dtype=torch.float32, device=dev)
            w["P"] = torch.zeros(B * H, T, Tp, dtype=dtp, device=dev)
            w["VT"] = torch.zeros(B * H, 64, Tp, dtype=dtp, device=dev)
        return w

    # ---- attention ----
    def uses_flash(self, need_probs: bool) -> bool:
        return self.dtype == torch.bfloat16 and flash_enabled() and not need_probs

    def query_scale(self, need_probs: bool) -> Optional[torch.Tensor]:
        """Per-column factors for the epilogue of the qkv product when the fused attention will consume it: the
        query columns come out multiplied by log2(e)/8 (rounded to bf16 once, from the f32 accumulator), so the
        attention kernels take the scores in the exp2 domain straight from the matrix pipe."""
        if not self.uses_flash(need_probs):
            return None
        if self._qscale is None:
            g = torch.ones(3 * self.D, dtype=torch.float32, device=self.device)
            g[: self.D] = ops.RAW_QUERY_SCALE_LOG2
            self._qscale = g
        return self._qscale

    def attention(self, w: dict, B: int, T: int, *, causal: bool, need_probs: bool, mean_acc=None):
        """qkv -> ao.  ``mean_acc`` = (acc [Pn,Pn] f32, off, Pn, scale): adds scale * sum over heads of the
        probabilities of image 0 restricted to tokens off..off+Pn.  On the fused path the queries in ``qkv``
        are the scaled ones of ``query_scale``; the unfused path takes raw queries."""
        D, H, Tp = self.D, self.H, w["Tp"]
        use_flash = self.uses_flash(need_probs)
        if use_flash and self.fp8_attention and not causal and mean_acc is None:
            ops.flash_attn_fp8(w["qkv"], w["ao"], B, T, H, w.setdefault("fp8", {}), score_scale_log2=1.0)
            return None
        if use_flash:
            ops.flash_attn(w["qkv"], w["ao"], w["lse"], B, T, H, causal, score_scale_log2=1.0)
            if mean_acc is not None:
                acc, off, Pn, scale = mean_acc
                ops.flash_attn_mean(w["qkv"], w["lse"], T, H, off, Pn, scale, acc, score_scale_log2=1.0)
            return None
        self.ws_unfused(w, B, T)
        qkv, S, P, VT = w["qkv"], w["S"], w["P"], w["VT"]
        # S = (q k^T) / 8 per (image, head)
        ops.gemm(qkv, qkv, S, T, T, 64, lda=3 * D, ldb=3 * D, ldc=Tp, a_off=0, b_off=D, nb0=B, nb1=H,
                 sA=(T * 3 * D, 64), sB=(T * 3 * D, 64), sC=(H * T * Tp, T * Tp), alpha=0.125)
        ops.softmax_rows(S, P, B * H, T, Tp, causal)
        ops.transpose_raw(qkv, 2 * D, 3 * D, (T * 3 * D, 64), VT, 0, Tp, (H * 64 * Tp, 64 * Tp), T, 64, B, H)
        ops.gemm(P, VT, w["ao"], T, 64, Tp, lda=Tp, ldb=Tp, ldc=D, nb0=B, nb1=H, sA=(H * T * Tp, T * Tp),
                 sB=(H * 64 * Tp, 64 * Tp), sC=(T * D, 64))
        if mean_acc is not None:
            acc, off, Pn, scale = mean_acc
            ops.attn_mean_accum(P, 0, Tp, T * Tp, H, off, Pn, scale, acc)
        return P

    # ---- one block ----
    def block_forward(self, x: torch.Tensor, blk: Block, *, causal=False, need_probs=False, mean_acc=None, keep=None):
        """x f32 [B,T,D], updated in place.  ``keep`` (dict) receives the intermediates the CAM backward needs:
        they are written to buffers of their own in this tower's workspace (valid until its next call with
        ``keep`` and the same shape), nothing is copied."""
        assert not blk.folded, "a block with folded LayerNorms runs in blocks_forward_folded"
        B, T, D = x.shape
        w = self.ws(B, T, blk.hidden_alloc)
        x2 = x.view(B * T, D)
        if keep is not None:
            assert blk.ffn == "mlp"
            k = w.get("keep")
            if k is None:
                z = lambda n, dt: torch.zeros(B * T, n, dtype=dt, device=self.device)     # noqa: E731
                k = w["keep"] = dict(ln1=z(D, self.dtype), ln2=z(D, self.dtype), x1=z(D, torch.float32), pre=z(blk.fc1.N, self.dtype))
            ops.layernorm(x2, blk.ln1.w, blk.ln1.b, k["ln1"], blk.ln1.eps)
            ops.linear(k["ln1"], blk.qkv.w, w["qkv"], bias=blk.qkv.b, gamma=self.query_scale(need_probs))
            P = self.attention(w, B, T, causal=causal, need_probs=need_probs, mean_acc=mean_acc)
            ops.linear(w["ao"], blk.proj.w, k["x1"], bias=blk.proj.b, gamma=blk.ls1, residual=x2)      # x1 = x + attn
            ops.layernorm(k["x1"], blk.ln2.w, blk.ln2.b, k["ln2"], blk.ln2.eps)
            ops.linear(k["ln2"], blk.fc1.w, k["pre"], bias=blk.fc1.b)                                   # before the activation
            ops.linear(k["ln2"], blk.fc1.w, w["hid"], bias=blk.fc1.b, act=blk.act)
            ops.linear(w["hid"], blk.fc2.w, x2, bias=blk.fc2.b, gamma=blk.ls2, residual=k["x1"])
            keep.update(ln1=k["ln1"], qkv=w["qkv"], ao=w["ao"], x1=k["x1"], ln2=k["ln2"], pre=k["pre"])
            return P
        ops.layernorm(x2, blk.ln1.w, blk.ln1.b, w["xn"], blk.ln1.eps)
        ops.linear(w["xn"], blk.qkv.w, w["qkv"], bias=blk.qkv.b, gamma=self.query_scale(need_probs))
        P = self.attention(w, B, T, causal=causal, need_probs=need_probs, mean_acc=mean_acc)
        ops.linear(w["ao"], blk.proj.w, x2, bias=blk.proj.b, gamma=blk.ls1, residual=x2)
        ops.layernorm(x2, blk.ln2.w, blk.ln2.b, w["xn"], blk.ln2.eps)
        if blk.ffn == "swiglu":
            ops.linear(w["xn"], blk.fc1.w, w["hid"], bias=blk.fc1.b)          # [M, 2*hidden]
            g = w.get("gate")
            if g is None:
                g = w["gate"] = torch.zeros(B * T, blk.fc2.Kp, dtype=self.dtype, device=self.device)
            ops.swiglu(w["hid"], g)                                            # pad columns stay zero
            ops.linear(g, blk.fc2.w, x2, bias=blk.fc2.b, gamma=blk.ls2, residual=x2)
        else:
            ops.linear(w["xn"], blk.fc1.w, w["hid"], bias=blk.fc1.b, act=blk.act)
            ops.linear(w["hid"], blk.fc2.w, x2, bias=blk.fc2.b, gamma=blk.ls2, residual=x2)
        return P


def _blocks_forward_folded(self, x: torch.Tensor, blocks: List[Block], on_block=None, watch=None) -> None:
    """``block_forward`` over a run of blocks built with ``fold=True``: no LayerNorm pass reads the stream.
    ``on_block(i)`` is called when block i has been queued (a caller with work for other streams can issue it
    between the blocks instead of behind all of them).
    Inside the run the f32 stream lives in two bf16 planes, hi = bf16(x) and lo = bf16(x - hi).  The products
    that add to it (attention projection, second MLP layer) read and write the planes and leave the row sums;
    the products that would have read normalised rows (qkv, first MLP layer) take the hi plane and apply mean
    and rstd to their accumulators.  x f32 [B,T,D], updated in place (written by the last product of the run).
    ``watch`` = (f32 [1] on the device, limit): raised to the largest |mean| * rstd above the limit that a row of
    the stream shows at any LayerNorm of the run (the accuracy the folded form gives up on that row)."""
    B, T, D = x.shape
    M = B * T
    w = self.ws(B, T, blocks[0].hidden_alloc)
    f = w.get("fold")
    if f is None:
        f = w["fold"] = dict(hi=torch.zeros(M, D, dtype=self.dtype, device=self.device),
                             lo=torch.zeros(M, D, dtype=self.dtype, device=self.device),
                             part=torch.zeros(M, D // 64, 2, dtype=torch.float32, device=self.device),
                             stat=torch.zeros(M, 2, dtype=torch.float32, device=self.device))
    hi, lo, part, stat = f["hi"], f["lo"], f["part"], f["stat"]
    x2 = x.view(M, D)
    wt, wl = watch if watch is not None else (None, 0.0)
    ops.row_stats(x2, blocks[0].ln1.eps, hi, lo, stat, wt, wl)
    for i, blk in enumerate(blocks):
        assert blk.folded
        ops.linear(hi, blk.qkv.w, w["qkv"], bias=blk.qkv.b, gamma=self.query_scale(False), ln_in=(stat, blk.qkv_colsum))
        self.attention(w, B, T, causal=False, need_probs=False)
        ops.linear(w["ao"], blk.proj.w, x2, bias=blk.proj.b, gamma=blk.ls1, ln_out=(hi, lo, part))
        ops.ln_finish(part, M, D, blk.ln2.eps, stat, wt, wl)
        ops.linear(hi, blk.fc1.w, w["hid"], bias=blk.fc1.b, act=blk.act, ln_in=(stat, blk.fc1_colsum))
        if i + 1 < len(blocks):
            ops.linear(w["hid"], blk.fc2.w, x2, bias=blk.fc2.b, gamma=blk.ls2, ln_out=(hi, lo, part))
            ops.ln_finish(part, M, D, blocks[i + 1].ln1.eps, stat, wt, wl)
        else:
            ops.linear(w["hid"], blk.fc2.w, x2, bias=blk.fc2.b, gamma=blk.ls2, ln_out=(hi, lo, None))
        if on_block is not None:
            on_block(i)


Tower.blocks_forward_folded = _blocks_forward_folded


def _first_token_block(self, x: torch.Tensor, blk: Block) -> torch.Tensor:
    """The block's output for token 0 of every sequence, [B, D] f32 (x is left as it is).  A tower whose head
    reads only the class token needs nothing else from its last block: keys and values still come from every
    token, but the query, the attention output, both residual adds and the MLP are formed for one row per
    sequence (same arithmetic per row as ``block_forward``; the attention is the unfused form)."""
    assert blk.ffn == "mlp" and not blk.folded
    B, T, D = x.shape
    H, dtp, dev = self.H, self.dtype, self.device
    w = self.ws(B, T, blk.hidden_alloc)
    Tp = w["Tp"]
    f = w.get("first")
    if f is None:
        f = w["first"] = dict(
            q=torch.zeros(B, D, dtype=dtp, device=dev), S=torch.zeros(B * H, 1, Tp, dtype=torch.float32, device=dev),
            P=torch.zeros(B * H, 1, Tp, dtype=dtp, device=dev), VT=torch.zeros(B * H, 64, Tp, dtype=dtp, device=dev),
            ao=torch.zeros(B, D, dtype=dtp, device=dev), xn=torch.zeros(B, D, dtype=dtp, device=dev),
            hid=torch.zeros(B, blk.hidden_alloc, dtype=dtp, device=dev))
    x2 = x.view(B * T, D)
    ops.layernorm(x2, blk.ln1.w, blk.ln1.b, w["xn"], blk.ln1.eps)
    qkv = w["qkv"]
    ops.linear(w["xn"], blk.qkv.w[D:], qkv[:, D:], bias=blk.qkv.b[D:])              # keys and values of every token
    ops.linear(w["xn"].view(B, T, D)[:, 0], blk.qkv.w[:D], f["q"], bias=blk.qkv.b[:D])   # queries of token 0
    ops.gemm(f["q"], qkv, f["S"], 1, T, 64, lda=D, ldb=3 * D, ldc=Tp, b_off=D, nb0=B, nb1=H, sA=(D, 64),
             sB=(T * 3 * D, 64), sC=(H * Tp, Tp), alpha=0.125)
    ops.softmax_bias_rows(f["S"], f["P"], B * H, T)
    ops.transpose_raw(qkv, 2 * D, 3 * D, (T * 3 * D, 64), f["VT"], 0, Tp, (H * 64 * Tp, 64 * Tp), T, 64, B, H)
    ops.gemm(f["P"], f["VT"], f["ao"], 1, 64, Tp, lda=Tp, ldb=Tp, ldc=D, nb0=B, nb1=H, sA=(H * Tp, Tp),
             sB=(H * 64 * Tp, 64 * Tp), sC=(D, 64))
    xc = x[:, 0, :].clone()
    ops.linear(f["ao"], blk.proj.w, xc, bias=blk.proj.b, gamma=blk.ls1, residual=xc)
    ops.layernorm(xc, blk.ln2.w, blk.ln2.b, f["xn"], blk.ln2.eps)
    ops.linear(f["xn"], blk.fc1.w, f["hid"][:, : blk.fc1.N], bias=blk.fc1.b, act=blk.act)
    ops.linear(f["hid"], blk.fc2.w, xc, bias=blk.fc2.b, gamma=blk.ls2, residual=xc)
    return xc


Tower.block_forward_first_token = _first_token_block


def _fold_norm(w: torch.Tensor, b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor):
    """Linear(LayerNorm(x)) = rstd * (x . (w*ln_w)^T - mean * rowsum(w*ln_w)) + (b + w . ln_b): the weights and
    bias of the product that takes x itself."""
    w, ln_w, ln_b = w.float(), ln_w.float(), ln_b.float()
    return w * ln_w[None, :], b.float() + w @ ln_b


def _make_block(sd: SD, names: dict, D: int, device, dtype, eps: float, act: int, ffn: str = "mlp", fold: bool = False) -> Block:
    """``fold``: the two LayerNorms are folded into the qkv and fc1 products (``Tower.blocks_forward_folded``
    is then the only way to run the block)."""
    b = Block()
    b.ln1 = Norm(sd, names["ln1"], device, eps)
    b.ln2 = Norm(sd, names["ln2"], device, eps)
    b.folded = fold
    b.names = names
    if fold:
        assert ffn == "mlp" and dtype == torch.bfloat16
        b.qkv = Lin(*_fold_norm(sd[names["qkv_w"]], sd[names["qkv_b"]], sd[names["ln1"] + ".weight"], sd[names["ln1"] + ".bias"]), dtype, device)
        b.fc1 = Lin(*_fold_norm(sd[names["fc1"] + ".weight"], sd[names["fc1"] + ".bias"], sd[names["ln2"] + ".weight"], sd[names["ln2"] + ".bias"]), dtype, device)
        # sums of the weights as the matrix pipe sees them (rounded), in f32
        b.qkv_colsum = b.qkv.w.float().sum(dim=1).contiguous()
        b.fc1_colsum = b.fc1.w.float().sum(dim=1).contiguous()
    else:
        b.qkv = Lin(sd[names["qkv_w"]], sd[names["qkv_b"]], dtype, device)
        b.fc1 = Lin(sd[names["fc1"] + ".weight"], sd[names["fc1"] + ".bias"], dtype, device)
    b.proj = Lin(sd[names["proj"] + ".weight"], sd[names["proj"] + ".bias"], dtype, device)
    b.fc2 = Lin(sd[names["fc2"] + ".weight"], sd[names["fc2"] + ".bias"], dtype, device)
    b.ls1 = sd[names["ls1"]].float().to(device).contiguous() if names.get("ls1") else None
    b.ls2 = sd[names["ls2"]].float().to(device).contiguous() if names.get("ls2") else None
    b.act, b.ffn = act, ffn
    b.hidden = b.fc2.K
    b.hidden_alloc = b.fc1.N
    return b


# ======================================================================================
# DINOv2
# ======================================================================================
class DinoVisionTransformer:
    family = "vits_dino2"

    def __init__(self, cfg: DinoCfg, sd: SD, device="cuda", dtype=torch.float32):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.embed_dim, self.patch_size, self.num_register_tokens = cfg.embed_dim, cfg.patch_size, cfg.num_regs
        self.num_heads, self.n_blocks = cfg.num_heads, cfg.depth
        D, p = cfg.embed_dim, cfg.patch_size
        self.tower = Tower(D, cfg.num_heads, self.device, dtype)
        for i in range(cfg.depth):
            pre = f"blocks.{i}"
            names = dict(ln1=pre + ".norm1", ln2=pre + ".norm2", qkv_w=pre + ".attn.qkv.weight", qkv_b=pre + ".attn.qkv.bias",
                         proj=pre + ".attn.proj", ls1=pre + ".ls1.gamma", ls2=pre + ".ls2.gamma")
            if cfg.ffn == "mlp":
                names.update(fc1=pre + ".mlp.fc1", fc2=pre + ".mlp.fc2")
            else:
                names.update(fc1=pre + ".mlp.w12", fc2=pre + ".mlp.w3")
            self.tower.blocks.append(_make_block(sd, names, D, self.device, dtype, 1e-6, ACT_GELU, cfg.ffn))
        self.pe = Lin(sd["patch_embed.proj.weight"].reshape(D, 3 * p * p), sd["patch_embed.proj.bias"], dtype, self.device)
        self.cls = sd["cls_token"].reshape(D).float().to(self.device)
        pos = sd["pos_embed"].reshape(-1, D).float().to(self.device).contiguous()
        self.pos0, self.pos_patch = pos[0].contiguous(), pos[1:].contiguous()
        self.regs = sd["register_tokens"].reshape(-1, D).float().to(self.device).contiguous() if cfg.num_regs else None
        self.norm = Norm(sd, "norm", self.device, 1e-6)
        self._cols: Dict[Tuple[int, int], torch.Tensor] = {}
        self._pos_cache: Dict[Tuple[int, int], torch.Tensor] = {}

    def _pos_for(self, gh: int, gw: int) -> torch.Tensor:
        """interpolate_pos_encoding (vision_transformer.py:188-221) for the patch part: the table as loaded
        for a square grid of the constructor's size, otherwise resized bicubically with the reference's
        scale factors (g + 0.1) / M.  Input independent, so prepared once per grid."""
        M = self.cfg.grid
        if gh == M and gw == M:
            return self.pos_patch
        key = (gh, gw)
        if key not in self._pos_cache:
            t = self.pos_patch.reshape(1, M, M, -1).permute(0, 3, 1, 2)
            r = F.interpolate(t, mode="bicubic", antialias=False, scale_factor=(float(gh + 0.1) / M, float(gw + 0.1) / M))
            assert tuple(r.shape[-2:]) == (gh, gw)
            self._pos_cache[key] = r.permute(0, 2, 3, 1).reshape(gh * gw, -1).contiguous()
        return self._pos_cache[key]

    def _tokens(self, x: torch.Tensor, mean=None, std=None) -> torch.Tensor:
        """prepare_tokens_with_masks (vision_transformer.py:223-243); inputs of another size than the
        constructor's (multiples of the patch size) get the resized positional table."""
        B, C, H, W = x.shape
        cfg = self.cfg
        p = cfg.patch_size
        assert C == 3 and H % p == 0 and W % p == 0 and H >= p and W >= p, "input sides must be multiples of the patch size"
        gh, gw = H // p, W // p
        Pn, D, R = gh * gw, cfg.embed_dim, cfg.num_regs
        T = 1 + R + Pn
        cols = self._cols.get((B, Pn))
        if cols is None:
            cols = self._cols[(B, Pn)] = torch.zeros(B * Pn, self.pe.Kp, dtype=self.dtype, device=self.device)
        ops.im2col(x.contiguous().float(), cfg.patch_size, cols, mean=mean, std=std)
        X = torch.empty(B, T, D, dtype=torch.float32, device=self.device)
        ops.gemm(cols, self.pe.w, X, Pn, D, self.pe.Kp, lda=self.pe.Kp, ldb=self.pe.Kp, ldc=D, c_off=(1 + R) * D, nb0=B,
                 sA=(Pn * self.pe.Kp, 0), sC=(T * D, 0), bias=self.pe.b, residual=self._pos_for(gh, gw), ldr=D)
        ops.write_prefix_tokens(X, self.cls, self.pos0, self.regs)
        return X

    def forward_tokens(self, x, mean=None, std=None, mean_attn: Optional[Tuple[torch.Tensor, int]] = None):
        """Run every block.  ``mean_attn`` = (acc [P,P] zeroed f32, last_n): accumulates the mean over the
        last n blocks and all heads of the patch-to-patch attention of image 0."""
        X = self._tokens(x, mean, std)
        L = self.cfg.depth
        for i, blk in enumerate(self.tower.blocks):
            ma = None
            if mean_attn is not None and i >= L - mean_attn[1]:
                n = min(mean_attn[1], L)
                ma = (mean_attn[0], 1 + self.cfg.num_regs, X.shape[1] - 1 - self.cfg.num_regs, 1.0 / (n * self.cfg.num_heads))
            self.tower.block_forward(X, blk, mean_acc=ma)
        return X

    def forward_features(self, x: torch.Tensor, masks=None) -> dict:
        assert masks is None
        X = self.forward_tokens(x)
        B, T, D = X.shape
        xn = torch.empty_like(X)
        ops.layernorm(X.view(B * T, D), self.norm.w, self.norm.b, xn.view(B * T, D), self.norm.eps)
        R = self.num_register_tokens
        return {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:R + 1], "x_norm_patchtokens": xn[:, R + 1:],
                "x_prenorm": X, "masks": masks}

    def get_last_self_attention(self, x: torch.Tensor, masks=None):
        """Tuple of per-block attention probabilities [B, heads, T, T] (vision_transformer.py:352-370).
        Materialises them, as the reference does; the components use ``forward_tokens(mean_attn=...)``."""
        X = self._tokens(x)
        B, T, _ = X.shape
        out = []
        for blk in self.tower.blocks:
            P = self.tower.block_forward(X, blk, need_probs=True)
            out.append(P.view(B, self.num_heads, T, -1)[..., :T].float().clone())
        return tuple(out)


# ======================================================================================
# CLIP towers (CLIP-ES flavour and AlphaCLIP)
# ======================================================================================
class _TextTower:
    def __init__(self, cfg: ClipCfg, sd: SD, device, dtype):
        self.cfg, self.device, self.dtype = cfg, device, dtype
        W = cfg.transformer_width
        self.tower = Tower(W, cfg.transformer_heads, device, dtype)
        for i in range(cfg.transformer_layers):
            pre = f"transformer.resblocks.{i}"
            names = dict(ln1=pre + ".ln_1", ln2=pre + ".ln_2", qkv_w=pre + ".attn.in_proj_weight", qkv_b=pre + ".attn.in_proj_bias",
                         proj=pre + ".attn.out_proj", fc1=pre + ".mlp.c_fc", fc2=pre + ".mlp.c_proj")
            self.tower.blocks.append(_make_block(sd, names, W, device, dtype, 1e-5, ACT_QUICKGELU))
        self.tok = sd["token_embedding.weight"].float().to(device)
        self.pos = sd["positional_embedding"].float().to(device)
        self.ln_final = Norm(sd, "ln_final", device, 1e-5)
        self.proj_t = Lin(sd["text_projection"].t().contiguous(), None, torch.float32, device)  # [E, W]

    def encode_text(self, tokens: torch.Tensor) -> torch.Tensor:
        """encode_text (clip/model.py:486-499): causal transformer, EOT feature, text_projection."""
        tokens_cpu = tokens.cpu()
        n, L = tokens_cpu.shape
        assert int(tokens_cpu.min()) >= 0 and int(tokens_cpu.max()) < self.tok.shape[0], "token id outside the vocabulary"
        idx = ops.upload_small(tokens_cpu.int().contiguous(), self.device)      # queued, not waited for (ops._PinnedRing)
        X = torch.empty(n, L, self.tok.shape[1], dtype=torch.float32, device=self.device)
        ops.embed_tokens(self.tok, idx, self.pos[:L].contiguous(), X)           # token_embedding(text) + positional_embedding
        for blk in self.tower.blocks:
            self.tower.block_forward(X, blk, causal=True)
        eot = tokens_cpu.argmax(dim=-1)
        rows = ops.upload_small((torch.arange(n) * L + eot).int().contiguous(), self.device)
        W = self.cfg.transformer_width
        sel = torch.empty(n, W, dtype=torch.float32, device=self.device)
        ops.gather_rows(X.view(n * L, W), rows, sel)
        y = torch.empty(n, W, dtype=torch.float32, device=self.device)
        ops.layernorm(sel, self.ln_final.w, self.ln_final.b, y, self.ln_final.eps)
        out = torch.empty(n, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        ops.linear(y, self.proj_t.w, out)
        return out


class _Visual:
    """Attribute bag mirroring ``model.visual`` of the reference for the names the callers touch."""
    pass


class CLIP(_TextTower):
    """CLIP-ES style CLIP used for the visual-text alignment."""

    def __init__(self, cfg: ClipCfg, sd: SD, device="cuda", dtype=torch.float32):
        device = torch.device(device)
        super().__init__(cfg, sd, device, dtype)
        W, p = cfg.vision_width, cfg.vision_patch_size
        self.vtower = Tower(W, cfg.vision_heads, device, dtype)
        self.vtower32 = self.vtower if dtype == torch.float32 else Tower(W, cfg.vision_heads, device, torch.float32)
        for i in range(cfg.vision_layers):
            pre = f"visual.transformer.resblocks.{i}"
            names = dict(ln1=pre + ".ln_1", ln2=pre + ".ln_2", qkv_w=pre + ".attn.in_proj_weight", qkv_b=pre + ".attn.in_proj_bias",
                         proj=pre + ".attn.out_proj", fc1=pre + ".mlp.c_fc", fc2=pre + ".mlp.c_proj")
            last = i == cfg.vision_layers - 1
            # the last block (forward_last_layer + GradCAM) always runs in f32
            self.vtower.blocks.append(_make_block(sd, names, W, device, torch.float32 if last else dtype, 1e-5, ACT_QUICKGELU))
        lp = f"visual.transformer.resblocks.{cfg.vision_layers - 1}"
        # transposed f32 copies of the last block's weights: backward products as TN GEMMs
        self.last_T = dict(
            in_proj=sd[lp + ".attn.in_proj_weight"].float().t().contiguous().to(device),     # [D, 3D]
            out_proj=sd[lp + ".attn.out_proj.weight"].float().t().contiguous().to(device),   # [D, D]
            c_fc=sd[lp + ".mlp.c_fc.weight"].float().t().contiguous().to(device),            # [D, 4D]
            c_proj=sd[lp + ".mlp.c_proj.weight"].float().t().contiguous().to(device),        # [4D, D]
        )
        self.conv1 = Lin(sd["visual.conv1.weight"].reshape(W, 3 * p * p), None, dtype, device)
        self.class_embedding = sd["visual.class_embedding"].float().to(device)
        self.pos_table = sd["visual.positional_embedding"].float()  # CPU: resized per grid on demand
        self.ln_pre = Norm(sd, "visual.ln_pre", device, 1e-5)
        self.ln_post = Norm(sd, "visual.ln_post", device, 1e-5)
        self.vproj = sd["visual.proj"].float().to(device).contiguous()  # [W, E]
        self.logit_scale_exp = float(sd["logit_scale"].float().exp())
        self._pos_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        self._cols: Dict[int, torch.Tensor] = {}
        self._cam_bufs: Dict[int, dict] = {}
        self.visual = _Visual()
        self.visual.patch_size = p
        self.visual.input_resolution = cfg.image_resolution
        self.visual.output_dim = cfg.embed_dim

    def _pos(self, gh: int, gw: int) -> torch.Tensor:
        """upsample_pos_emb (clip/model.py:99-115): bilinear resize of the patch table and a round trip
        through fp16; input independent, so prepared once per grid."""
        key = (gh, gw)
        if key not in self._pos_cache:
            emb = self.pos_table
            first, rest = emb[:1], emb[1:]
            n, d = rest.shape
            s = int(math.isqrt(n))
            rest = rest.permute(1, 0).reshape(1, d, s, s)
            rest = F.interpolate(rest, size=(gh, gw), mode="bilinear", align_corners=False)
            rest = rest.reshape(d, -1).permute(1, 0)
            self._pos_cache[key] = torch.cat([first, rest], 0).half().float().to(self.device).contiguous()
        return self._pos_cache[key]

    def _vision_tokens(self, image: torch.Tensor, mean=None, std=None) -> torch.Tensor:
        B, C, H, W = image.shape
        p, D = self.cfg.vision_patch_size, self.cfg.vision_width
        gh, gw = H // p, W // p
        Pn, T = gh * gw, gh * gw + 1
        pos = self._pos(gh, gw)
        key = (B, Pn)
        cols = self._cols.get(key)
        if cols is None:
            cols = self._cols[key] = torch.zeros(B * Pn, self.conv1.Kp, dtype=self.dtype, device=self.device)
        ops.im2col(image.contiguous().float(), p, cols, mean=mean, std=std)
        X = torch.empty(B, T, D, dtype=torch.float32, device=self.device)
        ops.gemm(cols, self.conv1.w, X, Pn, D, self.conv1.Kp, lda=self.conv1.Kp, ldb=self.conv1.Kp, ldc=D, c_off=D, nb0=B,
                 sA=(Pn * self.conv1.Kp, 0), sC=(T * D, 0), residual=pos, r_off=D, ldr=D)
        ops.write_prefix_tokens(X, self.class_embedding, pos[0], None)
        X2 = X.view(B * T, D)
        ops.layernorm(X2, self.ln_pre.w, self.ln_pre.b, X2, self.ln_pre.eps)
        return X

    def encode_image_tokens(self, image, mean=None, std=None, mean_attn=None, need_probs=False):
        """All blocks but the last.  Returns (X [B,T,D] f32, list of materialised probabilities or None)."""
        X = self._vision_tokens(image, mean, std)
        B, T, _ = X.shape
        L = self.cfg.vision_layers
        probs = [] if need_probs else None
        for i in range(L - 1):
            ma = None
            if mean_attn is not None and i >= L - mean_attn[1]:
                ma = (mean_attn[0], 1, T - 1, 1.0 / (min(mean_attn[1], L) * self.cfg.vision_heads))
            P = self.vtower.block_forward(X, self.vtower.blocks[i], need_probs=need_probs, mean_acc=ma)
            if need_probs:
                probs.append(P.view(B, self.cfg.vision_heads, T, -1)[..., :T].float().mean(dim=1))
        return X, probs

    def encode_image(self, image, H, W, extract=None):
        """encode_image(img, H, W) (clip/model.py:483-484,357-383): tokens in LND layout and the list of
        head-averaged attention maps of every block but the last."""
        X, probs = self.encode_image_tokens(image, need_probs=True)
        return X.permute(1, 0, 2), probs

    def forward_last_layer(self, image_features, text_features):
        """forward_last_layer (clip/model.py:501-524) on the LND tokens ``encode_image`` returned: last
        block in f32, ln_post, patch-token mean, projection, softmax over the {fg, bg} text pair.
        Returns (logits [B,2], head-averaged attention [B,T,T])."""
        X = image_features.permute(1, 0, 2).contiguous().float()
        B, T, _ = X.shape
        logits, P = self._last_layer(X, text_features)
        return logits, P.view(B, self.cfg.vision_heads, T, -1)[..., :T].mean(dim=1)

    def _last_layer(self, X, text_features, mean_attn=None, keep=None):
        """Same on batch-major tokens X f32 [B,T,D] (modified in place).  Returns (logits [B,2], the block's
        probabilities [B*H, T, Tp]); the head average is left to the one caller that hands it out."""
        B, T, D = X.shape
        blk = self.vtower.blocks[-1]
        P = self.vtower32.block_forward(X, blk, need_probs=True, mean_acc=mean_attn, keep=keep)
        c = self._cam_ws(T)
        ops.layernorm(X.view(B * T, D), self.ln_post.w, self.ln_post.b, c["y"], self.ln_post.eps)
        logits = torch.empty(B, 2, dtype=torch.float32, device=self.device)
        assert B == 1 and text_features.shape[0] == 2, "the CAM head is written for one image and the {fg, bg} text pair"
        ops.col_mean(c["y"], 1, T, c["f"])
        ops.cam_head(c["f"], self.vproj, text_features.float().contiguous(), self.logit_scale_exp, logits, c["df"])
        if keep is not None:
            keep.update(x2=X.view(B * T, D), df=c["df"], P=P, T=T)
        return logits, P

    def _cam_ws(self, T: int) -> dict:
        """Buffers of the CAM head and its backward for a T-token image, allocated (and zeroed) once: every
        kernel below writes the valid [.., T] part of its output only, so the pad columns that later products
        contract over stay zero from one call to the next."""
        c = self._cam_bufs.get(T)
        if c is None:
            D, H, Tp = self.cfg.vision_width, self.cfg.vision_heads, _pad_to(T, 64)
            z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.device)     # noqa: E731
            c = self._cam_bufs[T] = dict(
                y=z(T, D), f=z(D), df=z(D), dx2=z(T, D), dh=z(T, 4 * D), dm=z(T, D), dx1=z(T, D), do=z(T, D),
                dP=z(H, T, Tp), dS=z(H, T, Tp), dqkv=z(T, 3 * D), PT=z(H, Tp, Tp), dOT=z(H, 64, Tp), KT=z(H, 64, Tp),
                dST=z(H, Tp, Tp), QT=z(H, 64, Tp), grad=z(T, D), wgt=z(D))
        return c

    def gradcam(self, X: torch.Tensor, text_features: torch.Tensor, mean_attn=None):
        """Softmax-GradCAM of the last block w.r.t. its ln_1 output (pytorch_grad_cam/base_cam.py:62-118,
        grad_cam.py:16-23) by explicit backward kernels.  X f32 [1,T,D] = output of encode_image_tokens.
        Returns (cam [G,G] min-max scaled twice, logits [1,2])."""
        keep: dict = {}
        logits, _ = self._last_layer(X, text_features, mean_attn=mean_attn, keep=keep)      # X is consumed (updated in place)
        T, D, H = keep["T"], self.cfg.vision_width, self.cfg.vision_heads
        blk = self.vtower.blocks[-1]
        Tp = _pad_to(T, 64)
        c = self._cam_ws(T)
        # ln_post backward: dy = df/(T-1) on patch rows, 0 on the class row
        dx2 = c["dx2"]
        ops.layernorm_bwd(keep["x2"], self.ln_post.w, keep["df"], 0, 1.0 / (T - 1), 1, None, dx2, self.ln_post.eps)
        # MLP backward
        dh = c["dh"]
        ops.linear(dx2, self.last_T["c_proj"], dh)                      # dh = dx2 @ W_proj
        ops.quickgelu_bwd(dh, keep["pre"], dh)
        dm = c["dm"]
        ops.linear(dh, self.last_T["c_fc"], dm)                         # dm = dpre @ W_fc
        dx1 = c["dx1"]
        ops.layernorm_bwd(keep["x1"], blk.ln2.w, dm, D, 1.0, 0, dx2, dx1, blk.ln2.eps)
        # attention backward
        do = c["do"]
        ops.linear(dx1, self.last_T["out_proj"], do)                    # dO = dx1 @ W_out
        qkv, P = keep["qkv"], keep["P"]                                 # [T,3D], [H,T,Tp]
        dP = c["dP"]
        ops.gemm(do, qkv, dP, T, T, 64, lda=D, ldb=3 * D, ldc=Tp, b_off=2 * D, nb0=1, nb1=H, sA=(0, 64), sB=(0, 64),
                 sC=(0, T * Tp))                                        # dP = dO_h V_h^T
        dS = c["dS"]
        ops.softmax_bwd(P, dP, dS, Tp, H, T, 0.125)                     # includes the 1/8 of q.k/8
        dqkv, PT = c["dqkv"], c["PT"]
        ops.transpose_raw(P, 0, Tp, (0, T * Tp), PT, 0, Tp, (0, Tp * Tp), T, T, 1, H)
        dOT = c["dOT"]
        ops.transpose_raw(do, 0, D, (0, 64), dOT, 0, Tp, (0, 64 * Tp), T, 64, 1, H)
        # dV_h[key, d] = sum_q P[q,key] dO[q,d]
        ops.gemm(PT, dOT, dqkv, T, 64, Tp, lda=Tp, ldb=Tp, ldc=3 * D, c_off=2 * D, nb0=1, nb1=H, sA=(0, Tp * Tp),
                 sB=(0, 64 * Tp), sC=(0, 64))
        KT_ = c["KT"]
        ops.transpose_raw(qkv, D, 3 * D, (0, 64), KT_, 0, Tp, (0, 64 * Tp), T, 64, 1, H)
        # dQ_h = dS_h K_h
        ops.gemm(dS, KT_, dqkv, T, 64, Tp, lda=Tp, ldb=Tp, ldc=3 * D, c_off=0, nb0=1, nb1=H, sA=(0, T * Tp),
                 sB=(0, 64 * Tp), sC=(0, 64))
        dST = c["dST"]
        ops.transpose_raw(dS, 0, Tp, (0, T * Tp), dST, 0, Tp, (0, Tp * Tp), T, T, 1, H)
        QT = c["QT"]
        ops.transpose_raw(qkv, 0, 3 * D, (0, 64), QT, 0, Tp, (0, 64 * Tp), T, 64, 1, H)
        # dK_h = dS_h^T Q_h
        ops.gemm(dST, QT, dqkv, T, 64, Tp, lda=Tp, ldb=Tp, ldc=3 * D, c_off=D, nb0=1, nb1=H, sA=(0, Tp * Tp),
                 sB=(0, 64 * Tp), sC=(0, 64))
        grad = c["grad"]
        ops.linear(dqkv, self.last_T["in_proj"], grad)                  # d ln1 = dqkv @ W_in
        # channel weights = mean gradient over the patch tokens; cam = relu(sum_c w_c act_c)
        wgt = c["wgt"]
        ops.col_mean(grad, 1, T, wgt)
        cam = torch.empty(T - 1, dtype=torch.float32, device=self.device)
        ops.cam_rows(keep["ln1"], wgt, 1, T - 1, cam)
        ops.minmax(cam, cam)
        ops.minmax(cam, cam)
        g = int(math.isqrt(T - 1))
        return cam.view(g, g), logits, grad


class AlphaCLIP(_TextTower):
    def __init__(self, cfg: ClipCfg, sd: SD, device="cuda", dtype=torch.float32, fold_ln: Optional[bool] = None):
        device = torch.device(device)
        super().__init__(cfg, sd, device, dtype)
        W, p = cfg.vision_width, cfg.vision_patch_size
        self.vtower = Tower(W, cfg.vision_heads, device, dtype)
        if fold_ln is None:
            fold_ln = os.environ.get("MARS_FOLD_LN", "1") != "0"
        self.fold_ln = bool(fold_ln) and dtype == torch.bfloat16 and W % 64 == 0
        for i in range(cfg.vision_layers):
            pre = f"visual.transformer.resblocks.{i}"
            names = dict(ln1=pre + ".ln_1", ln2=pre + ".ln_2", qkv_w=pre + ".attn.in_proj.weight", qkv_b=pre + ".attn.in_proj.bias",
                         proj=pre + ".attn.out_proj", fc1=pre + ".mlp.c_fc", fc2=pre + ".mlp.c_proj")
            # every block but the last (whose first-token form normalises single rows) has its LayerNorms
            # folded into the products around them; MARS_FOLD_LN=0 keeps the stand-alone passes
            fold = self.fold_ln and i + 1 < cfg.vision_layers
            self.vtower.blocks.append(_make_block(sd, names, W, device, dtype, 1e-5, ACT_QUICKGELU, fold=fold))
        # RGB and alpha patch embeddings share one GEMM: K = 3*p*p (image) + p*p (alpha)
        w_rgba = torch.cat([sd["visual.conv1.weight"].reshape(W, 3 * p * p), sd["visual.conv1_alpha.weight"].reshape(W, p * p)], dim=1)
        self.conv_rgb = Lin(sd["visual.conv1.weight"].reshape(W, 3 * p * p), None, dtype, device)
        self.conv_alpha = Lin(sd["visual.conv1_alpha.weight"].reshape(W, p * p), None, dtype, device)
        self.conv_rgba = Lin(w_rgba, None, dtype, device)
        self.class_embedding = sd["visual.class_embedding"].float().to(device)
        self.vpos = sd["visual.positional_embedding"].float().to(device).contiguous()
        self.ln_pre = Norm(sd, "visual.ln_pre", device, 1e-5)
        self.ln_post = Norm(sd, "visual.ln_post", device, 1e-5)
        self.vproj_t = Lin(sd["visual.proj"].t().contiguous(), None, torch.float32, device)  # [E, W]
        self.first_token_last_block = os.environ.get("MARS_FULL_LAST_BLOCK", "0") != "1"
        # The folded LayerNorms give up a factor |mean| / std of accuracy on a row whose mean is large against its
        # spread (they round x, not x - mean, to bf16).  Every folded run leaves the largest such factor above this
        # limit it met; the next call looks at it and, if there is one, goes back to stand-alone LayerNorms for
        # the rest of the process (MARS_FOLD_LN_LIMIT, default 8; 0 or less: do not look).
        self._sd = sd                       # (a reference, for _unfold)
        self.fold_limit = float(os.environ.get("MARS_FOLD_LN_LIMIT", "8"))
        self._fold_watch: Optional[dict] = None
        self._bufs: Dict[int, dict] = {}
        self.input_resolution = cfg.image_resolution

    def _fold_now(self) -> bool:
        """Whether this call folds its LayerNorms: looks at what the folded runs so far have reported."""
        if not self.fold_ln:
            return False
        if self.fold_limit <= 0 or self.device.type != "cuda":
            return True
        wt = self._fold_watch
        if wt is None:
            wt = self._fold_watch = dict(dev=torch.zeros(1, dtype=torch.float32, device=self.device),
                                         host=torch.zeros(1, dtype=torch.float32).pin_memory(),
                                         event=torch.cuda.Event(), pending=False)
        if wt["pending"] and wt["event"].query():
            wt["pending"] = False
            worst = float(wt["host"][0])
            if worst > 0.0:
                warnings.warn(f"AlphaCLIP: a row of the residual stream has |mean| / std = {worst:.1f} (limit {self.fold_limit:g}); "
                              "the LayerNorms run stand-alone from here on (MARS_FOLD_LN_LIMIT)")
                self._unfold()
        return self.fold_ln

    def _unfold(self) -> None:
        """The blocks built with folded LayerNorms get the plain weights of their qkv and fc1 products back (from the
        state dict the model was built from) and run through ``Tower.block_forward`` from here on."""
        sd = self._sd
        for b in self.vtower.blocks:
            if b.folded:
                n = b.names
                b.qkv = Lin(sd[n["qkv_w"]], sd[n["qkv_b"]], self.dtype, self.device)
                b.fc1 = Lin(sd[n["fc1"] + ".weight"], sd[n["fc1"] + ".bias"], self.dtype, self.device)
                b.folded = False
        self.fold_ln = False

    def visual(self, image: torch.Tensor, alpha: torch.Tensor, image_u8: bool = False, mean=None, std=None, on_block=None) -> torch.Tensor:
        """model.visual(image, alpha) (alpha_clip/model.py:358-386).  ``image`` is [n,3,R,R] or [1,3,R,R]
        (one image shared by all alpha planes: its patch embedding is computed once and broadcast).
        ``on_block(i)``: called when transformer block i has been queued."""
        cfg = self.cfg
        n = alpha.shape[0]
        p, D, R = cfg.vision_patch_size, cfg.vision_width, cfg.image_resolution
        g = R // p
        Pn, T = g * g, g * g + 1
        assert alpha.shape[1:] == (1, R, R) and image.shape[1:] == (3, R, R) and image.shape[0] in (1, n)
        b = self._bufs.get(n)
        if b is None:
            b = self._bufs[n] = dict(
                cols_a=torch.zeros(n * Pn, self.conv_alpha.Kp, dtype=self.dtype, device=self.device),
                cols_rgb={},
                rgb=torch.zeros(Pn, D, dtype=torch.float32, device=self.device),
                cls=torch.zeros(n, D, dtype=torch.float32, device=self.device),
                feat=torch.zeros(n, cfg.embed_dim, dtype=torch.float32, device=self.device),
            )
        X = torch.empty(n, T, D, dtype=torch.float32, device=self.device)
        shared = image.shape[0] == 1
        nb_img = image.shape[0]
        cols_rgb = b["cols_rgb"].get(nb_img)
        if cols_rgb is None:
            cols_rgb = b["cols_rgb"][nb_img] = torch.zeros(nb_img * Pn, self.conv_rgb.Kp, dtype=self.dtype, device=self.device)
        img = image.contiguous() if image.dtype == torch.uint8 else image.contiguous().float()
        ops.im2col(img, p, cols_rgb, mean=mean, std=std, prescale=(1.0 / 255.0) if image.dtype == torch.uint8 else 1.0)
        alpha_f = alpha.contiguous().float() if alpha.dtype != torch.float32 else alpha.contiguous()
        ops.im2col(alpha_f, p, b["cols_a"])
        if shared:
            # rgb[Pn, D] = conv1(image) + pos[1:], then X[b, 1+i] = conv1_alpha(alpha_b)[i] + rgb[i]
            ops.gemm(cols_rgb, self.conv_rgb.w, b["rgb"], Pn, D, self.conv_rgb.Kp, lda=self.conv_rgb.Kp, ldb=self.conv_rgb.Kp,
                     ldc=D, residual=self.vpos, r_off=D, ldr=D)
            ops.gemm(b["cols_a"], self.conv_alpha.w, X, Pn, D, self.conv_alpha.Kp, lda=self.conv_alpha.Kp, ldb=self.conv_alpha.Kp,
                     ldc=D, c_off=D, nb0=n, sA=(Pn * self.conv_alpha.Kp, 0), sC=(T * D, 0), residual=b["rgb"], ldr=D)
        else:
            ops.gemm(cols_rgb, self.conv_rgb.w, X, Pn, D, self.conv_rgb.Kp, lda=self.conv_rgb.Kp, ldb=self.conv_rgb.Kp, ldc=D,
                     c_off=D, nb0=n, sA=(Pn * self.conv_rgb.Kp, 0), sC=(T * D, 0), residual=self.vpos, r_off=D, ldr=D)
            ops.gemm(b["cols_a"], self.conv_alpha.w, X, Pn, D, self.conv_alpha.Kp, lda=self.conv_alpha.Kp, ldb=self.conv_alpha.Kp,
                     ldc=D, c_off=D, nb0=n, sA=(Pn * self.conv_alpha.Kp, 0), sC=(T * D, 0), residual=X, r_off=D, ldr=D,
                     sR=(T * D, 0))
        ops.write_prefix_tokens(X, self.class_embedding, self.vpos[0], None)
        X2 = X.view(n * T, D)
        ops.layernorm(X2, self.ln_pre.w, self.ln_pre.b, X2, self.ln_pre.eps)
        if self._fold_now() and len(self.vtower.blocks) > 1:
            wt = self._fold_watch
            self.vtower.blocks_forward_folded(X, self.vtower.blocks[:-1], on_block=on_block,
                                              watch=(wt["dev"], self.fold_limit) if wt is not None else None)
            if wt is not None:              # its value travels to the host behind the run; nobody waits for it here
                wt["host"].copy_(wt["dev"], non_blocking=True)
                wt["event"].record()
                wt["pending"] = True
        else:
            for i, blk in enumerate(self.vtower.blocks[:-1]):
                self.vtower.block_forward(X, blk)
                if on_block is not None:
                    on_block(i)
        if self.first_token_last_block:
            cls_rows = self.vtower.block_forward_first_token(X, self.vtower.blocks[-1])   # only the class token is read below
        else:
            self.vtower.block_forward(X, self.vtower.blocks[-1])
            cls_rows = X[:, 0, :]                                # view with row stride T*D
        ops.layernorm(cls_rows, self.ln_post.w, self.ln_post.b, b["cls"], self.ln_post.eps)
        ops.linear(b["cls"], self.vproj_t.w, b["feat"])
        return b["feat"]
