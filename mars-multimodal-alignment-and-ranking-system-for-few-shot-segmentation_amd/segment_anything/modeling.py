"""Segment-Anything model on the HIP kernels.

Interfaces of /root/reference/segment_anything/modeling/{image_encoder,prompt_encoder,mask_decoder,
transformer,sam}.py: ``ImageEncoderViT(x)``, ``PromptEncoder(points, boxes, masks)`` /
``get_dense_pe()``, ``MaskDecoder(image_embeddings, image_pe, sparse, dense, multimask_output)``,
``Sam.preprocess`` / ``postprocess_masks`` / ``mask_threshold`` / ``image_format``.

Layout: token-major everywhere ([tokens, channels], the reference's NCHW / NHWC are views of it).
Weights keep the reference's state-dict names.  Head dims that are not a multiple of 64 (ViT-H: 80,
the decoder: 16 / 32) are zero-padded inside the packed q/k/v projections, which leaves every dot
product unchanged.  Attention is the unfused path (score GEMM, bias + softmax kernel, PV GEMM): the
decomposed relative-position bias is added inside the softmax kernel.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import os

import torch

from .. import ops, resample
from ..models import Lin, _pad_to
from ..synth import SamCfg

SD = Dict[str, torch.Tensor]
PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().float().to(device).contiguous()


def _pad_heads_rows(w: torch.Tensor, b: Optional[torch.Tensor], heads: int, hd: int, hdp: int):
    """[heads*hd, K] (+bias) -> [heads*hdp, K] with zero rows for the padded head dims."""
    K = w.shape[1]
    wp = torch.zeros(heads, hdp, K)
    wp[:, :hd] = w.float().view(heads, hd, K)
    bp = None
    if b is not None:
        bp = torch.zeros(heads, hdp)
        bp[:, :hd] = b.float().view(heads, hd)
        bp = bp.reshape(-1)
    return wp.reshape(heads * hdp, K), bp


def _pad_heads_cols(w: torch.Tensor, heads: int, hd: int, hdp: int):
    """[N, heads*hd] -> [N, heads*hdp] with zero columns for the padded head dims."""
    N = w.shape[0]
    wp = torch.zeros(N, heads, hdp)
    wp[:, :, :hd] = w.float().view(N, heads, hd)
    return wp.reshape(N, heads * hdp)


class _Attn:
    """Scores / softmax / PV for packed projections with padded head dim; buffers are cached per shape."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self._ws: Dict[tuple, dict] = {}

    def ws(self, nb: int, H: int, Tq: int, Tk: int, hdp: int):
        key = (nb, H, Tq, Tk, hdp)
        w = self._ws.get(key)
        if w is None:
            Tp = _pad_to(Tk, 64)
            dev = self.device
            w = self._ws[key] = dict(
                Tp=Tp, S=torch.zeros(nb * H, Tq, Tp, dtype=torch.float32, device=dev),
                P=torch.zeros(nb * H, Tq, Tp, dtype=self.dtype, device=dev),
                VT=torch.zeros(nb * H, hdp, Tp, dtype=self.dtype, device=dev))
        return w

    def run(self, q: torch.Tensor, q_off: int, k: torch.Tensor, k_off: int, v: torch.Tensor, v_off: int, out: torch.Tensor,
            nb: int, H: int, Tq: int, Tk: int, hdp: int, scale: float, rel=None):
        """q rows nb*Tq, k / v rows nb*Tk (row-major, head h at column off + h*hdp); out [nb*Tq, H*hdp]."""
        w = self.ws(nb, H, Tq, Tk, hdp)
        Tp, S, P, VT = w["Tp"], w["S"], w["P"], w["VT"]
        ldq, ldk, ldv = q.stride(0), k.stride(0), v.stride(0)
        ops.gemm(q, k, S, Tq, Tk, hdp, lda=ldq, ldb=ldk, ldc=Tp, a_off=q_off, b_off=k_off, nb0=nb, nb1=H,
                 sA=(Tq * ldq, hdp), sB=(Tk * ldk, hdp), sC=(H * Tq * Tp, Tq * Tp), alpha=scale)
        if rel is None:
            ops.softmax_bias_rows(S, P, nb * H * Tq, Tk)
        else:
            rel_h, rel_w, Hk, Wk = rel
            ops.softmax_bias_rows(S, P, nb * H * Tq, Tk, rel_h, rel_w, Hk, Wk)
        ops.transpose_raw(v, v_off, ldv, (Tk * ldv, hdp), VT, 0, Tp, (H * hdp * Tp, hdp * Tp), Tk, hdp, nb, H)
        ops.gemm(P, VT, out, Tq, hdp, Tp, lda=Tp, ldb=Tp, ldc=out.stride(0), nb0=nb, nb1=H, sA=(H * Tq * Tp, Tq * Tp),
                 sB=(H * hdp * Tp, hdp * Tp), sC=(Tq * out.stride(0), hdp))


# ==========================================================================================
# image encoder (image_encoder.py:17-116)
# ==========================================================================================
def _rel_table(q_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """get_rel_pos (image_encoder.py:292-322) for equal query / key sizes -> [q, k, hd]."""
    L = 2 * q_size - 1
    r = rel_pos.float()
    if r.shape[0] != L:
        r = torch.nn.functional.interpolate(r.reshape(1, r.shape[0], -1).permute(0, 2, 1), size=L, mode="linear")
        r = r.reshape(-1, L).permute(1, 0)
    idx = (torch.arange(q_size)[:, None] - torch.arange(q_size)[None, :]) + (q_size - 1)
    return r[idx.long()].contiguous()


class _EncBlock:
    pass


class ImageEncoderViT:
    def __init__(self, cfg: SamCfg, sd: SD, device="cuda", dtype=torch.float32, prefix: str = "image_encoder"):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.img_size, self.patch_size = cfg.img_size, cfg.patch_size
        D, H = cfg.embed_dim, cfg.num_heads
        assert D % 64 == 0 and cfg.out_chans % 64 == 0, "widths must be multiples of 64 (GEMM K granularity)"
        self.G = G = cfg.img_size // cfg.patch_size
        self.hd = hd = D // H
        # bf16 towers: scores, bias, softmax and P.V in one kernel (no [heads, T, T] score matrix); MARS_FLASH=0 or
        # an f32 tower keep the three-launch form, whose products need head dims padded to a multiple of 64.  The fused
        # kernel also takes the 80-dim heads of ViT-H as they are (MARS_SAM_HEAD80=0: padded to 128 like the rest): the
        # qkv product and the projection lose the 37 % of their columns / K that were zeros.
        fused_ok = dtype == torch.bfloat16 and os.environ.get("MARS_FLASH", "1") != "0"
        self.hdp = hdp = 80 if (hd == 80 and fused_ok and os.environ.get("MARS_SAM_HEAD80", "1") != "0") else _pad_to(hd, 64)
        dev = self.device
        self.patch = Lin(sd[prefix + ".patch_embed.proj.weight"].reshape(D, -1), sd[prefix + ".patch_embed.proj.bias"], dtype, dev)
        self.pos = _dev_f32(sd[prefix + ".pos_embed"].reshape(G * G, D), dev)
        self.blocks: List[_EncBlock] = []
        for i in range(cfg.depth):
            p = f"{prefix}.blocks.{i}"
            b = _EncBlock()
            b.ws = 0 if i in cfg.global_attn_indexes else cfg.window_size
            b.ln1 = (_dev_f32(sd[p + ".norm1.weight"], dev), _dev_f32(sd[p + ".norm1.bias"], dev))
            b.ln2 = (_dev_f32(sd[p + ".norm2.weight"], dev), _dev_f32(sd[p + ".norm2.bias"], dev))
            wq, bq = sd[p + ".attn.qkv.weight"].float(), sd[p + ".attn.qkv.bias"].float()
            parts = [_pad_heads_rows(wq[s * D:(s + 1) * D], bq[s * D:(s + 1) * D], H, hd, hdp) for s in range(3)]
            b.qkv = Lin(torch.cat([w for w, _ in parts]), torch.cat([bb for _, bb in parts]), dtype, dev)
            b.proj = Lin(_pad_heads_cols(sd[p + ".attn.proj.weight"], H, hd, hdp), sd[p + ".attn.proj.bias"], dtype, dev)
            # a window's padding token as the projection leaves it: the bias, rounded as its output is (what the row-map kernels
            # need: bf16 queries, windows of 65 to 224 tokens, head dims a multiple of 8 up to 128)
            b.qkv_pad = b.qkv.b.to(torch.bfloat16) if (dtype == torch.bfloat16 and 64 < b.ws * b.ws <= 224 and hd % 8 == 0 and hd <= 128) else None
            size = G if b.ws == 0 else b.ws
            b.Rh = _dev_f32(_rel_table(size, sd[p + ".attn.rel_pos_h"]), dev)
            b.Rw = _dev_f32(_rel_table(size, sd[p + ".attn.rel_pos_w"]), dev)
            b.fc1 = Lin(sd[p + ".mlp.lin1.weight"], sd[p + ".mlp.lin1.bias"], dtype, dev)
            b.fc2 = Lin(sd[p + ".mlp.lin2.weight"], sd[p + ".mlp.lin2.bias"], dtype, dev)
            self.blocks.append(b)
        C = cfg.out_chans
        self.neck0 = Lin(sd[prefix + ".neck.0.weight"].reshape(C, D), None, dtype, dev)
        self.neck1 = (_dev_f32(sd[prefix + ".neck.1.weight"], dev), _dev_f32(sd[prefix + ".neck.1.bias"], dev))
        # 3x3 convolution as a GEMM over gathered neighbourhoods: K index = (ky, kx, cin)
        self.neck2 = Lin(sd[prefix + ".neck.2.weight"].permute(0, 2, 3, 1).reshape(C, 9 * C), None, dtype, dev)
        self.neck3 = (_dev_f32(sd[prefix + ".neck.3.weight"], dev), _dev_f32(sd[prefix + ".neck.3.bias"], dev))
        self._attn = _Attn(dev, dtype)
        self.fused_attention = fused_ok and hdp in (64, 80, 128)
        self.fused_partition = os.environ.get("MARS_SAM_FUSED_PARTITION", "1") != "0"
        self._build_index_maps()
        self._mean = torch.tensor(PIXEL_MEAN, dtype=torch.float32, device=dev)
        self._std = torch.tensor(PIXEL_STD, dtype=torch.float32, device=dev)
        T, hid = G * G, self.blocks[0].fc1.N
        Tw = self.n_win * cfg.window_size ** 2 if self.n_win else T
        z = lambda r, c, dt=dtype: torch.zeros(r, c, dtype=dt, device=dev)
        self.w = dict(a0=z(T, self.patch.Kp), X=z(T, D, torch.float32), xn=z(T, D), xw=z(Tw, D), qkv=z(max(T, Tw), 3 * H * hdp),
                      ao=z(max(T, Tw), H * hdp), aou=z(T, H * hdp), hid=z(T, hid), y0=z(T, C, torch.float32), y1=z(T, C),
                      a3=z(T, self.neck2.Kp), y2=z(T, C, torch.float32), feats=z(T, C, torch.float32))

    def _build_index_maps(self):
        """Row maps of window_partition / window_unpartition (image_encoder.py:243-289) and of the 3x3
        neighbourhoods of the neck convolution; -1 selects a zero row."""
        G, ws, dev = self.G, self.cfg.window_size, self.device
        self.n_win = 0
        if any(b.ws > 0 for b in self.blocks):
            Gp = _pad_to(G, ws)
            nw = Gp // ws
            self.n_win = nw * nw
            part = torch.full((self.n_win * ws * ws,), -1, dtype=torch.int32)
            unpart = torch.zeros(G * G, dtype=torch.int32)
            for wy in range(nw):
                for wx in range(nw):
                    for iy in range(ws):
                        for ix in range(ws):
                            y, x = wy * ws + iy, wx * ws + ix
                            r = ((wy * nw + wx) * ws + iy) * ws + ix
                            if y < G and x < G:
                                part[r] = y * G + x
                                unpart[y * G + x] = r
            self.part_idx, self.unpart_idx = part.to(dev), unpart.to(dev)
        taps = []
        ys, xs = torch.meshgrid(torch.arange(G), torch.arange(G), indexing="ij")
        for ky in (-1, 0, 1):
            for kx in (-1, 0, 1):
                yy, xx = ys + ky, xs + kx
                ok = (yy >= 0) & (yy < G) & (xx >= 0) & (xx < G)
                taps.append(torch.where(ok, yy * G + xx, torch.full_like(yy, -1)).reshape(-1).int().to(dev))
        self.tap_idx = taps

    def forward_tokens(self, image: torch.Tensor) -> torch.Tensor:
        """image: f32 [1, 3, S, S] of raw pixel values whose padding already holds the channel means (so
        that it normalises to zero, Sam.preprocess); returns the embedding as tokens [G*G, out_chans] f32."""
        cfg, w, D, H, hd, hdp, G = self.cfg, self.w, self.cfg.embed_dim, self.cfg.num_heads, self.hd, self.hdp, self.G
        assert image.shape == (1, 3, cfg.img_size, cfg.img_size), "one padded image at a time"
        T = G * G
        ops.im2col(image, cfg.patch_size, w["a0"], mean=self._mean, std=self._std)
        ops.linear(w["a0"], self.patch.w, w["X"], bias=self.patch.b, residual=self.pos)       # patch embed + pos embed
        X = w["X"]
        for b in self.blocks:
            # fused_partition: the windows are not materialised.  The projection runs on the image's rows; the relative-position
            # kernel and the attention read a window's tokens through the row map of window_partition (a padding token is
            # what the projection makes of a zero row: its bias) and the attention writes in image order.
            fp = self.fused_partition and b.ws > 0 and self.fused_attention and b.qkv_pad is not None
            ops.layernorm(X, b.ln1[0], b.ln1[1], w["xn"], 1e-6)
            if b.ws > 0:
                if not fp:
                    ops.gather_rows(w["xn"], self.part_idx, w["xw"])                             # zero rows for the padding
                src, nb, Tt, gw = (w["xn"] if fp else w["xw"]), self.n_win, b.ws * b.ws, b.ws
            else:
                src, nb, Tt, gw = w["xn"], 1, T, G
            rows = T if fp else nb * Tt
            qkv = w["qkv"][:rows]
            ops.linear(src[:rows], b.qkv.w, qkv, bias=b.qkv.b)
            rel_h = self._rel_buf("h", nb * H * Tt * (Tt // gw))
            rel_w = self._rel_buf("w", nb * H * Tt * gw)
            ao = w["ao"][:rows]
            if fp:
                ops.relpos_bias(qkv, nb, Tt, H, hd, hdp, gw, b.Rh, b.Rw, rel_h, rel_w, in_rows=self.part_idx, pad_row=b.qkv_pad)
                ao = w["aou"]
                ops.sam_attn(qkv, ao, rel_h, rel_w, nb, Tt, H, hdp, Tt // gw, gw, hd ** -0.5, in_rows=self.part_idx, pad_row=b.qkv_pad,
                             out_rows=self.part_idx)
            else:
                ops.relpos_bias(qkv, nb, Tt, H, hd, hdp, gw, b.Rh, b.Rw, rel_h, rel_w)
                if self.fused_attention:
                    ops.sam_attn(qkv, ao, rel_h, rel_w, nb, Tt, H, hdp, Tt // gw, gw, hd ** -0.5)
                else:
                    if hdp % 64:
                        raise RuntimeError("this tower stores its heads unpadded for the fused attention kernel; build it with MARS_FLASH=0 "
                                           "(or MARS_SAM_HEAD80=0) to run the three-launch form")
                    self._attn.run(qkv, 0, qkv, H * hdp, qkv, 2 * H * hdp, ao, nb, H, Tt, Tt, hdp, hd ** -0.5, rel=(rel_h, rel_w, Tt // gw, gw))
            if b.ws > 0 and not fp:
                ops.gather_rows(ao, self.unpart_idx, w["aou"])
                ao = w["aou"]
            ops.linear(ao[:T], b.proj.w, X, bias=b.proj.b, residual=X)
            ops.layernorm(X, b.ln2[0], b.ln2[1], w["xn"], 1e-6)
            ops.linear(w["xn"], b.fc1.w, w["hid"], bias=b.fc1.b, act=1)
            ops.linear(w["hid"], b.fc2.w, X, bias=b.fc2.b, residual=X)
        # neck: 1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d
        ops.add_cast(X, None, w["xn"])
        ops.linear(w["xn"], self.neck0.w, w["y0"])
        ops.layernorm(w["y0"], self.neck1[0], self.neck1[1], w["y1"], 1e-6)
        C = cfg.out_chans
        for t, idx in enumerate(self.tap_idx):
            ops.gather_rows(w["y1"], idx, w["a3"], dst_col=t * C, ncols=C)
        ops.linear(w["a3"], self.neck2.w, w["y2"])
        ops.layernorm(w["y2"], self.neck3[0], self.neck3[1], w["feats"], 1e-6)
        return w["feats"]

    def _rel_buf(self, tag: str, n: int) -> torch.Tensor:
        key = "rel_" + tag
        buf = self.w.get(key)
        if buf is None or buf.numel() < n:
            buf = self.w[key] = torch.zeros(n, dtype=torch.float32, device=self.device)
        return buf

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """Reference layout: normalised and zero-padded [1,3,S,S] in, [1, out_chans, G, G] out (:106-116)."""
        raw = x.to(self.device).float() * self._std.view(1, 3, 1, 1) + self._mean.view(1, 3, 1, 1)
        f = self.forward_tokens(raw.contiguous())
        return f.view(1, self.G, self.G, -1).permute(0, 3, 1, 2).contiguous()


# ==========================================================================================
# prompt encoder (prompt_encoder.py:16-214), points only
# ==========================================================================================
class PromptEncoder:
    def __init__(self, cfg: SamCfg, sd: SD, device="cuda"):
        self.device = torch.device(device)
        self.embed_dim = C = cfg.out_chans
        G = cfg.img_size // cfg.patch_size
        self.image_embedding_size = (G, G)
        self.input_image_size = (cfg.img_size, cfg.img_size)
        dev = self.device
        self.gauss = _dev_f32(sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"], dev)
        self.table = torch.cat([_dev_f32(sd["prompt_encoder.not_a_point_embed.weight"], dev),
                                _dev_f32(sd["prompt_encoder.point_embeddings.0.weight"], dev),
                                _dev_f32(sd["prompt_encoder.point_embeddings.1.weight"], dev),
                                _dev_f32(sd["prompt_encoder.point_embeddings.2.weight"], dev),
                                _dev_f32(sd["prompt_encoder.point_embeddings.3.weight"], dev)], dim=0).contiguous()
        self.no_mask = _dev_f32(sd["prompt_encoder.no_mask_embed.weight"], dev).reshape(-1)
        md = "prompt_encoder.mask_downscaling"
        self.mask_params = tuple(_dev_f32(sd[f"{md}.{i}.{n}"], dev) for i, n in
                                 ((0, "weight"), (0, "bias"), (1, "weight"), (1, "bias"), (3, "weight"), (3, "bias"),
                                  (4, "weight"), (4, "bias"), (6, "weight"), (6, "bias")))
        # dense positional code of the embedding grid (:194-205), token-major [G*G, C]
        ys, xs = torch.meshgrid((torch.arange(G) + 0.5) / G, (torch.arange(G) + 0.5) / G, indexing="ij")
        grid = torch.stack([xs, ys], dim=-1).reshape(-1, 2).float().to(dev).contiguous()
        self.dense_pe_tokens = torch.zeros(G * G, C, dtype=torch.float32, device=dev)
        ops.fourier_pe(grid, self.gauss, self.dense_pe_tokens)

    def get_dense_pe(self) -> torch.Tensor:
        G = self.image_embedding_size[0]
        return self.dense_pe_tokens.view(1, G, G, -1).permute(0, 3, 1, 2)

    def embed_points(self, coords: torch.Tensor, labels: torch.Tensor, pad: bool = True) -> torch.Tensor:
        """_embed_points (:73-91): coords [B, N, 2] in the input frame, labels [B, N] -> [B, N(+1), C]."""
        B, N, _ = coords.shape
        c = coords.to(self.device).float() + 0.5
        lab = labels.to(self.device).int()
        if pad:
            c = torch.cat([c, torch.zeros(B, 1, 2, device=self.device)], dim=1)
            lab = torch.cat([lab, -torch.ones(B, 1, dtype=torch.int32, device=self.device)], dim=1)
        c = c / torch.tensor([self.input_image_size[1], self.input_image_size[0]], dtype=torch.float32, device=self.device)
        out = torch.zeros(B, c.shape[1], self.embed_dim, dtype=torch.float32, device=self.device)
        ops.fourier_pe(c.contiguous(), self.gauss, out, labels=lab.contiguous().view(-1), table=self.table)
        return out

    def embed_boxes(self, boxes: torch.Tensor) -> torch.Tensor:
        """_embed_boxes (:93-101): boxes [B, 4] (x0, y0, x1, y1) in the input frame -> two corner tokens [B, 2, C]."""
        b = boxes.to(self.device).float().reshape(-1, 2, 2) + 0.5
        B = b.shape[0]
        c = b / torch.tensor([self.input_image_size[1], self.input_image_size[0]], dtype=torch.float32, device=self.device)
        lab = torch.tensor([2, 3], dtype=torch.int32, device=self.device).repeat(B)
        out = torch.zeros(B, 2, self.embed_dim, dtype=torch.float32, device=self.device)
        ops.fourier_pe(c.contiguous(), self.gauss, out, labels=lab, table=self.table)
        return out

    def embed_prompts(self, points=None, boxes=None) -> torch.Tensor:
        """Sparse tokens of forward (:147-155): points (padded with a not-a-point token when there is no box), then box corners."""
        parts = []
        if points is not None:
            parts.append(self.embed_points(torch.as_tensor(points[0]), torch.as_tensor(points[1]), pad=boxes is None))
        if boxes is not None:
            parts.append(self.embed_boxes(torch.as_tensor(boxes)))
        if not parts:
            raise NotImplementedError("a prompt needs points or a box (mask prompts are not built)")
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1).contiguous()

    def embed_masks_tokens(self, masks: torch.Tensor) -> torch.Tensor:
        """_embed_masks (:102-105): low-resolution mask logits [B, 1, 4G, 4G] -> dense embedding, token-major [B, G*G, C]."""
        G = self.image_embedding_size[0]
        m = masks.to(self.device).float().contiguous()
        out = torch.empty(m.shape[0], G * G, self.embed_dim, dtype=torch.float32, device=self.device)
        ops.mask_embed(m, G, self.mask_params, 1e-6, out)
        return out

    def __call__(self, points=None, boxes=None, masks=None):
        sparse = self.embed_prompts(points, boxes)
        G = self.image_embedding_size[0]
        if masks is not None:
            dense = self.embed_masks_tokens(masks).view(-1, G, G, self.embed_dim).permute(0, 3, 1, 2)
        else:
            dense = self.no_mask.view(1, -1, 1, 1).expand(sparse.shape[0], -1, G, G)
        return sparse, dense


# ==========================================================================================
# mask decoder (mask_decoder.py:16-149, transformer.py:16-240)
# ==========================================================================================
class _DecAttn:
    def __init__(self, sd: SD, p: str, C: int, heads: int, down: int, device, dtype):
        inner = C // down
        self.H, self.hd = heads, inner // heads
        self.hdp = _pad_to(self.hd, 64)
        mk = lambda n: Lin(*_pad_heads_rows(sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], heads, self.hd, self.hdp), dtype, device)
        self.q, self.k, self.v = mk("q_proj"), mk("k_proj"), mk("v_proj")
        self.o = Lin(_pad_heads_cols(sd[p + ".out_proj.weight"], heads, self.hd, self.hdp), sd[p + ".out_proj.bias"], dtype, device)
        # Heads of 16 or 32 dims with a handful of tokens on one side (every product of the two-way transformer) run
        # on unpadded projections and ops.attn_small: the padded form above moves four times the bytes at 16 dims a head.
        # MARS_SAM_SMALL_ATTN=0 keeps the padded form.
        self.small = self.hd in (16, 32) and heads in (1, 2, 4, 8, 16, 32, 64) and os.environ.get("MARS_SAM_SMALL_ATTN", "1") != "0"
        if self.small:
            mks = lambda n: Lin(sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], dtype, device)
            self.qs, self.ks, self.vs, self.os = mks("q_proj"), mks("k_proj"), mks("v_proj"), mks("out_proj")
            # token -> image attention with the projections of the image stream inside the kernel (ops.sam_t2i_block): the
            # weights in runs of eight K, run-major
            if C == 256 and heads == 8 and self.hd == 16 and dtype == torch.bfloat16 and os.environ.get("MARS_SAM_FUSED_T2I", "1") != "0":
                runs = lambda n: sd[f"{p}.{n}.weight"].float().view(128, 32, 8).permute(1, 0, 2).contiguous().to(device, torch.bfloat16)
                self.wk_runs, self.wv_runs = runs("k_proj"), runs("v_proj")
            # key and value projections of one input as one product (rows: keys, then values)
            self.kvs = Lin(torch.cat([sd[f"{p}.k_proj.weight"], sd[f"{p}.v_proj.weight"]], 0),
                           torch.cat([sd[f"{p}.k_proj.bias"], sd[f"{p}.v_proj.bias"]], 0), dtype, device)


class MaskDecoder:
    def __init__(self, cfg: SamCfg, sd: SD, device="cuda", dtype=torch.float32):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        dev, C = self.device, cfg.out_chans
        assert C % 64 == 0, "the decoder width must be a multiple of 64"
        self.C = C
        self.out_tokens = torch.cat([_dev_f32(sd["mask_decoder.iou_token.weight"], dev), _dev_f32(sd["mask_decoder.mask_tokens.weight"], dev)], 0)
        pre = "mask_decoder.transformer"
        self.layers = []
        ln = lambda p: (_dev_f32(sd[p + ".weight"], dev), _dev_f32(sd[p + ".bias"], dev))
        for i in range(cfg.dec_depth):
            p = f"{pre}.layers.{i}"
            L = dict(self_attn=_DecAttn(sd, p + ".self_attn", C, cfg.dec_heads, 1, dev, dtype),
                     t2i=_DecAttn(sd, p + ".cross_attn_token_to_image", C, cfg.dec_heads, 2, dev, dtype),
                     i2t=_DecAttn(sd, p + ".cross_attn_image_to_token", C, cfg.dec_heads, 2, dev, dtype),
                     n1=ln(p + ".norm1"), n2=ln(p + ".norm2"), n3=ln(p + ".norm3"), n4=ln(p + ".norm4"),
                     lin1=Lin(sd[p + ".mlp.lin1.weight"], sd[p + ".mlp.lin1.bias"], dtype, dev),
                     lin2=Lin(sd[p + ".mlp.lin2.weight"], sd[p + ".mlp.lin2.bias"], dtype, dev))
            # image -> token attention, norm4 and the next layer's operands as one kernel (ops.sam_i2t_block) at the shipped
            # width in bf16: Wq in runs of eight K, run-major; Wo the same with its K index in the order a lane holds o
            A = L["i2t"]
            if C == 256 and A.H == 8 and A.hd == 16 and dtype == torch.bfloat16 and os.environ.get("MARS_SAM_FUSED_I2T", "1") != "0":
                pi = p + ".cross_attn_image_to_token"
                wq, wo = sd[pi + ".q_proj.weight"].float(), sd[pi + ".out_proj.weight"].float()          # [128, 256], [256, 128]
                slot = torch.tensor([16 * h + 8 * (e // 4) + 4 * g + e % 4 for h in range(8) for g in range(2) for e in range(8)])
                L["i2t_fused"] = dict(
                    wq=wq.view(128, 32, 8).permute(1, 0, 2).contiguous().to(dev, torch.bfloat16),
                    wo=wo[:, slot].view(256, 16, 8).permute(1, 0, 2).contiguous().to(dev, torch.bfloat16),
                    prm=torch.cat([sd[pi + ".out_proj.bias"].float(), sd[p + ".norm4.weight"].float(), sd[p + ".norm4.bias"].float()]).contiguous().to(dev))
            self.layers.append(L)
        self.final = _DecAttn(sd, pre + ".final_attn_token_to_image", C, cfg.dec_heads, 2, dev, dtype)
        self.nf = ln(pre + ".norm_final_attn")
        # transposed convolutions (kernel 2, stride 2) as one GEMM per output row parity:
        # weight [Cin, Cout, 2, 2] -> per dy a [2*Cout, Cin] matrix with N index = (dx, cout)
        def convT(name):
            wt = sd[name + ".weight"].float()                      # [Cin, Cout, 2, 2]
            bias = sd[name + ".bias"].float()
            return [Lin(wt[:, :, dy, :].permute(2, 1, 0).reshape(2 * wt.shape[1], wt.shape[0]), bias.repeat(2), dtype, dev) for dy in (0, 1)]
        self.up0 = convT("mask_decoder.output_upscaling.0")
        self.up_ln = ln("mask_decoder.output_upscaling.1")
        self.up3 = convT("mask_decoder.output_upscaling.3")
        # At the shipped width (256 -> 64 -> 32) in bf16 the upscaling runs as two fused kernels (ops.sam_upscale_*):
        # whole weights with rows (dy, dx, channel); the first in runs of eight K, run-major, as its kernel keeps it in LDS.
        # MARS_SAM_FUSED_UPSCALE=0 keeps the product-per-stage form.
        self.fused_up = C == 256 and dtype == torch.bfloat16 and os.environ.get("MARS_SAM_FUSED_UPSCALE", "1") != "0"
        if self.fused_up:
            w0 = sd["mask_decoder.output_upscaling.0.weight"].float().permute(2, 3, 1, 0).reshape(256, 256)
            self.up0_runs = w0.view(256, 32, 8).permute(1, 0, 2).contiguous().to(dev, torch.bfloat16)
            self.up0_prm = torch.stack([sd["mask_decoder.output_upscaling.0.bias"].float(), sd["mask_decoder.output_upscaling.1.weight"].float(),
                                        sd["mask_decoder.output_upscaling.1.bias"].float()]).contiguous().to(dev)
            self.up3_rows = sd["mask_decoder.output_upscaling.3.weight"].float().permute(2, 3, 1, 0).reshape(128, 64).contiguous().to(dev, torch.bfloat16)
            self.up3_bias = _dev_f32(sd["mask_decoder.output_upscaling.3.bias"], dev)
        self.hyper = [[Lin(sd[f"mask_decoder.output_hypernetworks_mlps.{i}.layers.{j}.weight"],
                           sd[f"mask_decoder.output_hypernetworks_mlps.{i}.layers.{j}.bias"], dtype, dev) for j in range(3)] for i in range(4)]
        self.iou = [Lin(sd[f"mask_decoder.iou_prediction_head.layers.{j}.weight"], sd[f"mask_decoder.iou_prediction_head.layers.{j}.bias"], dtype, dev)
                    for j in range(cfg.iou_head_depth)]
        self._attn = _Attn(dev, dtype)
        self._bufs: Dict[tuple, torch.Tensor] = {}

    def _buf(self, name: str, shape, dtype=None) -> torch.Tensor:
        dtype = dtype or self.dtype
        key = (name, tuple(shape), dtype)
        b = self._bufs.get(key)
        if b is None:
            b = self._bufs[key] = torch.zeros(*shape, dtype=dtype, device=self.device)
        return b

    def _attention(self, A: _DecAttn, tag: str, q_in, k_in, v_in, nb: int, Tq: int, Tk: int, out: torch.Tensor, residual,
                   q_shared: bool = False, kv_shared: bool = False, k_pe: Optional[torch.Tensor] = None):
        """transformer.py:218-240: q_in [nb*Tq, C], k_in / v_in [nb*Tk, C] in the compute dtype -> out f32 (+ residual).
        ``q_shared``: q_in and residual hold one set of Tq rows for every prompt; ``kv_shared``: k_in / v_in one set of Tk rows
        (the image side of the first layer: projected once, not once a prompt).  ``k_pe`` f32 [Tk, 2*H*hd]: k_in (which is also
        v_in) lacks the positional term; columns [0, H*hd) of the table are its key projection with the bias, the rest the
        bias of the value projection: keys and values come from one product."""
        if A.small and (Tk <= 16 or Tq <= 16):
            W = A.H * A.hd
            nq, nk = (1 if q_shared else nb), (1 if kv_shared else nb)
            q = self._buf(tag + "qs", (nq * Tq, W))
            ops.linear(q_in, A.qs.w, q, bias=A.qs.b)
            if k_pe is not None:                 # k_in is v_in: one product with 2 W columns, k_pe holds both bias terms
                kv = self._buf(tag + "kvs", (nk * Tk, 2 * W))
                ops.gemm(k_in, A.kvs.w, kv, Tk, 2 * W, A.kvs.w.shape[1], lda=k_in.stride(0), ldb=A.kvs.w.stride(0), ldc=2 * W, nb0=nk,
                         sA=(Tk * k_in.stride(0), 0), sC=(Tk * 2 * W, 0), residual=k_pe, ldr=2 * W, sR=(0, 0))
                k, v = kv[:, :W], kv[:, W:]
            else:
                k = self._buf(tag + "ks", (nk * Tk, W)); v = self._buf(tag + "vs", (nk * Tk, W))
                ops.linear(k_in, A.ks.w, k, bias=A.ks.b)
                ops.linear(v_in, A.vs.w, v, bias=A.vs.b)
            Wp = _pad_to(W, 64)
            o = self._buf(tag + "os", (nb * Tq, Wp))                     # (K of the output projection, padded like its weight)
            ops.attn_small(q, k, v, o, nb, A.H, A.hd, Tq, Tk, 1.0 / math.sqrt(A.hd), q_shared=q_shared, kv_shared=kv_shared)
            if q_shared:                                                 # one product a prompt: the residual rows repeat
                N = A.os.w.shape[0]
                ops.gemm(o, A.os.w, out, Tq, N, Wp, lda=Wp, ldb=A.os.w.stride(0), ldc=out.stride(0), nb0=nb, sA=(Tq * Wp, 0),
                         sC=(Tq * out.stride(0), 0), bias=A.os.b, residual=residual, ldr=residual.stride(0), sR=(0, 0))
            else:
                ops.linear(o, A.os.w, out, bias=A.os.b, residual=residual)
            return
        assert not (q_shared or kv_shared) and k_pe is None
        W = A.H * A.hdp
        q = self._buf(tag + "q", (nb * Tq, W)); k = self._buf(tag + "k", (nb * Tk, W)); v = self._buf(tag + "v", (nb * Tk, W))
        ops.linear(q_in, A.q.w, q, bias=A.q.b)
        ops.linear(k_in, A.k.w, k, bias=A.k.b)
        ops.linear(v_in, A.v.w, v, bias=A.v.b)
        o = self._buf(tag + "o", (nb * Tq, W))
        self._attn.run(q, 0, k, 0, v, 0, o, nb, A.H, Tq, Tk, A.hdp, 1.0 / math.sqrt(A.hd))
        ops.linear(o, A.o.w, out, bias=A.o.b, residual=residual)

    def _t2i_fused(self, A: _DecAttn, kpe_ops: torch.Tensor, tag: str, q_in, x, nb: int, Tq: int, Ni: int, out: torch.Tensor, residual,
                   x_tiled: bool = False):
        """Token -> image attention of ``nb`` prompts against their image streams x (bf16 keys, [nb*Ni, C]) with the key and
        value projections inside the attention kernel; then the output projection (+ residual) as in _attention."""
        q = self._buf(tag + "qs", (nb * Tq, 128))
        ops.linear(q_in, A.qs.w, q, bias=A.qs.b)
        qo = self._buf(tag + "qo", (nb, 8, 64, 8))
        ops.sam_t2i_q_operands(q, nb, Tq, qo)
        o = self._buf(tag + "os", (nb * Tq, 128))
        ops.sam_t2i_block(x, A.wk_runs, A.wv_runs, kpe_ops, A.vs.b, qo, nb, Ni, Tq, o, x_tiled=x_tiled)
        ops.linear(o, A.os.w, out, bias=A.os.b, residual=residual)

    def _pe_terms(self, pe_tokens: torch.Tensor) -> Dict[tuple, torch.Tensor]:
        """key_pe W^T + b for the projections that read keys + key_pe: ("q", i) f32 [Ni, 128] the query projection of layer i's
        image -> token attention; ("k", i) f32 [Ni, 256] the key projection of layer i's token -> image attention (i = number
        of layers: the final attention) followed by the value projection's bias, for the joint key / value product.
        key_pe belongs to the model, not to the image: made at first use."""
        key = (pe_tokens.data_ptr(), tuple(pe_tokens.shape))
        if getattr(self, "_pe_key", None) != key:
            ped = pe_tokens.to(self.dtype)
            terms = {}
            for i, L in enumerate(self.layers):
                t = terms[("q", i)] = torch.empty(pe_tokens.shape[0], L["i2t"].qs.w.shape[0], dtype=torch.float32, device=self.device)
                ops.linear(ped, L["i2t"].qs.w, t, bias=L["i2t"].qs.b)
            for i, A in enumerate([L["t2i"] for L in self.layers] + [self.final]):
                W = A.ks.w.shape[0]
                t = terms[("k", i)] = torch.empty(pe_tokens.shape[0], 2 * W, dtype=torch.float32, device=self.device)
                ops.linear(ped, A.ks.w, t[:, :W], bias=A.ks.b)
                t[:, W:] = A.vs.b
                if hasattr(A, "wk_runs") and pe_tokens.shape[0] % 32 == 0:
                    terms[("kops", i)] = ops.sam_t2i_kpe_operands(t[:, :W])
            self._pe_key, self._pe_tab = key, terms
        return self._pe_tab

    def predict_masks(self, feats_tokens: torch.Tensor, dense_pe_tokens: torch.Tensor, sparse: torch.Tensor, no_mask: torch.Tensor,
                      dense_tokens: Optional[torch.Tensor] = None):
        """predict_masks (:112-149): feats_tokens [Ni, C] f32 (one image), sparse [B, Np, C]; the dense prompt embedding is
        the ``no_mask`` vector, or ``dense_tokens`` [B, Ni, C] for mask prompts.  Returns (masks [B, 4, 4G, 4G] f32 logits,
        iou [B, 4])."""
        cfg, C, dev, dtp = self.cfg, self.C, self.device, self.dtype
        B, Np, _ = sparse.shape
        Ni = feats_tokens.shape[0]
        G = int(round(math.sqrt(Ni)))
        Nt = 5 + Np
        f32 = torch.float32
        tokens = self._buf("tokens", (B, Nt, C), f32)
        tokens[:, :5] = self.out_tokens
        tokens[:, 5:] = sparse
        pe = tokens.clone()                                        # query_pe = the initial point embedding
        Q = self._buf("Q", (B * Nt, C), f32); Q.copy_(tokens.view(B * Nt, C))
        # src = image embedding + dense prompt embedding, one copy per prompt (:126-127)
        Kf = self._buf("K", (B * Ni, C), f32)
        qd, vd = self._buf("qd", (B * Nt, C)), self._buf("vd", (B * Ni, C))
        # Without mask prompts the image stream entering the first layer is the same for every prompt: its key, value and
        # query projections are made once, and the stream only becomes per-prompt when the first layer's image -> token
        # attention writes it.
        first = self.layers[0] if self.layers else None
        share0 = dense_tokens is None and first is not None and first["t2i"].small and first["i2t"].small and Nt <= 16
        if dense_tokens is None:
            src0 = self._buf("src0", (Ni, C), f32)
            ops.add_cast(feats_tokens, no_mask.view(1, C), src0)
            if share0:
                kd0, vd0 = self._buf("kd0", (Ni, C)), self._buf("vd0", (Ni, C))
                ops.stream_copies(src0, kd0, vd0, add=dense_pe_tokens)
            else:
                Kf.view(B, Ni, C).copy_(src0)
        else:
            ops.add_cast(dense_tokens.reshape(B * Ni, C).contiguous(), feats_tokens.contiguous(), Kf)
        td = self._buf("td", (B * Nt, C))
        hid = self._buf("hid", (B * Nt, self.layers[0]["lin2"].Kp))      # as wide as the padded K of lin2; pads stay zero
        pe2 = pe.view(B * Nt, C)
        # kd = keys + key_pe and vd = keys in the compute dtype, made once per state of the image stream: here, and by
        # the pass that applies norm4 at the end of each layer
        # With the image -> token block as one kernel (ops.sam_i2t_block) no copy of keys + key_pe exists after the first
        # layer: the positional term enters the projections that would have read it as a table, made once per model.
        fused_i2t = all("i2t_fused" in L and L["t2i"].small for L in self.layers) and self.final.small and Nt <= 8 and Ni % 32 == 0
        pe_terms = self._pe_terms(dense_pe_tokens) if fused_i2t else None
        # (a workgroup of the fused token -> image kernel serves four prompts: small batches leave the GPU empty and keep the
        # separate projection and attention launches)
        fused_t2i = (fused_i2t and B >= int(os.environ.get("MARS_SAM_FUSED_T2I_MIN", "512"))
                     and all(("kops", i) in pe_terms for i in range(1, len(self.layers) + 1)))
        up_fused = self.fused_up and (2 * G) % 32 == 0 and B <= 65535
        # When every reader of the per-prompt image stream is one of the fused kernels (they all hold a 32-token tile the same
        # way), the stream is kept in that lane order instead of row-major after the first layer: loads and stores of 1 KB a
        # wave instead of 32 bytes a token row.  MARS_SAM_TILED=0 keeps it row-major.
        tiled = share0 and fused_t2i and up_fused and os.environ.get("MARS_SAM_TILED", "1") != "0"
        kd = None if fused_i2t else self._buf("kd", (B * Ni, C))          # (keys + key_pe: only the separate launches read it)
        if not share0:
            ops.stream_copies(Kf, None if fused_i2t else kd, vd, add=None if fused_i2t else dense_pe_tokens)

        def ln(x, wb, rows):
            ops.layernorm(x[:rows], wb[0], wb[1], x[:rows], 1e-5)

        for i, L in enumerate(self.layers):
            # (1) self attention of the tokens
            if i == 0:
                ops.add_cast(Q, None, qd)
                self._attention(L["self_attn"], "sa", qd, qd, qd, B, Nt, Nt, Q, None)
            else:
                ops.add_cast(Q, pe2, qd)
                ops.add_cast(Q, None, td)
                self._attention(L["self_attn"], "sa", qd, qd, td, B, Nt, Nt, Q, Q)
            ln(Q, L["n1"], B * Nt)
            # (2) tokens attend to the image
            ops.add_cast(Q, pe2, qd)
            if share0 and i == 0 and fused_t2i and ("kops", 0) in pe_terms:
                self._t2i_fused(L["t2i"], pe_terms[("kops", 0)], "t2i", qd, vd0, B, Nt, Ni, Q, Q)          # (vd0: one stream for all prompts)
            elif share0 and i == 0:
                self._attention(L["t2i"], "t2i", qd, kd0, vd0, B, Nt, Ni, Q, Q, kv_shared=True)
            elif fused_t2i:
                self._t2i_fused(L["t2i"], pe_terms[("kops", i)], "t2i", qd, vd, B, Nt, Ni, Q, Q, x_tiled=tiled)
            elif fused_i2t:
                self._attention(L["t2i"], "t2i", qd, vd, vd, B, Nt, Ni, Q, Q, k_pe=pe_terms[("k", i)])
            else:
                self._attention(L["t2i"], "t2i", qd, kd, vd, B, Nt, Ni, Q, Q)
            ln(Q, L["n2"], B * Nt)
            # (3) MLP on the tokens
            ops.add_cast(Q, None, qd)
            ops.linear(qd, L["lin1"].w, hid[:, : L["lin1"].N], bias=L["lin1"].b, act=3)
            ops.linear(hid, L["lin2"].w, Q, bias=L["lin2"].b, residual=Q)
            ln(Q, L["n3"], B * Nt)
            # (4) the image attends to the tokens
            ops.add_cast(Q, pe2, qd)
            ops.add_cast(Q, None, td)
            last = i + 1 == len(self.layers)
            if fused_i2t:
                A, Fz = L["i2t"], L["i2t_fused"]
                k = self._buf("i2tks", (B * Nt, 128)); v = self._buf("i2tvs", (B * Nt, 128))
                ops.linear(qd, A.ks.w, k, bias=A.ks.b)
                ops.linear(td, A.vs.w, v, bias=A.vs.b)
                kvo = self._buf("i2tkv", (B, 8, 64, 2, 8))
                ops.sam_i2t_kv_operands(k, v, B, Nt, kvo)
                first0 = share0 and i == 0
                ops.sam_i2t_block(vd0 if first0 else vd, Fz["wq"], Fz["wo"], pe_terms[("q", i)], Fz["prm"], kvo, Nt, src0 if first0 else Kf, B, Ni,
                                  1e-5, None if last else Kf, vd, tiled=(4 | (0 if first0 else 3)) if tiled else 0)
                continue
            if share0 and i == 0:
                self._attention(L["i2t"], "i2t", kd0, qd, td, B, Ni, Nt, Kf, src0, q_shared=True)
            else:
                self._attention(L["i2t"], "i2t", kd, qd, td, B, Ni, Nt, Kf, Kf)
            ops.stream_copies(Kf, kd, vd, add=dense_pe_tokens, norm=L["n4"], eps=1e-5,
                              write_back=i + 1 < len(self.layers))        # (nothing reads the f32 stream after the last layer)
        ops.add_cast(Q, pe2, qd)
        if fused_t2i:
            self._t2i_fused(self.final, pe_terms[("kops", len(self.layers))], "fin", qd, vd, B, Nt, Ni, Q, Q, x_tiled=tiled)
        elif fused_i2t:
            self._attention(self.final, "fin", qd, vd, vd, B, Nt, Ni, Q, Q, k_pe=pe_terms[("k", len(self.layers))])
        else:
            self._attention(self.final, "fin", qd, kd, vd, B, Nt, Ni, Q, Q)
        ln(Q, self.nf, B * Nt)
        hs = Q.view(B, Nt, C)
        # upscaling: ConvT(2,2) -> LayerNorm2d -> GELU -> ConvT(2,2) -> GELU, written straight into the
        # pixel-shuffled layout: one batched GEMM per output-row parity (batch = (prompt, input row))
        C4, C8 = C // 4, C // 8
        G2 = 2 * G
        fused = up_fused
        u1d = self._buf("u1d", (B * 4 * Ni, _pad_to(C4, 64)))              # (pad columns stay zero)
        if fused:
            ops.sam_upscale_ln_gelu(vd, self.up0_runs, self.up0_prm, u1d, B, G, 1e-6, x_tiled=tiled)
        else:
            u1 = self._buf("u1", (B * 4 * Ni, C4), f32)
            for dy in (0, 1):
                ops.gemm(vd, self.up0[dy].w, u1, G, 2 * C4, self.up0[dy].Kp, lda=C, ldb=self.up0[dy].Kp, ldc=2 * C4, c_off=dy * G * 2 * C4,
                         nb0=B * G, sA=(G * C, 0), sC=(2 * G * 2 * C4, 0), bias=self.up0[dy].b)
            ops.layernorm_act_rows(u1, self.up_ln[0], self.up_ln[1], u1d, 1e-6, 1)
            u2 = self._buf("u2", (B * 16 * Ni, C8), f32)
            for dy in (0, 1):
                ops.gemm(u1d, self.up3[dy].w, u2, G2, 2 * C8, self.up3[dy].Kp, lda=u1d.shape[1], ldb=self.up3[dy].Kp, ldc=2 * C8,
                         c_off=dy * G2 * 2 * C8, nb0=B * G2, sA=(G2 * u1d.shape[1], 0), sC=(2 * G2 * 2 * C8, 0), bias=self.up3[dy].b, act=1)
        # hypernetwork MLPs on the four mask tokens, IoU head on the IoU token
        Kh = _pad_to(C8, 32)
        hyper = self._buf("hyper", (B, 4, Kh), f32)
        x0 = self._buf("h_in", (B, C)); h1 = self._buf("h1", (B, C)); h2 = self._buf("h2", (B, C)); h3 = self._buf("h3", (B, C8), f32)
        for m in range(4):
            ops.add_cast(hs[:, 1 + m].contiguous(), None, x0)
            ops.linear(x0, self.hyper[m][0].w, h1, bias=self.hyper[m][0].b, act=3)
            ops.linear(h1, self.hyper[m][1].w, h2, bias=self.hyper[m][1].b, act=3)
            ops.linear(h2, self.hyper[m][2].w, h3, bias=self.hyper[m][2].b)
            hyper[:, m, :C8] = h3
        ops.add_cast(hs[:, 0].contiguous(), None, x0)
        cur = x0
        for j, lin in enumerate(self.iou):
            last = j == len(self.iou) - 1
            nxt = self._buf(f"iou{j}", (B, lin.N if last else self.iou[j + 1].Kp), f32 if last else dtp)
            ops.linear(cur, lin.w, nxt[:, : lin.N], bias=lin.b, act=0 if last else 3)
            cur = nxt
        iou = cur
        # masks[b, m, pixel] = hyper[b, m, :] . upscaled[b, pixel, :]   (f32 product, K = C/8)
        P4 = 16 * Ni
        masks = self._buf("masks", (B, 4, P4), f32)
        if fused:
            ops.sam_upscale_masks(u1d, self.up3_rows, self.up3_bias, hyper, masks, B, G2)
            return masks.view(B, 4, 4 * G, 4 * G), iou
        if C8 <= 64 and C8 % 4 == 0 and B <= 65535:
            ops.hyper_product(hyper, u2, masks, B, P4, C8)
            return masks.view(B, 4, 4 * G, 4 * G), iou
        if Kh != C8:
            u2p = self._buf("u2p", (B * P4, Kh), f32)
            u2p[:, :C8] = u2
        else:
            u2p = u2
        ops.gemm(hyper, u2p, masks, 4, P4, Kh, lda=Kh, ldb=Kh, ldc=P4, nb0=B, sA=(4 * Kh, 0), sB=(P4 * Kh, 0), sC=(4 * P4, 0))
        return masks.view(B, 4, 4 * G, 4 * G), iou

    def __call__(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output: bool):
        """Reference signature (:71-110) for a [1, C, G, G] image embedding and [B, C, G, G] dense prompt embeddings."""
        C = self.C
        feats = image_embeddings.to(self.device).float().permute(0, 2, 3, 1).reshape(-1, C).contiguous()
        pe = image_pe.to(self.device).float().permute(0, 2, 3, 1).reshape(-1, C).contiguous()
        sparse = sparse_prompt_embeddings.to(self.device).float()
        dense = dense_prompt_embeddings.to(self.device).float().expand(sparse.shape[0], -1, -1, -1)
        dense_tokens = dense.permute(0, 2, 3, 1).reshape(sparse.shape[0], -1, C).contiguous()
        masks, iou = self.predict_masks(feats, pe, sparse, None, dense_tokens=dense_tokens)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, sl], iou[:, sl]


# ==========================================================================================
# Sam (sam.py:18-174)
# ==========================================================================================
class Sam:
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, cfg: SamCfg, sd: SD, device="cuda", dtype=torch.float32):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.image_encoder = ImageEncoderViT(cfg, sd, device, dtype)
        self.prompt_encoder = PromptEncoder(cfg, sd, device)
        self.mask_decoder = MaskDecoder(cfg, sd, device, dtype)
        self._tables: Dict[tuple, tuple] = {}

    def preprocess_u8(self, image_u8_chw: torch.Tensor) -> torch.Tensor:
        """Sam.preprocess (:164-174) up to the normalisation, which the patch extraction applies:
        u8 [3, h, w] (already resized) -> f32 [1, 3, S, S] whose padding holds the channel means."""
        S = self.cfg.img_size
        _, h, w = image_u8_chw.shape
        buf = torch.tensor(PIXEL_MEAN, dtype=torch.float32, device=self.device).view(1, 3, 1, 1).expand(1, 3, S, S).contiguous()
        buf[0, :, :h, :w] = image_u8_chw.to(self.device).float()
        return buf

    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        """Sam.preprocess (sam.py:164-174): normalise [.., 3, h, w] pixel values and zero-pad to S x S."""
        S = self.cfg.img_size
        x = x.to(self.device).float()
        h, w = x.shape[-2:]
        mean = torch.tensor(PIXEL_MEAN, dtype=torch.float32, device=self.device).view(-1, 1, 1)
        std = torch.tensor(PIXEL_STD, dtype=torch.float32, device=self.device).view(-1, 1, 1)
        out = torch.zeros(x.shape[:-2] + (S, S), dtype=torch.float32, device=self.device)
        out[..., :h, :w] = (x - mean) / std
        return out

    def _table(self, kind: str, *args):
        key = (kind,) + args
        t = self._tables.get(key)
        if t is None:
            fn = {"post": resample.sam_postprocess_table, "pil": lambda a, b: resample.pil_table_u8(a, b, "bilinear")}[kind]
            t = self._tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for a in fn(*args))
        return t

    def postprocess_plan(self, L: int, input_size: Tuple[int, int], original_size: Tuple[int, int]):
        """The tables of postprocess_masks together with the band plan of the kernels that evaluate it on the fly
        (ops.resample_mask_stats / resample_threshold): (tx, ty, band, max_rows), or None when no band fits in LDS."""
        S = self.cfg.img_size
        oh, ow = int(original_size[0]), int(original_size[1])
        key = ("plan", L, S, int(input_size[0]), int(input_size[1]), oh, ow)
        if key not in self._tables:
            bp = resample.band_plan(resample.sam_postprocess_table(L, S, int(input_size[0]), oh), L, ow)
            self._tables[key] = None if bp is None else (self._table("post", L, S, int(input_size[1]), ow),
                                                         self._table("post", L, S, int(input_size[0]), oh), bp[0], bp[1])
        return self._tables[key]

    def postprocess_masks(self, masks: torch.Tensor, input_size: Tuple[int, int], original_size: Tuple[int, int]) -> torch.Tensor:
        """sam.py:134-162: [B, C, 4G, 4G] logits -> [B, C, H, W] logits (both bilinear passes and the crop
        in one table per axis)."""
        B, Cm, L, _ = masks.shape
        S = self.cfg.img_size
        oh, ow = original_size
        th = self._table("post", L, S, int(input_size[0]), int(oh))
        tw = self._table("post", L, S, int(input_size[1]), int(ow))
        m = masks.contiguous().float()
        tmp = torch.empty(B * Cm, L, ow, dtype=torch.float32, device=self.device)
        out = torch.empty(B, Cm, oh, ow, dtype=torch.float32, device=self.device)
        ops.resample_axis(m, B * Cm, L, L, 1, tw, tmp)
        ops.resample_axis(tmp, B * Cm, L, ow, 0, th, out)
        return out
