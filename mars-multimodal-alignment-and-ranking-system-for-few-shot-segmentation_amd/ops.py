"""Tensor-level wrappers over the C-ABI.

Every function validates shapes and extents on the host (a kernel that reads out of
bounds can take the whole GPU down), then passes raw device pointers and the current
HIP stream to ``libmars_hip.so``.  Nothing here computes with torch.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import hip
from .hip import BF16, F32, MarsGemm

_vp = ctypes.c_void_p

# When bench.py sets GEMM_PROFILE to a list, every GEMM launch of dtype GEMM_PROFILE_DTYPE is
# bracketed by timing events and (start, end, flops, kernel variant) is appended.
GEMM_PROFILE = None
GEMM_PROFILE_DTYPE = BF16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> _vp:
    """The current stream of the current device: ``_dev`` has checked that the operands live there.  (torch's own accessor
    of the raw handle where this build has it: making a Stream object per launch was 8 of the 10 us a call spent in Python.)"""
    if _raw_stream is not None:
        return _vp(_raw_stream(torch.cuda.current_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor], byte_offset: int = 0) -> Optional[_vp]:
    if t is None:
        return None
    return _vp(t.data_ptr() + byte_offset)


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _req(cond: bool, msg: str):
    if not cond:
        raise ValueError(msg)


class _PinnedRing:
    """Small host-to-device uploads that do not stall the host.  A copy from pageable memory waits until the
    stream has run dry before it returns; from one of these pinned slots it is queued like a kernel.  A slot is
    reused only after the copy that last read it has run (its event), which with ``slots`` uploads in flight
    never blocks in practice."""

    def __init__(self, device, slots: int = 16, nbytes: int = 1 << 14):
        self.device, self.nbytes, self.i = device, nbytes, 0
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.events = [None] * slots

    def upload(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        n = t.numel() * t.element_size()
        out = torch.empty(t.shape, dtype=t.dtype, device=self.device)
        if n == 0:
            return out
        if n > self.nbytes:                      # too large for a slot: the plain (blocking) copy
            out.copy_(t)
            return out
        i, self.i = self.i, (self.i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        view = self.bufs[i][:n].view(t.dtype).view(t.shape)
        view.copy_(t)
        out.copy_(view, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev
        return out


_RINGS: dict = {}


def upload_small(t: torch.Tensor, device) -> torch.Tensor:
    """A CPU tensor of a few KiB (token ids, row indices) on ``device``, queued on the current stream
    without waiting for the work already queued there.  Tensors already on the device are returned as they are."""
    if t.is_cuda:
        return t
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ring = _RINGS.get(key)
    if ring is None:
        ring = _RINGS[key] = _PinnedRing(device)
    return ring.upload(t)


def _dev(*ts):
    """Every operand must live on the GPU that is current: the kernels are launched on the current
    device's current stream, so an operand of another GPU would be reached across devices with no
    ordering against that GPU's work.  Callers select the device first (``torch.cuda.set_device`` or a
    ``torch.cuda.device`` block; ``MARS.predict`` and the builders do)."""
    cur = None
    for t in ts:
        if t is not None:
            _req(t.is_cuda, "tensor must live on the GPU")
            if cur is None:
                cur = torch.cuda.current_device()
            if t.device.index != cur:
                raise ValueError(f"operand on cuda:{t.device.index} but the current device is cuda:{cur}: select the "
                                 f"operands' device first (torch.cuda.set_device / torch.cuda.device)")


def gemm(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, *, lda: int, ldb: int, ldc: int,
         a_off: int = 0, b_off: int = 0, c_off: int = 0, nb0: int = 1, nb1: int = 1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), bias: Optional[torch.Tensor] = None,
         gamma: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, r_off: int = 0, ldr: int = 0,
         sR=(0, 0), alpha: float = 1.0, act: int = 0, ln_out=None, ln_in=None):
    """C[M,N] (+batches) = epilogue(A[M,K] . B[N,K]^T).  Offsets and strides are in elements.
    A LayerNorm folded into the products around it: ``ln_out`` = (hi bf16 [M,N], lo bf16 [M,N], part f32
    [M,N/64,2] or None) on a product that adds to the stream kept in the two planes (part None: the result goes
    to C as f32, otherwise back into the planes), ``ln_in`` = (stat f32 [M,2], colsum f32 [N]) on the one that
    reads the hi plane in place of normalised rows."""
    _dev(A, B, C, bias, gamma, residual, *(t for t in (ln_out or ()) if t is not None), *(ln_in or ()))
    _req(A.dtype == B.dtype, "A and B must share a dtype")
    _req(M > 0 and N > 0 and K > 0, "empty gemm")
    ea = A.element_size()
    _req((K * ea) % 128 == 0, f"K={K} must span a multiple of 128 bytes")
    # extent checks: last element touched by each operand
    last_a = a_off + (nb0 - 1) * sA[0] + (nb1 - 1) * sA[1] + (M - 1) * lda + K
    last_b = b_off + (nb0 - 1) * sB[0] + (nb1 - 1) * sB[1] + (N - 1) * ldb + K
    last_c = c_off + (nb0 - 1) * sC[0] + (nb1 - 1) * sC[1] + (M - 1) * ldc + N
    _req(last_a <= _avail(A), f"gemm A overruns its buffer ({last_a} > {_avail(A)})")
    _req(last_b <= _avail(B), f"gemm B overruns its buffer ({last_b} > {_avail(B)})")
    _req(last_c <= _avail(C), f"gemm C overruns its buffer ({last_c} > {_avail(C)})")
    if bias is not None:
        _req(bias.dtype == torch.float32 and bias.numel() >= N, "bias must be f32 [N]")
    if gamma is not None:
        _req(gamma.dtype == torch.float32 and gamma.numel() >= N, "gamma must be f32 [N]")
    if residual is not None:
        _req(residual.dtype == torch.float32, "residual must be f32")
        last_r = r_off + (nb0 - 1) * sR[0] + (nb1 - 1) * sR[1] + (M - 1) * ldr + N
        _req(last_r <= _avail(residual), "gemm residual overruns its buffer")
    g = MarsGemm()
    g.A, g.B, g.C = A.data_ptr() + a_off * ea, B.data_ptr() + b_off * ea, C.data_ptr() + c_off * C.element_size()
    g.lda, g.ldb, g.ldc = lda, ldb, ldc
    g.sA0, g.sA1, g.sB0, g.sB1, g.sC0, g.sC1 = sA[0], sA[1], sB[0], sB[1], sC[0], sC[1]
    g.M, g.N, g.K, g.nb0, g.nb1 = M, N, K, nb0, nb1
    g.bias = bias.data_ptr() if bias is not None else None
    g.gamma = gamma.data_ptr() if gamma is not None else None
    g.residual = residual.data_ptr() + r_off * 4 if residual is not None else None
    g.ldr, g.sR0, g.sR1 = ldr, sR[0], sR[1]
    g.alpha, g.act, g.in_dtype, g.out_dtype = alpha, act, dt(A), dt(C)
    if ln_out is not None:
        hi, lo, part = ln_out
        for pl in (hi, lo):
            _req(pl.dtype == torch.bfloat16 and pl.dim() == 2 and pl.stride(1) == 1 and pl.shape[0] >= M and pl.shape[1] >= N, "ln_out planes must be bf16 [M,N]")
        _req(hi.stride(0) == lo.stride(0), "ln_out planes must share a row stride")
        g.ln_copy, g.ln_lo, g.ldcopy = hi.data_ptr(), lo.data_ptr(), hi.stride(0)
        if part is not None:
            _req(part.dtype == torch.float32 and part.is_contiguous() and N % 64 == 0 and part.numel() >= M * (N // 64) * 2, "ln_out part must be f32 [M,N/64,2]")
            g.ln_part = part.data_ptr()
    if ln_in is not None:
        stat, colsum = ln_in
        _req(stat.dtype == torch.float32 and stat.is_contiguous() and stat.numel() >= 2 * M, "ln_in stat must be f32 [M,2]")
        _req(colsum.dtype == torch.float32 and colsum.is_contiguous() and colsum.numel() >= N, "ln_in colsum must be f32 [N]")
        g.ln_stat, g.ln_colsum = stat.data_ptr(), colsum.data_ptr()
    if GEMM_PROFILE is not None and dt(A) == GEMM_PROFILE_DTYPE:
        rows = hip.lib().marship_gemm_split_rows(ctypes.byref(g))
        if rows < M:
            # the library would run this product as two launches: time each as a launch of its own
            old = hip.lib().marship_set_gemm_split(0)
            try:
                for r0, m in ((0, rows), (rows, M - rows)):
                    lo = (ln_out[0][r0:], ln_out[1][r0:], None if ln_out[2] is None else ln_out[2].view(-1, N // 64, 2)[r0:]) if ln_out is not None else None
                    li = (ln_in[0].view(-1, 2)[r0:], ln_in[1]) if ln_in is not None else None
                    gemm(A, B, C, m, N, K, lda=lda, ldb=ldb, ldc=ldc, a_off=a_off + r0 * lda, b_off=b_off,
                         c_off=c_off + r0 * ldc, bias=bias, gamma=gamma, residual=residual, r_off=r_off + r0 * ldr,
                         ldr=ldr, alpha=alpha, act=act, ln_out=lo, ln_in=li)
            finally:
                hip.lib().marship_set_gemm_split(old)
            return
        # bench.py: time this launch with events on the stream it runs on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(hip.lib().marship_gemm(ctypes.byref(g), stream()), "marship_gemm")
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K * nb0 * nb1, hip.lib().marship_gemm_variant_for(ctypes.byref(g))))
        return
    hip.check(hip.lib().marship_gemm(ctypes.byref(g), stream()), "marship_gemm")


def _avail(t: torch.Tensor) -> int:
    """elements available from t's first element to the end of its storage"""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


def linear(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, act: int = 0, gamma=None,
           residual: Optional[torch.Tensor] = None, alpha: float = 1.0, ln_out=None, ln_in=None):
    """out[M,N] = epilogue(x[M,K] . w[N,K]^T) for row-major 2-D views (last dim contiguous)."""
    _req(x.dim() == 2 and w.dim() == 2 and out.dim() == 2, "linear expects 2-D tensors")
    _req(x.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1, "last dim must be contiguous")
    M, K = x.shape
    N = w.shape[0]
    _req(w.shape[1] == K and out.shape[0] == M and out.shape[1] == N, f"linear shape mismatch {x.shape} {w.shape} {out.shape}")
    if residual is not None:
        _req(residual.dim() == 2 and residual.shape[0] == M and residual.shape[1] == N and residual.stride(1) == 1,
             "residual shape mismatch")
    gemm(x, w, out, M, N, K, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=bias, gamma=gamma,
         residual=residual, ldr=residual.stride(0) if residual is not None else 0, alpha=alpha, act=act,
         ln_out=ln_out, ln_in=ln_in)


def ln_finish(part: torch.Tensor, rows: int, D: int, eps: float, stat: torch.Tensor, watch: Optional[torch.Tensor] = None, watch_limit: float = 0.0):
    """stat[rows,2] = mean, rstd of the rows whose sums a stream product left in ``part`` [rows, D/64, 2]."""
    _dev(part, stat)
    _req(part.dtype == torch.float32 and stat.dtype == torch.float32 and part.is_contiguous() and stat.is_contiguous(), "ln_finish: f32 contiguous")
    _req(D % 64 == 0 and part.numel() >= rows * (D // 64) * 2 and stat.numel() >= 2 * rows, "ln_finish: sizes")
    _req(watch is None or (watch.dtype == torch.float32 and watch.numel() >= 1), "ln_finish: watch must be f32")
    if watch is not None:
        _dev(watch)
    hip.check(hip.lib().marship_ln_finish(ptr(part), D // 64, rows, D, eps, ptr(stat), ptr(watch), watch_limit, stream()), "marship_ln_finish")


def row_stats(x: torch.Tensor, eps: float, hi: torch.Tensor, lo: torch.Tensor, stat: torch.Tensor, watch: Optional[torch.Tensor] = None,
              watch_limit: float = 0.0):
    """hi = bf16(x), lo = bf16(x - hi), stat[rows,2] = mean, rstd of the f32 rows of x."""
    _dev(x, hi, lo, stat)
    _req(x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1, "row_stats: x must be f32 rows")
    rows, D = x.shape
    for pl in (hi, lo):
        _req(pl.dtype == torch.bfloat16 and pl.dim() == 2 and pl.stride(1) == 1 and pl.shape[0] >= rows and pl.shape[1] >= D, "row_stats: planes")
    _req(hi.stride(0) == lo.stride(0), "row_stats: planes must share a row stride")
    _req(stat.dtype == torch.float32 and stat.is_contiguous() and stat.numel() >= 2 * rows, "row_stats: stat")
    _req((rows - 1) * x.stride(0) + D <= _avail(x), "row_stats: overrun")
    _req(watch is None or (watch.dtype == torch.float32 and watch.numel() >= 1), "row_stats: watch must be f32")
    if watch is not None:
        _dev(watch)
    hip.check(hip.lib().marship_row_stats(ptr(x), x.stride(0), rows, D, eps, ptr(hi), ptr(lo), hi.stride(0), ptr(stat), ptr(watch), watch_limit,
                                          stream()), "marship_row_stats")


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, eps: float):
    _dev(x, w, b, out)
    _req(x.dtype == torch.float32 and x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1, "layernorm: bad x/out")
    rows, D = x.shape
    _req(out.shape[0] == rows and out.shape[1] == D and w.numel() == D and b.numel() == D, "layernorm: shape mismatch")
    _req((rows - 1) * x.stride(0) + D <= _avail(x) and (rows - 1) * out.stride(0) + D <= _avail(out), "layernorm: overrun")
    hip.check(hip.lib().marship_layernorm(ptr(x), x.stride(0), ptr(w), ptr(b), ptr(out), out.stride(0), rows, D, eps,
                                          dt(out), stream()), "marship_layernorm")


def im2col(img: torch.Tensor, p: int, out: torch.Tensor, *, mean=None, std=None, prescale: float = 1.0, col_off: int = 0):
    _dev(img, out, mean, std)
    _req(img.dim() == 4 and img.is_contiguous(), "im2col: image must be contiguous [B,C,H,W]")
    _req(img.dtype in (torch.float32, torch.uint8), "im2col: image must be f32 or u8")
    B, C, H, W = img.shape
    _req(out.dim() == 2 and out.stride(1) == 1 and out.shape[0] == B * (H // p) * (W // p), "im2col: bad out rows")
    _req(out.shape[1] >= C * p * p + col_off, "im2col: out too narrow")
    if mean is not None:
        _req(mean.numel() == C and std.numel() == C and mean.dtype == torch.float32, "im2col: mean/std size")
    hip.check(hip.lib().marship_im2col(ptr(img), 1 if img.dtype == torch.uint8 else 0, B, C, H, W, p, ptr(mean), ptr(std),
                                       prescale, ptr(out), out.stride(0), col_off, dt(out), stream()), "marship_im2col")


def embed_tokens(table: torch.Tensor, ids: torch.Tensor, pos: torch.Tensor, X: torch.Tensor):
    """X[n, l] = table[ids[n, l]] + pos[l]; table f32 [vocab, D], ids int32 [n, L], pos f32 [L, D], X f32 [n, L, D]."""
    _dev(table, ids, pos, X)
    _req(table.dim() == 2 and table.is_contiguous() and table.dtype == torch.float32, "embed_tokens: table must be contiguous f32 [vocab, D]")
    _req(ids.dim() == 2 and ids.is_contiguous() and ids.dtype == torch.int32, "embed_tokens: ids must be contiguous int32 [n, L]")
    n, L = ids.shape
    D = table.shape[1]
    _req(pos.dtype == torch.float32 and pos.is_contiguous() and tuple(pos.shape) == (L, D), "embed_tokens: pos must be contiguous f32 [L, D]")
    _req(X.dtype == torch.float32 and X.is_contiguous() and tuple(X.shape) == (n, L, D), "embed_tokens: X must be contiguous f32 [n, L, D]")
    hip.check(hip.lib().marship_embed_tokens(ptr(table), table.shape[0], ptr(ids), ptr(pos), ptr(X), n * L, L, D, stream()), "marship_embed_tokens")


def write_prefix_tokens(X: torch.Tensor, cls: torch.Tensor, pos0: Optional[torch.Tensor], regs: Optional[torch.Tensor]):
    _dev(X, cls, pos0, regs)
    _req(X.dim() == 3 and X.is_contiguous() and X.dtype == torch.float32, "prefix tokens: X must be contiguous f32 [B,T,D]")
    B, T, D = X.shape
    nr = 0 if regs is None else regs.numel() // D
    _req(cls.numel() == D and (pos0 is None or pos0.numel() >= D), "prefix tokens: cls/pos size")
    hip.check(hip.lib().marship_write_prefix_tokens(ptr(X), B, T, D, ptr(cls), ptr(pos0), ptr(regs), nr, stream()),
              "marship_write_prefix_tokens")


def transpose_raw(src: torch.Tensor, s_off: int, ldi: int, sI, dst: torch.Tensor, d_off: int, ldo: int, sO, rows: int,
                  cols: int, nb0: int, nb1: int):
    _dev(src, dst)
    _req(src.element_size() == dst.element_size() and src.element_size() in (2, 4), "transpose: element size")
    last_i = s_off + (nb0 - 1) * sI[0] + (nb1 - 1) * sI[1] + (rows - 1) * ldi + cols
    last_o = d_off + (nb0 - 1) * sO[0] + (nb1 - 1) * sO[1] + (cols - 1) * ldo + rows
    _req(last_i <= _avail(src) and last_o <= _avail(dst), "transpose: overrun")
    es = src.element_size()
    hip.check(hip.lib().marship_transpose(ptr(src, s_off * es), ldi, sI[0], sI[1], ptr(dst, d_off * es), ldo, sO[0], sO[1],
                                          rows, cols, nb0, nb1, es, stream()), "marship_transpose")


def softmax_rows(S: torch.Tensor, P: torch.Tensor, nb: int, T: int, ld: int, causal: bool):
    _dev(S, P)
    _req(S.dtype == torch.float32 and nb * T * ld <= _avail(S) and nb * T * ld <= _avail(P), "softmax_rows: overrun")
    hip.check(hip.lib().marship_softmax_rows(ptr(S), ld, ptr(P), ld, nb, T, 1 if causal else 0, dt(P), stream()),
              "marship_softmax_rows")


def attn_mean_accum(P: torch.Tensor, p_off: int, ldp: int, head_stride: int, H: int, off: int, Pn: int, scale: float,
                    acc: torch.Tensor):
    _dev(P, acc)
    _req(acc.dtype == torch.float32 and acc.dim() == 2 and acc.stride(1) == 1 and acc.shape[0] >= Pn and acc.shape[1] >= Pn,
         "attn_mean_accum: bad accumulator")
    _req(p_off + (H - 1) * head_stride + (off + Pn - 1) * ldp + off + Pn <= _avail(P), "attn_mean_accum: overrun")
    hip.check(hip.lib().marship_attn_mean_accum(ptr(P, p_off * P.element_size()), ldp, head_stride, H, off, Pn, scale,
                                                ptr(acc), acc.stride(0), dt(P), stream()), "marship_attn_mean_accum")


def swiglu(x12: torch.Tensor, out: torch.Tensor):
    _dev(x12, out)
    _req(x12.dim() == 2 and x12.is_contiguous() and out.dim() == 2 and out.stride(1) == 1 and x12.dtype == out.dtype, "swiglu: layout")
    M, H2 = x12.shape
    _req(out.shape[0] == M and out.shape[1] >= H2 // 2, "swiglu: shape")
    hip.check(hip.lib().marship_swiglu(ptr(x12), ptr(out), out.stride(0), M, H2 // 2, dt(out), stream()), "marship_swiglu")


def cast(src: torch.Tensor, dst: torch.Tensor):
    _dev(src, dst)
    _req(src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel(), "cast: layout")
    hip.check(hip.lib().marship_cast(ptr(src), ptr(dst), src.numel(), dt(src), dt(dst), stream()), "marship_cast")


# What raw queries lack for the softmax to be 2^(scale * q.k): log2(e) / sqrt(64).  Callers whose qkv product has
# already scaled the queries by it (one rounding instead of two) pass 1.0.
RAW_QUERY_SCALE_LOG2 = 0.125 * 1.4426950408889634


def flash_attn(qkv: torch.Tensor, O: torch.Tensor, lse: Optional[torch.Tensor], B: int, T: int, H: int, causal: bool,
               score_scale_log2: float = RAW_QUERY_SCALE_LOG2):
    _dev(qkv, O, lse)
    D = H * 64
    _req(qkv.dtype == torch.bfloat16 and O.dtype == torch.bfloat16, "flash_attn: bf16 only")
    _req(B * T * 3 * D <= _avail(qkv) and B * T * D <= _avail(O), "flash_attn: overrun")
    if lse is not None:
        _req(lse.dtype == torch.float32 and B * H * T <= _avail(lse), "flash_attn: lse overrun")
    hip.check(hip.lib().marship_flash_attn(ptr(qkv), ptr(O), ptr(lse), B, T, H, 1 if causal else 0, float(score_scale_log2), stream()),
              "marship_flash_attn")


def flash_attn_fp8(qkv: torch.Tensor, O: torch.Tensor, B: int, T: int, H: int, ws: dict, score_scale_log2: float = RAW_QUERY_SCALE_LOG2):
    """Attention with fp8 (e4m3) operands on packed bf16 qkv.  ``ws`` keeps the quantised operands between calls."""
    _dev(qkv, O)
    D = H * 64
    _req(qkv.dtype == torch.bfloat16 and O.dtype == torch.bfloat16, "flash_attn_fp8: bf16 qkv and output")
    _req(B * T * 3 * D <= _avail(qkv) and B * T * D <= _avail(O), "flash_attn_fp8: overrun")
    Tp = (T + 31) // 32 * 32
    key = (B, T, H)
    if ws.get("key") != key:
        n = B * H * Tp * 64
        ws.update(key=key, q8=torch.zeros(n, dtype=torch.uint8, device=qkv.device), k8=torch.zeros(n, dtype=torch.uint8, device=qkv.device),
                  vt8=torch.zeros(n, dtype=torch.uint8, device=qkv.device), scales=torch.zeros(B * H * 6, dtype=torch.float32, device=qkv.device))
    hip.check(hip.lib().marship_flash_attn_fp8(ptr(qkv), ptr(O), B, T, H, ptr(ws["q8"]), ptr(ws["k8"]), ptr(ws["vt8"]), ptr(ws["scales"]),
                                               Tp, float(score_scale_log2), stream()), "marship_flash_attn_fp8")


def flash_attn_mean(qkv: torch.Tensor, lse: torch.Tensor, T: int, H: int, off: int, Pn: int, scale: float,
                    acc: torch.Tensor, score_scale_log2: float = RAW_QUERY_SCALE_LOG2):
    _dev(qkv, lse, acc)
    _req(qkv.dtype == torch.bfloat16 and T * 3 * H * 64 <= _avail(qkv) and H * T <= _avail(lse), "flash_attn_mean: overrun")
    _req(acc.dtype == torch.float32 and acc.dim() == 2 and acc.stride(1) == 1 and acc.shape[0] >= Pn and acc.shape[1] >= Pn,
         "flash_attn_mean: bad accumulator")
    hip.check(hip.lib().marship_flash_attn_mean(ptr(qkv), ptr(lse), T, H, off, Pn, scale, ptr(acc), acc.stride(0), float(score_scale_log2), stream()),
              "marship_flash_attn_mean")


def l2norm_rows(x: torch.Tensor, row_off: int, rows: int, out: torch.Tensor):
    """x f32 [B,T,D] -> out [B*rows, D]: rows row_off.. of every image, unit length."""
    _dev(x, out)
    _req(x.dim() == 3 and x.is_contiguous() and x.dtype == torch.float32, "l2norm_rows: x must be contiguous f32 [B,T,D]")
    B, T, D = x.shape
    _req(row_off + rows <= T and out.dim() == 2 and out.shape[0] == B * rows and out.shape[1] >= D and out.stride(1) == 1,
         "l2norm_rows: shape")
    hip.check(hip.lib().marship_l2norm_rows(ptr(x), T * D, D, row_off, B, rows, D, ptr(out), out.stride(0), dt(out), stream()),
              "marship_l2norm_rows")


def vva_prior(S: torch.Tensor, fg: torch.Tensor, P: int, ws: torch.Tensor, chunks: int, out: torch.Tensor):
    _dev(S, fg, ws, out)
    R = S.shape[0]
    _req(S.dtype == torch.float32 and S.dim() == 2 and S.stride(1) == 1 and S.shape[1] >= P, "vva_prior: S layout")
    _req(fg.dtype == torch.uint8 and fg.numel() == R and ws.numel() >= 4 * chunks * P and out.numel() >= P, "vva_prior: sizes")
    hip.check(hip.lib().marship_vva_prior(ptr(S), S.stride(0), ptr(fg), R, P, ptr(ws), chunks, ptr(out), stream()),
              "marship_vva_prior")


def minmax(x: torch.Tensor, y: torch.Tensor):
    _dev(x, y)
    _req(x.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous() and y.numel() == x.numel(), "minmax: layout")
    hip.check(hip.lib().marship_minmax(ptr(x), ptr(y), x.numel(), stream()), "marship_minmax")


def box_mask(prior: torch.Tensor, threshold: float, out: torch.Tensor):
    _dev(prior, out)
    G = prior.shape[0]
    _req(prior.dim() == 2 and prior.shape[1] == G and prior.is_contiguous() and prior.dtype == torch.float32, "box_mask: prior")
    _req(out.numel() == G * G and out.dtype == torch.float32, "box_mask: out")
    hip.check(hip.lib().marship_box_mask(ptr(prior), G, float(threshold), ptr(out), stream()), "marship_box_mask")


def sinkhorn1(A: torch.Tensor, D: torch.Tensor, P: int, scratch: torch.Tensor):
    _dev(A, D, scratch)
    _req(A.dtype == torch.float32 and A.dim() == 2 and A.stride(1) == 1 and A.shape[0] >= P and A.shape[1] >= P, "sinkhorn1: A")
    _req(D.dim() == 2 and D.stride(1) == 1 and D.shape[0] >= P and D.stride(0) == D.shape[1], "sinkhorn1: D")
    _req(scratch.numel() >= 33 * P, "sinkhorn1: scratch needs 33*P floats")
    hip.check(hip.lib().marship_sinkhorn1(ptr(A), A.stride(0), ptr(D), D.stride(0), P, ptr(scratch), stream()),
              "marship_sinkhorn1")


def max_and_transpose(D: torch.Tensor, DDT: torch.Tensor, P: int, R: torch.Tensor, RT: torch.Tensor):
    _dev(D, DDT, R, RT)
    ld = D.stride(0)
    for t in (D, DDT, R, RT):
        _req(t.dtype == torch.float32 and t.dim() == 2 and t.stride(0) == ld and t.shape[0] >= P and t.shape[1] == ld,
             "max_and_transpose: all four matrices must be [>=P, ld]")
    hip.check(hip.lib().marship_max_and_transpose(ptr(D), ptr(DDT), ld, P, ptr(R), ptr(RT), stream()),
              "marship_max_and_transpose")


def masked_matvec(R2: torch.Tensor, Bm: torch.Tensor, prior: torch.Tensor, P: int, out: torch.Tensor):
    _dev(R2, Bm, prior, out)
    _req(R2.dim() == 2 and R2.shape[0] >= P and R2.shape[1] >= P and Bm.numel() >= P and prior.numel() >= P and out.numel() >= P,
         "masked_matvec: sizes")
    hip.check(hip.lib().marship_masked_matvec(ptr(R2), R2.stride(0), ptr(Bm), ptr(prior), P, ptr(out), stream()),
              "marship_masked_matvec")


def resample_axis(src: torch.Tensor, planes: int, H: int, W: int, axis: int, table, dst: torch.Tensor, *,
                  quant_u8: bool = False, shift: float = 0.0, scale: float = 1.0):
    """table = (xmin i32 [nout], xsize i32 [nout], weights f32 [nout, kmax]) on the device."""
    xmin, xsize, wts = table
    _dev(src, dst, xmin, xsize, wts)
    nout, kmax = wts.shape
    _req(src.dtype == torch.float32 and planes * H * W <= _avail(src), "resample_axis: src overrun")
    total = planes * (nout if axis == 0 else H) * (nout if axis == 1 else W)
    _req(total <= _avail(dst), "resample_axis: dst overrun")
    hip.check(hip.lib().marship_resample_axis(ptr(src), planes, H, W, axis, nout, ptr(xmin), ptr(xsize), ptr(wts), kmax,
                                              1 if quant_u8 else 0, shift, scale, ptr(dst), dt(dst), stream()),
              "marship_resample_axis")


def resample_axis_u8(src: torch.Tensor, planes: int, H: int, W: int, axis: int, table, dst: torch.Tensor):
    xmin, xsize, kk = table
    _dev(src, dst, xmin, xsize, kk)
    nout, kmax = kk.shape
    _req(src.dtype == torch.uint8 and dst.dtype == torch.uint8 and planes * H * W <= _avail(src), "resample_axis_u8: src")
    total = planes * (nout if axis == 0 else H) * (nout if axis == 1 else W)
    _req(total <= _avail(dst), "resample_axis_u8: dst overrun")
    hip.check(hip.lib().marship_resample_axis_u8(ptr(src), planes, H, W, axis, nout, ptr(xmin), ptr(xsize), ptr(kk), kmax,
                                                 ptr(dst), stream()), "marship_resample_axis_u8")


def f32_to_u8(src: torch.Tensor, dst: torch.Tensor):
    _dev(src, dst)
    _req(src.dtype == torch.float32 and dst.dtype == torch.uint8 and src.is_contiguous() and dst.numel() == src.numel(), "f32_to_u8")
    hip.check(hip.lib().marship_f32_to_u8(ptr(src), ptr(dst), src.numel(), stream()), "marship_f32_to_u8")


def pool_masks(masks: torch.Tensor, G: int, bits: torch.Tensor):
    _dev(masks, bits)
    _req(masks.dim() == 3 and masks.is_contiguous() and masks.dtype == torch.float32 and masks.shape[1] == masks.shape[2],
         "pool_masks: masks must be contiguous f32 [N,S,S]")
    N, S, _ = masks.shape
    _req(bits.dtype == torch.uint8 and bits.numel() >= N * G * G, "pool_masks: bits")
    hip.check(hip.lib().marship_pool_masks(ptr(masks), N, S, G, ptr(bits), stream()), "marship_pool_masks")


def proposal_stats(bits, N, P, vva, vta, cnt, svv, svt, ucnt):
    _dev(bits, vva, vta, cnt, svv, svt, ucnt)
    _req(bits.numel() >= N * P and vva.numel() >= P and vta.numel() >= P and cnt.numel() >= N and svv.numel() >= N and svt.numel() >= N,
         "proposal_stats: sizes")
    _req(cnt.dtype == torch.int32 and ucnt.dtype == torch.int32 and vva.dtype == torch.float32 and vta.dtype == torch.float32,
         "proposal_stats: dtypes")
    hip.check(hip.lib().marship_proposal_stats(ptr(bits), N, P, ptr(vva), ptr(vta), ptr(cnt), ptr(svv), ptr(svt), ptr(ucnt),
                                               stream()), "marship_proposal_stats")


def cosine_scores(feats: torch.Tensor, text: torch.Tensor, scores: torch.Tensor):
    _dev(feats, text, scores)
    _req(feats.dim() == 2 and feats.stride(1) == 1 and feats.dtype == torch.float32 and text.dtype == torch.float32, "cosine_scores")
    N, E = feats.shape
    _req(text.numel() >= E and scores.numel() >= N, "cosine_scores: sizes")
    hip.check(hip.lib().marship_cosine_scores(ptr(feats), feats.stride(0), N, E, ptr(text), ptr(scores), stream()),
              "marship_cosine_scores")


def cosine_table(feats: torch.Tensor, text: torch.Tensor, out: torch.Tensor, scale: float = 1.0):
    """out[i, p] = scale * cos(feats[i], text[p])."""
    _dev(feats, text, out)
    _req(feats.dim() == 2 and text.dim() == 2 and feats.stride(1) == 1 and text.is_contiguous() and out.is_contiguous(), "cosine_table: layout")
    _req(feats.dtype == torch.float32 and text.dtype == torch.float32 and out.dtype == torch.float32, "cosine_table: dtypes")
    N, E = feats.shape
    Pn = text.shape[0]
    _req(text.shape[1] == E and out.numel() >= N * Pn, "cosine_table: sizes")
    hip.check(hip.lib().marship_cosine_table(ptr(feats), feats.stride(0), N, E, ptr(text), Pn, float(scale), ptr(out), stream()),
              "marship_cosine_table")


def points_in_masks(masks: torch.Tensor, points: torch.Tensor, counts: torch.Tensor):
    """counts[k] = number of (x, y) points whose pixel of mask k is set."""
    _dev(masks, points, counts)
    _req(masks.dim() == 3 and masks.is_contiguous() and masks.dtype == torch.float32, "points_in_masks: masks must be f32 [n,H,W]")
    _req(points.dim() == 2 and points.shape[1] == 2 and points.is_contiguous() and points.dtype == torch.float32, "points_in_masks: points")
    _req(counts.dtype == torch.int32 and counts.numel() >= masks.shape[0], "points_in_masks: counts")
    n, H, W = masks.shape
    hip.check(hip.lib().marship_points_in_masks(ptr(masks), n, H, W, ptr(points), points.shape[0], ptr(counts), stream()),
              "marship_points_in_masks")


def rank(emd, aclip, cnt, svv, svt, ucnt, N, alpha, sthr, dthr, score, order, keep):
    _dev(emd, aclip, cnt, svv, svt, ucnt, score, order, keep)
    _req(emd.dtype == torch.float64 and score.dtype == torch.float64 and order.dtype == torch.int32 and keep.dtype == torch.uint8,
         "rank: dtypes")
    for t in (emd, aclip, cnt, svv, svt, score, order, keep):
        _req(t.numel() >= N, "rank: sizes")
    hip.check(hip.lib().marship_rank(ptr(emd), ptr(aclip), ptr(cnt), ptr(svv), ptr(svt), ptr(ucnt), N, alpha, sthr, dthr,
                                     ptr(score), ptr(order), ptr(keep), stream()), "marship_rank")


def merge_masks(masks: torch.Tensor, keep: torch.Tensor, merged: torch.Tensor):
    _dev(masks, keep, merged)
    _req(masks.is_contiguous() and masks.dtype == torch.float32 and merged.dtype == torch.float32, "merge_masks")
    N = masks.shape[0]
    HW = masks.numel() // N
    _req(merged.numel() == HW and keep.numel() >= N, "merge_masks: sizes")
    hip.check(hip.lib().marship_merge_masks(ptr(masks), ptr(keep), N, HW, ptr(merged), stream()), "marship_merge_masks")


def iou_counts(pred: torch.Tensor, gt: torch.Tensor, ignore: Optional[torch.Tensor], counts: torch.Tensor):
    _dev(pred, gt, ignore, counts)
    _req(pred.is_contiguous() and gt.is_contiguous() and pred.numel() == gt.numel() and pred.dtype == torch.float32 and gt.dtype == torch.float32,
         "iou_counts: layout")
    _req(counts.dtype == torch.float32 and counts.numel() >= 4, "iou_counts: counts")
    if ignore is not None:
        _req(ignore.is_contiguous() and ignore.numel() == pred.numel() and ignore.dtype == torch.float32, "iou_counts: ignore")
    hip.check(hip.lib().marship_iou_counts(ptr(pred), ptr(gt), ptr(ignore), pred.numel(), ptr(counts), stream()),
              "marship_iou_counts")


def layernorm_bwd(x, w, dy, dy_stride, dy_scale, skip_rows, add, dx, eps):
    _dev(x, w, dy, add, dx)
    _req(x.dim() == 2 and x.is_contiguous() and dx.is_contiguous() and dx.shape == x.shape and x.dtype == torch.float32, "ln_bwd")
    rows, D = x.shape
    _req(w.numel() == D and (rows - 1) * dy_stride + D <= _avail(dy), "ln_bwd: sizes")
    if add is not None:
        _req(add.is_contiguous() and add.shape == x.shape, "ln_bwd: add")
    hip.check(hip.lib().marship_layernorm_bwd(ptr(x), ptr(w), ptr(dy), dy_stride, dy_scale, skip_rows, ptr(add), ptr(dx),
                                              rows, D, eps, stream()), "marship_layernorm_bwd")


def quickgelu_bwd(dh, pre, out):
    _dev(dh, pre, out)
    _req(dh.is_contiguous() and pre.is_contiguous() and out.is_contiguous() and dh.numel() == pre.numel() == out.numel(), "qg_bwd")
    hip.check(hip.lib().marship_quickgelu_bwd(ptr(dh), ptr(pre), ptr(out), dh.numel(), stream()), "marship_quickgelu_bwd")


def softmax_bwd(P, dP, dS, ld, nb, T, scale):
    _dev(P, dP, dS)
    _req(nb * T * ld <= min(_avail(P), _avail(dP), _avail(dS)) and P.dtype == torch.float32, "softmax_bwd: overrun")
    hip.check(hip.lib().marship_softmax_bwd(ptr(P), ptr(dP), ptr(dS), ld, nb, T, scale, stream()), "marship_softmax_bwd")


def col_mean(x, r0, r1, out):
    _dev(x, out)
    _req(x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and r1 <= x.shape[0] and out.numel() >= x.shape[1], "col_mean")
    hip.check(hip.lib().marship_col_mean(ptr(x), x.stride(0), r0, r1, x.shape[1], ptr(out), stream()), "marship_col_mean")


def cam_head(f, proj, text, ls_exp, logits, df):
    _dev(f, proj, text, logits, df)
    D, E = proj.shape
    _req(proj.is_contiguous() and f.numel() >= D and text.numel() >= 2 * E and logits.numel() >= 2 and df.numel() >= D, "cam_head")
    hip.check(hip.lib().marship_cam_head(ptr(f), ptr(proj), ptr(text), D, E, ls_exp, ptr(logits), ptr(df), stream()),
              "marship_cam_head")


def cam_rows(act, w, r0, rows, cam):
    _dev(act, w, cam)
    _req(act.dim() == 2 and act.stride(1) == 1 and r0 + rows <= act.shape[0] and w.numel() >= act.shape[1] and cam.numel() >= rows,
         "cam_rows")
    hip.check(hip.lib().marship_cam_rows(ptr(act), act.stride(0), ptr(w), r0, rows, act.shape[1], ptr(cam), stream()),
              "marship_cam_rows")


def emd_scores_host(S_host: torch.Tensor, fg_host: torch.Tensor, bits_host: torch.Tensor, threads: int = 0,
                    is_cost: bool = False) -> torch.Tensor:
    """Host routine: S f32 [R, ld] (CPU), fg u8 [R], bits u8 [N, P] -> f64 [N] scores 1 - emd."""
    _req(not S_host.is_cuda and S_host.dtype == torch.float32 and S_host.dim() == 2 and S_host.stride(1) == 1, "emd: S must be CPU f32")
    R = S_host.shape[0]
    N, P = bits_host.shape
    _req(S_host.shape[1] >= P and fg_host.numel() == R and fg_host.dtype == torch.uint8 and bits_host.dtype == torch.uint8
         and bits_host.is_contiguous() and fg_host.is_contiguous(), "emd: sizes")
    out = torch.empty(N, dtype=torch.float64)
    hip.check(hip.lib().marshost_emd_scores(_vp(S_host.data_ptr()), S_host.stride(0), R, P, _vp(fg_host.data_ptr()),
                                            _vp(bits_host.data_ptr()), N, 1 if is_cost else 0, _vp(out.data_ptr()), threads), "marshost_emd_scores")
    return out


# ---------------------------------------------------------------- Segment-Anything proposal stage
def gather_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor, *, src_col: int = 0, dst_col: int = 0, ncols: Optional[int] = None):
    """dst[r, dst_col:dst_col+ncols] = src[idx[r], src_col:src_col+ncols], zeros where idx[r] < 0."""
    _dev(src, idx, dst)
    _req(src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1 and src.dtype == dst.dtype, "gather_rows: layout")
    _req(src.dtype in (torch.float32, torch.bfloat16) and idx.dtype == torch.int32 and idx.is_contiguous(), "gather_rows: dtypes")
    rows = idx.numel()
    ncols = ncols if ncols is not None else min(src.shape[1] - src_col, dst.shape[1] - dst_col)
    _req(rows <= dst.shape[0] and src_col + ncols <= src.shape[1] and dst_col + ncols <= dst.shape[1], "gather_rows: sizes")
    hip.check(hip.lib().marship_gather_rows(ptr(src), src.stride(0), src_col, ptr(idx), ptr(dst), dst.stride(0), dst_col, rows, ncols,
                                            dt(src), stream()), "marship_gather_rows")


def attn_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, B: int, H: int, hd: int, Tq: int, Tk: int, scale: float,
               q_shared: bool = False, kv_shared: bool = False):
    """Attention on unpadded heads of 16 or 32 dims where one side has at most 16 tokens: q [B*Tq, >= H*hd], k / v [B*Tk, ...],
    out [B*Tq, ...], rows may be wider than H*hd (row stride = stride(0)).  ``q_shared`` / ``kv_shared``: that operand holds one
    set of rows ([Tq, ...] / [Tk, ...]) used by every prompt."""
    _dev(q, k, v, out)
    _req(q.dtype == k.dtype == v.dtype == out.dtype and q.dtype in (torch.float32, torch.bfloat16), "attn_small: one dtype, f32 or bf16")
    for t, rows in ((q, Tq if q_shared else B * Tq), (k, Tk if kv_shared else B * Tk), (v, Tk if kv_shared else B * Tk), (out, B * Tq)):
        _req(t.dim() == 2 and t.stride(1) == 1 and t.shape[0] >= rows and t.shape[1] >= H * hd, "attn_small: operand layout")
    hip.check(hip.lib().marship_attn_small(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), ptr(out), out.stride(0),
                                           B, H, hd, Tq, Tk, scale, int(q_shared) | 2 * int(kv_shared), dt(q), stream()), "marship_attn_small")


def _row_map(what: str, in_rows, pad_row, n: int, width: int):
    _req((in_rows is None) == (pad_row is None), what + ": in_rows and pad_row come together")
    if in_rows is not None:
        _req(in_rows.dtype == torch.int32 and in_rows.is_contiguous() and in_rows.numel() >= n, what + ": in_rows must be i32, one entry a token row")
        _req(pad_row.dtype == torch.bfloat16 and pad_row.is_contiguous() and pad_row.numel() >= width, what + ": pad_row must be one bf16 row of qkv")


def relpos_bias(qkv: torch.Tensor, B: int, T: int, H: int, hd: int, hdp: int, gw: int, Rh: torch.Tensor, Rw: torch.Tensor,
                rel_h: torch.Tensor, rel_w: torch.Tensor, in_rows: Optional[torch.Tensor] = None, pad_row: Optional[torch.Tensor] = None):
    """Decomposed relative-position terms of B images (windows) of T tokens.  With ``in_rows`` (i32 on the device, the caller's own
    map: every entry negative or a row of ``qkv``) token row r is row in_rows[r] of qkv, or ``pad_row`` for a negative entry."""
    _dev(qkv, Rh, Rw, rel_h, rel_w, in_rows, pad_row)
    _req(qkv.dim() == 2 and qkv.stride(1) == 1 and (in_rows is not None or qkv.shape[0] >= B * T), "relpos_bias: qkv")
    _req(Rh.dtype == torch.float32 and Rw.dtype == torch.float32 and Rh.is_contiguous() and Rw.is_contiguous(), "relpos_bias: tables")
    gh = T // gw
    Hk, Wk = Rh.shape[1], Rw.shape[1]
    _req(tuple(Rh.shape) == (gh, Hk, hd) and tuple(Rw.shape) == (gw, Wk, hd), "relpos_bias: table shapes")
    _req(rel_h.dtype == torch.float32 and rel_w.dtype == torch.float32 and rel_h.numel() >= B * H * T * Hk and rel_w.numel() >= B * H * T * Wk,
         "relpos_bias: outputs")
    _row_map("relpos_bias", in_rows, pad_row, B * T, 3 * H * hdp)
    if in_rows is not None:
        hip.check(hip.lib().marship_relpos_bias_rows(ptr(qkv), qkv.stride(0), B, T, H, hd, hdp, gw, Hk, Wk, ptr(Rh), ptr(Rw), ptr(rel_h), ptr(rel_w),
                                                     ptr(in_rows), ptr(pad_row), dt(qkv), stream()), "marship_relpos_bias_rows")
        return
    hip.check(hip.lib().marship_relpos_bias(ptr(qkv), qkv.stride(0), B, T, H, hd, hdp, gw, Hk, Wk, ptr(Rh), ptr(Rw), ptr(rel_h), ptr(rel_w),
                                            dt(qkv), stream()), "marship_relpos_bias")


def sam_attn(qkv: torch.Tensor, out: torch.Tensor, rel_h: torch.Tensor, rel_w: torch.Tensor, nb: int, T: int, H: int, hdp: int,
             Hk: int, Wk: int, scale: float, in_rows: Optional[torch.Tensor] = None, pad_row: Optional[torch.Tensor] = None,
             out_rows: Optional[torch.Tensor] = None):
    """out[nb*T, H*hdp] = attention of packed bf16 qkv rows with the decomposed relative-position bias, fused.  Row maps (i32 on
    the device, one entry a token row, the caller's own: every entry negative or a row of ``qkv`` / ``out``): token row r is row
    in_rows[r] of qkv, or ``pad_row`` (one row laid out like a row of qkv) for a negative entry; its result goes to
    out[out_rows[r]], or nowhere for a negative entry."""
    _dev(qkv, out, rel_h, rel_w, in_rows, pad_row, out_rows)
    _req(qkv.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and qkv.dim() == 2 and out.dim() == 2, "sam_attn: bf16 rows")
    _req(qkv.stride(1) == 1 and out.stride(1) == 1 and (in_rows is not None or qkv.shape[0] >= nb * T)
         and (out_rows is not None or out.shape[0] >= nb * T), "sam_attn: rows")
    _req(out_rows is None or (out_rows.dtype == torch.int32 and out_rows.is_contiguous() and out_rows.numel() >= nb * T),
         "sam_attn: out_rows must be i32, one entry a token row")
    _row_map("sam_attn", in_rows, pad_row, nb * T, 3 * H * hdp)
    _req(qkv.shape[1] >= 3 * H * hdp and out.shape[1] >= H * hdp, "sam_attn: columns")
    _req(rel_h.dtype == torch.float32 and rel_w.dtype == torch.float32 and rel_h.numel() >= nb * H * T * Hk and rel_w.numel() >= nb * H * T * Wk,
         "sam_attn: bias terms")
    if in_rows is not None or out_rows is not None:
        hip.check(hip.lib().marship_sam_attn_rows(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(rel_h), ptr(rel_w), nb, T, H, hdp, Hk, Wk,
                                                  scale, ptr(in_rows), ptr(pad_row), ptr(out_rows), stream()), "marship_sam_attn_rows")
        return
    hip.check(hip.lib().marship_sam_attn(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(rel_h), ptr(rel_w), nb, T, H, hdp, Hk, Wk,
                                         scale, stream()), "marship_sam_attn")


def softmax_bias_rows(S: torch.Tensor, P: torch.Tensor, nrows: int, T: int, rel_h=None, rel_w=None, Hk: int = 0, Wk: int = 0):
    _dev(S, P, rel_h, rel_w)
    _req(S.dtype == torch.float32 and S.stride(-1) == 1 and P.stride(-1) == 1, "softmax_bias_rows: layout")
    lds_, ldp = S.stride(-2), P.stride(-2)
    _req((nrows - 1) * lds_ + T <= _avail(S) and nrows * ldp <= _avail(P), "softmax_bias_rows: overrun")
    hip.check(hip.lib().marship_softmax_bias_rows(ptr(S), lds_, ptr(P), ldp, nrows, T, ptr(rel_h), ptr(rel_w), Hk, Wk, dt(P), stream()),
              "marship_softmax_bias_rows")


def fourier_pe(coords: torch.Tensor, G: torch.Tensor, out: torch.Tensor, labels: Optional[torch.Tensor] = None,
               table: Optional[torch.Tensor] = None):
    """out [n, 2F] from coords [n, 2] in [0,1]; with labels the point-prompt embedding rule (see the header)."""
    _dev(coords, G, out, labels, table)
    _req(coords.dtype == torch.float32 and coords.is_contiguous() and coords.shape[-1] == 2, "fourier_pe: coords")
    _req(G.dtype == torch.float32 and G.is_contiguous() and G.shape[0] == 2 and out.dtype == torch.float32 and out.is_contiguous(), "fourier_pe: G/out")
    n, F_ = coords.numel() // 2, G.shape[1]
    _req(out.numel() >= n * 2 * F_, "fourier_pe: out size")
    if labels is not None:
        _req(labels.dtype == torch.int32 and labels.numel() == n and table is not None and table.dtype == torch.float32
             and table.is_contiguous() and table.numel() == 5 * 2 * F_, "fourier_pe: labels/table (5 rows: not-a-point, "
             "negative, positive, box corner 0, box corner 1)")
    hip.check(hip.lib().marship_fourier_pe(ptr(coords), ptr(G), F_, ptr(labels), ptr(table), ptr(out), 2 * F_, n, stream()),
              "marship_fourier_pe")


def add_cast(a: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor):
    """out = a + b (b broadcast over leading rows by repetition), f32 -> out.dtype; a, out [rows, D] contiguous."""
    _dev(a, b, out)
    _req(a.dtype == torch.float32 and a.is_contiguous() and out.is_contiguous() and out.numel() == a.numel(), "add_cast: a/out")
    D = a.shape[-1]
    rows = a.numel() // D
    b_rows = 1
    if b is not None:
        _req(b.dtype == torch.float32 and b.is_contiguous() and b.shape[-1] == D and rows % (b.numel() // D) == 0, "add_cast: b")
        b_rows = b.numel() // D
    hip.check(hip.lib().marship_add_cast(ptr(a), ptr(b), b_rows, ptr(out), rows, D, dt(out), stream()), "marship_add_cast")


def stream_copies(x: torch.Tensor, c0: Optional[torch.Tensor], c1: Optional[torch.Tensor], *, add: Optional[torch.Tensor] = None, norm=None,
                  eps: float = 1e-5, write_back: bool = True):
    """One pass over x f32 [rows, D]: y = LayerNorm(x) with ``norm`` = (weight, bias) (written back into x when
    ``write_back``) or y = x; c0 = y + add (add broadcast over leading rows by repetition), c1 = y, both in their dtype."""
    _dev(x, c0, c1, add)
    _req(x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2, "stream_copies: x must be contiguous f32 [rows, D]")
    rows, D = x.shape
    outs = [t for t in (c0, c1) if t is not None]
    _req(len(outs) > 0 or (norm is not None and write_back), "stream_copies: nothing to write")
    for t in outs:
        _req(t.is_contiguous() and t.numel() == x.numel() and t.dtype == outs[0].dtype and t.dtype in (torch.float32, torch.bfloat16), "stream_copies: copies")
    add_rows = 1
    if add is not None:
        _req(add.dtype == torch.float32 and add.is_contiguous() and add.shape[-1] == D and rows % (add.numel() // D) == 0, "stream_copies: add")
        add_rows = add.numel() // D
    w = b = None
    if norm is not None:
        w, b = norm
        _dev(w, b)
        _req(w.numel() == D and b.numel() == D and w.dtype == torch.float32 and b.dtype == torch.float32, "stream_copies: norm")
    hip.check(hip.lib().marship_stream_copies(ptr(x), ptr(w), ptr(b), eps, ptr(x) if (norm is not None and write_back) else None, ptr(add), add_rows,
                                              ptr(c0), ptr(c1), rows, D, dt(outs[0]) if outs else 0, stream()), "marship_stream_copies")


def layernorm_act_rows(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, eps: float, act: int):
    """out[:, :D] = act(LayerNorm(x)) for x f32 [rows, D <= 256]; out [rows, >= D] in its dtype (row stride = stride(0))."""
    _dev(x, w, b, out)
    _req(x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2 and out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= x.shape[0]
         and out.shape[1] >= x.shape[1], "layernorm_act_rows: layout")
    hip.check(hip.lib().marship_layernorm_act_rows(ptr(x), ptr(w), ptr(b), eps, ptr(out), out.stride(0), x.shape[0], x.shape[1], act, dt(out), stream()),
              "marship_layernorm_act_rows")


def sam_upscale_ln_gelu(x: torch.Tensor, w0: torch.Tensor, prm: torch.Tensor, out: torch.Tensor, prompts: int, G: int, eps: float = 1e-6,
                        x_tiled: bool = False):
    """ConvTranspose2d(256, 64, 2, 2) + LayerNorm2d + GELU: x bf16 [prompts*G*G, >= 256] -> out bf16 [prompts*4*G*G, >= 64];
    w0 bf16 [32, 256, 8] (runs of eight K, run-major), prm f32 [3, 64]."""
    _dev(x, w0, prm, out)
    _req(x.dtype == out.dtype == w0.dtype == torch.bfloat16 and prm.dtype == torch.float32, "sam_upscale_ln_gelu: dtypes")
    _req(x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1 and x.shape[0] >= prompts * G * G and x.shape[1] >= 256
         and out.shape[0] >= prompts * 4 * G * G and out.shape[1] >= 64, "sam_upscale_ln_gelu: activations")
    _req(w0.is_contiguous() and w0.numel() == 256 * 256 and prm.is_contiguous() and prm.numel() == 192, "sam_upscale_ln_gelu: parameters")
    hip.check(hip.lib().marship_sam_upscale_ln_gelu(ptr(x), x.stride(0), ptr(w0), ptr(prm), eps, ptr(out), out.stride(0), prompts, G, int(x_tiled), stream()),
              "marship_sam_upscale_ln_gelu")


def sam_upscale_masks(u: torch.Tensor, w3: torch.Tensor, bias: torch.Tensor, hyper: torch.Tensor, masks: torch.Tensor, prompts: int, G2: int):
    """ConvTranspose2d(64, 32, 2, 2) + GELU + product with the prompt's four hypernetwork vectors: u bf16 [prompts*G2*G2, >= 64],
    w3 bf16 [128, 64], bias f32 [32], hyper f32 [prompts, 4, >= 32] -> masks f32 [prompts, 4, 2*G2, 2*G2]."""
    _dev(u, w3, bias, hyper, masks)
    _req(u.dtype == w3.dtype == torch.bfloat16 and bias.dtype == hyper.dtype == masks.dtype == torch.float32, "sam_upscale_masks: dtypes")
    _req(u.dim() == 2 and u.stride(1) == 1 and u.shape[0] >= prompts * G2 * G2 and u.shape[1] >= 64, "sam_upscale_masks: u")
    _req(w3.is_contiguous() and w3.numel() == 128 * 64 and bias.is_contiguous() and bias.numel() == 32, "sam_upscale_masks: parameters")
    _req(hyper.dim() == 3 and hyper.shape[0] >= prompts and hyper.shape[1] == 4 and hyper.shape[2] >= 32 and hyper.stride(2) == 1
         and hyper.stride(0) == 4 * hyper.stride(1), "sam_upscale_masks: hyper")
    _req(masks.is_contiguous() and masks.numel() == prompts * 16 * G2 * G2, "sam_upscale_masks: masks")
    hip.check(hip.lib().marship_sam_upscale_masks(ptr(u), u.stride(0), ptr(w3), ptr(bias), ptr(hyper), hyper.stride(1), ptr(masks), prompts, G2,
                                                  stream()), "marship_sam_upscale_masks")


def sam_i2t_kv_operands(k: torch.Tensor, v: torch.Tensor, prompts: int, Tk: int, kv_ops: torch.Tensor):
    """The prompts' projected tokens k, v bf16 [prompts*Tk, >= 128] as the operands sam_i2t_block reads: kv_ops bf16 [prompts, 8, 64, 2, 8]."""
    _dev(k, v, kv_ops)
    _req(k.dtype == v.dtype == kv_ops.dtype == torch.bfloat16, "sam_i2t_kv_operands: bf16")
    for t in (k, v):
        _req(t.dim() == 2 and t.stride(1) == 1 and t.shape[0] >= prompts * Tk and t.shape[1] >= 128, "sam_i2t_kv_operands: k / v")
    _req(kv_ops.is_contiguous() and kv_ops.numel() >= prompts * 8 * 64 * 16, "sam_i2t_kv_operands: kv_ops")
    hip.check(hip.lib().marship_sam_i2t_kv_operands(ptr(k), k.stride(0), ptr(v), v.stride(0), prompts, Tk, ptr(kv_ops), stream()),
              "marship_sam_i2t_kv_operands")


def sam_i2t_block(qin: torch.Tensor, wq: torch.Tensor, wo: torch.Tensor, qbias: torch.Tensor, prm: torch.Tensor, kv_ops: torch.Tensor, Tk: int,
                  resid: torch.Tensor, prompts: int, Ni: int, eps: float, y: Optional[torch.Tensor], c1: torch.Tensor, tiled: int = 0):
    """Image -> token attention block + norm4 + the bf16 copy of the result in one pass (see marship_sam_i2t_block).  qin with Ni
    rows (instead of prompts*Ni) is shared by every prompt, and so is resid with Ni rows; qbias f32 [Ni, 128] = key_pe Wq^T + bq.
    ``tiled`` bit 0 / 1 / 2: qin / resid / the outputs are in the kernel's own lane order instead of row-major (same sizes)."""
    _dev(qin, wq, wo, qbias, prm, kv_ops, resid, y, c1)
    rows = prompts * Ni
    _req(qin.dtype == wq.dtype == wo.dtype == kv_ops.dtype == c1.dtype == torch.bfloat16, "sam_i2t_block: bf16 operands")
    _req(prm.dtype == resid.dtype == qbias.dtype == torch.float32 and (y is None or y.dtype == torch.float32), "sam_i2t_block: f32 stream")
    _req(qin.dim() == 2 and qin.stride(1) == 1 and qin.shape[1] >= 256 and qin.shape[0] in (rows, Ni), "sam_i2t_block: qin")
    _req(resid.is_contiguous() and resid.dim() == 2 and resid.shape[1] == 256 and resid.shape[0] in (rows, Ni), "sam_i2t_block: resid")
    _req(qbias.is_contiguous() and qbias.numel() == Ni * 128 and prm.is_contiguous() and prm.numel() == 3 * 256, "sam_i2t_block: qbias / prm")
    _req(wq.is_contiguous() and wq.numel() == 128 * 256 and wo.is_contiguous() and wo.numel() == 256 * 128, "sam_i2t_block: weights")
    _req(kv_ops.is_contiguous() and kv_ops.numel() >= prompts * 8 * 64 * 16, "sam_i2t_block: kv_ops")
    for t in (c1,) + ((y,) if y is not None else ()):
        _req(t.is_contiguous() and t.numel() == rows * 256, "sam_i2t_block: outputs")
    hip.check(hip.lib().marship_sam_i2t_block(ptr(qin), qin.stride(0), int(qin.shape[0] == Ni and prompts > 1), ptr(wq), ptr(wo), ptr(qbias), ptr(prm),
                                              ptr(kv_ops), Tk, ptr(resid), int(resid.shape[0] == Ni and prompts > 1), prompts, Ni, eps, ptr(y), ptr(c1),
                                              tiled, stream()), "marship_sam_i2t_block")


def sam_t2i_q_operands(q: torch.Tensor, prompts: int, Tq: int, q_ops: torch.Tensor):
    """The prompts' projected queries q bf16 [prompts*Tq, >= 128] as the operands sam_t2i_block reads: q_ops bf16 [prompts, 8, 64, 8]."""
    _dev(q, q_ops)
    _req(q.dtype == q_ops.dtype == torch.bfloat16 and q.dim() == 2 and q.stride(1) == 1 and q.shape[0] >= prompts * Tq and q.shape[1] >= 128,
         "sam_t2i_q_operands: q")
    _req(q_ops.is_contiguous() and q_ops.numel() >= prompts * 8 * 64 * 8, "sam_t2i_q_operands: q_ops")
    hip.check(hip.lib().marship_sam_t2i_q_operands(ptr(q), q.stride(0), prompts, Tq, ptr(q_ops), stream()), "marship_sam_t2i_q_operands")


def sam_t2i_kpe_operands(kpe: torch.Tensor) -> torch.Tensor:
    """key_pe Wk^T (any float dtype, [Ni, 128], Ni a multiple of 32) in the operand order of sam_t2i_block: bf16 [Ni/32, 8, 64, 8]."""
    Ni = kpe.shape[0]
    _req(kpe.dim() == 2 and kpe.shape[1] == 128 and Ni % 32 == 0, "sam_t2i_kpe_operands: [Ni, 128], Ni a multiple of 32")
    lane = torch.arange(64, device=kpe.device)
    n, g = lane % 32, lane // 32
    e = torch.arange(8, device=kpe.device)
    ps = torch.arange(8, device=kpe.device)
    col = 16 * ps[:, None, None] + 8 * (e[None, None, :] // 4) + 4 * g[None, :, None] + e[None, None, :] % 4          # [8, 64, 8]
    row = 32 * torch.arange(Ni // 32, device=kpe.device)[:, None, None, None] + n[None, None, :, None]               # [tiles, 1, 64, 1]
    return kpe[row.expand(-1, 8, -1, 8), col[None].expand(Ni // 32, -1, -1, -1)].to(torch.bfloat16).contiguous()


def sam_t2i_block(x: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, kpe_ops: torch.Tensor, bv: torch.Tensor, q_ops: torch.Tensor, prompts: int,
                  Ni: int, Tq: int, out: torch.Tensor, x_tiled: bool = False):
    """Token -> image attention with the key / value projections of the stream made inside the kernel (see marship_sam_t2i_block).
    x with Ni rows (instead of prompts*Ni) is one stream shared by every prompt."""
    _dev(x, wk, wv, kpe_ops, bv, q_ops, out)
    _req(x.dtype == wk.dtype == wv.dtype == kpe_ops.dtype == q_ops.dtype == out.dtype == torch.bfloat16 and bv.dtype == torch.float32, "sam_t2i_block: dtypes")
    x_shared = prompts > 1 and x.shape[0] == Ni
    _req(x.dim() == 2 and x.stride(1) == 1 and (x_shared or x.shape[0] >= prompts * Ni) and x.shape[1] >= 256, "sam_t2i_block: x")
    _req(wk.is_contiguous() and wk.numel() == 128 * 256 and wv.is_contiguous() and wv.numel() == 128 * 256, "sam_t2i_block: weights")
    _req(kpe_ops.is_contiguous() and kpe_ops.numel() == Ni // 32 * 8 * 64 * 8 and bv.is_contiguous() and bv.numel() == 128, "sam_t2i_block: kpe_ops / bv")
    _req(q_ops.is_contiguous() and q_ops.numel() >= prompts * 8 * 64 * 8, "sam_t2i_block: q_ops")
    _req(out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= prompts * Tq and out.shape[1] >= 128, "sam_t2i_block: out")
    hip.check(hip.lib().marship_sam_t2i_block(ptr(x), x.stride(0), ptr(wk), ptr(wv), ptr(kpe_ops), ptr(bv), ptr(q_ops), prompts, Ni, Tq, ptr(out),
                                              out.stride(0), int(x_tiled), int(x_shared), stream()), "marship_sam_t2i_block")


def hyper_product(hyper: torch.Tensor, up: torch.Tensor, masks: torch.Tensor, B: int, P: int, C: int):
    """masks[b, m, p] = hyper[b, m, :C] . up[b*P + p, :C]; hyper f32 [B, 4, >= C], up f32 [B*P, >= C], masks f32 [B, 4, P]."""
    _dev(hyper, up, masks)
    _req(hyper.dtype == torch.float32 and up.dtype == torch.float32 and masks.dtype == torch.float32, "hyper_product: f32 operands")
    _req(hyper.dim() == 3 and hyper.shape[0] == B and hyper.shape[1] == 4 and hyper.stride(2) == 1 and hyper.stride(0) == 4 * hyper.stride(1), "hyper_product: hyper")
    _req(up.dim() == 2 and up.stride(1) == 1 and up.shape[0] >= B * P and masks.is_contiguous() and masks.numel() == B * 4 * P, "hyper_product: up / masks")
    hip.check(hip.lib().marship_hyper_product(ptr(hyper), hyper.stride(1), ptr(up), up.stride(0), ptr(masks), B, P, C, stream()), "marship_hyper_product")


def act_cast(x: torch.Tensor, out: torch.Tensor, act: int):
    _dev(x, out)
    _req(x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous() and out.numel() == x.numel(), "act_cast")
    hip.check(hip.lib().marship_act_cast(ptr(x), ptr(out), x.numel(), act, dt(out), stream()), "marship_act_cast")


def mask_stats(masks: torch.Tensor, thr: float, off: float, out: torch.Tensor):
    _dev(masks, out)
    _req(masks.dim() == 3 and masks.is_contiguous() and masks.dtype == torch.float32, "mask_stats: masks must be f32 [n,H,W]")
    _req(out.dtype == torch.int32 and out.is_contiguous() and out.numel() >= 6 * masks.shape[0], "mask_stats: out")
    n, H, W = masks.shape
    hip.check(hip.lib().marship_mask_stats(ptr(masks), n, H, W, float(thr), float(off), ptr(out), stream()), "marship_mask_stats")


def _axis_table(t) -> "hip.MarsAxisTable":
    lo, size, w = t
    _dev(lo, size, w)
    _req(lo.dtype == torch.int32 and size.dtype == torch.int32 and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
         and lo.numel() == w.shape[0] and size.numel() == w.shape[0], "axis table = (lo i32 [n], size i32 [n], weights f32 [n, kmax])")
    a = hip.MarsAxisTable()
    a.lo, a.size, a.w, a.kmax = lo.data_ptr(), size.data_ptr(), w.data_ptr(), w.shape[1]
    return a


def _low_planes(low: torch.Tensor):
    _req(low.dim() == 4 and low.dtype == torch.float32 and low.shape[2] == low.shape[3] and low.stride(3) == 1 and low.stride(2) == low.shape[3],
         "low-resolution logits must be f32 [B, nm, L, L] with contiguous planes")
    B, nm, L, _ = low.shape
    _req((B - 1) * low.stride(0) + (nm - 1) * low.stride(1) + L * L <= _avail(low), "low-resolution logits overrun their buffer")
    return B, nm, L


def resample_mask_stats(low: torch.Tensor, tx, ty, band: int, max_rows: int, thr: float, off: float, out: torch.Tensor,
                        bits: Optional[torch.Tensor] = None):
    """mask_stats of the image-size logits that (tx, ty) make from ``low`` [B, nm, L, L] (a view is fine), without storing them:
    out i32 [B*nm, 6].  (band, max_rows) from resample.band_plan on the host copy of ty.  ``bits`` f32 [B*nm, H, W]: also the
    0 / 1 masks of every plane, in the same pass."""
    B, nm, L = _low_planes(low)
    _dev(low, out, bits)
    _req(out.dtype == torch.int32 and out.is_contiguous() and out.numel() >= 6 * B * nm, "resample_mask_stats: out")
    _req(bits is None or (bits.dtype == torch.float32 and bits.is_contiguous() and bits.numel() >= B * nm * ty[0].numel() * tx[0].numel()),
         "resample_mask_stats: bits")
    ax, ay = _axis_table(tx), _axis_table(ty)
    hip.check(hip.lib().marship_resample_mask_stats(ptr(low), low.stride(0), low.stride(1), nm, B * nm, L, ty[0].numel(), tx[0].numel(),
                                                    ctypes.byref(ax), ctypes.byref(ay), band, max_rows, float(thr), float(off), ptr(out), ptr(bits),
                                                    stream()), "marship_resample_mask_stats")


def resample_threshold(low: torch.Tensor, sel: Optional[torch.Tensor], tx, ty, band: int, max_rows: int, thr: float, out: torch.Tensor):
    """out[i] f32 [H, W] = 0 / 1 of (image-size logits of plane sel[i] of ``low`` [B, nm, L, L] > thr); sel i32 on the device
    (None: every plane in order)."""
    B, nm, L = _low_planes(low)
    _dev(low, out, sel)
    H, W = ty[0].numel(), tx[0].numel()
    n = B * nm if sel is None else sel.numel()
    _req(sel is None or (sel.dtype == torch.int32 and sel.is_contiguous()), "resample_threshold: sel must be i32")
    _req(out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= n * H * W, "resample_threshold: out")
    ax, ay = _axis_table(tx), _axis_table(ty)
    for a in range(0, n, 65535):                                      # (grid limit a launch)
        k = min(65535, n - a)
        if sel is None:
            _req(n <= 65535, "resample_threshold: more than 65535 planes need a selection")
        hip.check(hip.lib().marship_resample_threshold(ptr(low), low.stride(0), low.stride(1), nm, (sel.data_ptr() + 4 * a) if sel is not None else None, k,
                                                       L, H, W, ctypes.byref(ax), ctypes.byref(ay), band, max_rows, float(thr),
                                                       out.data_ptr() + 4 * a * H * W, stream()), "marship_resample_threshold")


def threshold_masks(src: torch.Tensor, thr: float, dst: torch.Tensor):
    _dev(src, dst)
    _req(src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and dst.numel() >= src.numel(),
         "threshold_masks")
    hip.check(hip.lib().marship_threshold_masks(ptr(src), float(thr), ptr(dst), src.numel(), stream()), "marship_threshold_masks")


def box_nms_host(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS on host tensors; returns the kept indices (int64) by decreasing score."""
    b = boxes.detach().cpu().float().contiguous()
    s = scores.detach().cpu().float().contiguous()
    if b.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64)
    n = b.shape[0]
    keep = torch.zeros(max(n, 1), dtype=torch.int32)
    cnt = ctypes.c_int32(0)
    hip.check(hip.lib().marshost_box_nms(_vp(b.data_ptr()), _vp(s.data_ptr()), n, float(iou_threshold), _vp(keep.data_ptr()), ctypes.byref(cnt)),
              "marshost_box_nms")
    return keep[: cnt.value].long()


def remove_small_regions_host(mask, area_thresh: float, mode: str):
    """remove_small_regions (segment_anything/utils/amg.py:274-298) on one host [H, W] bool array:
    (new mask, changed)."""
    import numpy as np
    assert mode in ("holes", "islands")
    m = np.ascontiguousarray(np.asarray(mask) != 0).astype(np.uint8)
    _req(m.ndim == 2 and m.size > 0, "remove_small_regions: mask must be a non-empty [H, W] array")
    changed = ctypes.c_int32(0)
    hip.check(hip.lib().marshost_remove_small_regions(_vp(m.ctypes.data), m.shape[0], m.shape[1], float(area_thresh),
                                                      1 if mode == "holes" else 0, ctypes.byref(changed)),
              "marshost_remove_small_regions")
    return m.astype(bool), bool(changed.value)


def lap_host(weights: torch.Tensor, maximize: bool = False):
    """Rectangular linear assignment on a host f32 matrix: (row_ind, col_ind) int64, min(n, m) pairs by row."""
    w = weights.detach().cpu().float()
    if w.stride(1) != 1:
        w = w.contiguous()
    _req(w.dim() == 2 and w.shape[0] > 0 and w.shape[1] > 0, "lap_host: weights must be a non-empty matrix")
    n, m = w.shape
    k = min(n, m)
    rows, cols = torch.empty(k, dtype=torch.int64), torch.empty(k, dtype=torch.int64)
    rc = hip.lib().marshost_lap(_vp(w.data_ptr()), w.stride(0), n, m, int(bool(maximize)), _vp(rows.data_ptr()), _vp(cols.data_ptr()))
    if rc < 0:
        hip.check(rc, "marshost_lap")
    return rows, cols


def pack_masks(masks: torch.Tensor, words: torch.Tensor):
    """masks f32 [n, ...] -> words int32 [n, ceil(HW / 32)], bit i of word j = pixel 32 j + i > 0."""
    _dev(masks, words)
    n = masks.shape[0]
    HW = masks.numel() // n
    _req(masks.is_contiguous() and masks.dtype == torch.float32, "pack_masks: masks must be contiguous f32")
    _req(words.dtype == torch.int32 and words.is_contiguous() and words.numel() == n * ((HW + 31) // 32), "pack_masks: words")
    hip.check(hip.lib().marship_pack_masks(ptr(masks), n, HW, ptr(words), stream()), "marship_pack_masks")


def unpack_masks(words: torch.Tensor, masks: torch.Tensor):
    _dev(masks, words)
    n = masks.shape[0]
    HW = masks.numel() // n
    _req(masks.is_contiguous() and masks.dtype == torch.float32, "unpack_masks: masks must be contiguous f32")
    _req(words.dtype == torch.int32 and words.is_contiguous() and words.numel() == n * ((HW + 31) // 32), "unpack_masks: words")
    hip.check(hip.lib().marship_unpack_masks(ptr(words), n, HW, ptr(masks), stream()), "marship_unpack_masks")


def mask_embed(masks: torch.Tensor, G: int, params, eps: float, out: torch.Tensor):
    """masks f32 [B, 1, 4G, 4G] -> out f32 [B, G*G, C]; params = (w1, b1, g1, h1, w2, b2, g2, h2, w3, b3), all f32 contiguous."""
    _dev(masks, out, *params)
    B = masks.shape[0]
    w1, b1, g1, h1, w2, b2, g2, h2, w3, b3 = params
    m4, m, C = w1.shape[0], w2.shape[0], w3.shape[0]
    _req(masks.dtype == torch.float32 and masks.is_contiguous() and tuple(masks.shape[1:]) == (1, 4 * G, 4 * G), "mask_embed: masks must be f32 [B,1,4G,4G]")
    _req(all(p.dtype == torch.float32 and p.is_contiguous() for p in params), "mask_embed: parameters must be contiguous f32")
    _req(w1.numel() == m4 * 4 and b1.numel() == m4 and g1.numel() == m4 and h1.numel() == m4 and w2.numel() == m * m4 * 4
         and b2.numel() == m and g2.numel() == m and h2.numel() == m and w3.numel() == C * m and b3.numel() == C, "mask_embed: parameter sizes")
    _req(out.dtype == torch.float32 and out.is_contiguous() and out.numel() == B * G * G * C, "mask_embed: out must be f32 [B, G*G, C]")
    hip.check(hip.lib().marship_mask_embed(ptr(masks), B, G, m4, m, C, *[ptr(p) for p in params], float(eps), ptr(out), stream()),
              "marship_mask_embed")
