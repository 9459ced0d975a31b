"""Kernel-level parity on the GPU: every C-ABI entry point against a plain torch fp32 (or fp64)
statement of the same arithmetic, at the odd shapes the pipeline produces."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (130, 70, 192), (1, 5, 64), (257, 384, 128), (81, 81, 128), (1374, 1024, 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_plain(pkg, cuda, M, N, K, dtype):
    a = _rand((M, K), 1, cuda).to(dtype)
    b = _rand((N, K), 2, cuda).to(dtype)
    out = torch.full((M, N), float("nan"), device=cuda)
    pkg.ops.linear(a, b, out)
    # bf16 operands are exact in the f32 products, so both types only differ from the f64
    # reference by the f32 accumulation error
    ref = a.double() @ b.double().t()
    assert float((out.double() - ref).abs().max()) <= 1e-3


def test_gemm_asymmetric_identity(pkg, cuda):
    """A = I with an asymmetric B catches a swapped row/column map in the C write."""
    K = 128
    a = torch.eye(K, device=cuda)
    b = torch.arange(96 * K, device=cuda, dtype=torch.float32).reshape(96, K) / 7.0
    out = torch.zeros(K, 96, device=cuda)
    pkg.ops.linear(a, b, out)
    assert torch.equal(out, b.t().contiguous())


def _act(z, act):
    return F.gelu(z) if act == 1 else (z * torch.sigmoid(1.702 * z) if act == 2 else (F.relu(z) if act == 3 else z))


@pytest.mark.parametrize("variant", [5, 6, 7, 8, 11, 12])
@pytest.mark.parametrize("act", [1, 3])
def test_gemm_fast_epilogue_activations(pkg, cuda, variant, act):
    """erf-GELU and ReLU on the register-direct epilogue of the ring kernels (aligned rows, no residual)."""
    lib = pkg.hip.lib()
    M, N, K = 300, 256, 128
    a, b = _rand((M, K), 31, cuda), _rand((N, K), 32, cuda) * 0.2
    bias = _rand((N,), 33, cuda)
    old = lib.marship_set_gemm_variant(variant)
    try:
        for dt, tol in ((torch.float32, 2e-4), (torch.bfloat16, 0.05)):
            out = torch.full((M, N), float("nan"), device=cuda, dtype=dt)
            pkg.ops.linear(a.to(dt), b.to(dt), out, bias=bias, act=act)
            ref = _act(a.to(dt).double() @ b.to(dt).double().t() + bias.double(), act)
            assert float((out.double() - ref).abs().max()) < tol
    finally:
        lib.marship_set_gemm_variant(old)


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogue(pkg, cuda, act):
    M, N, K = 77, 200, 128
    a, b = _rand((M, K), 3, cuda), _rand((N, K), 4, cuda) * 0.2
    bias, gamma = _rand((N,), 5, cuda), _rand((N,), 6, cuda)
    res = _rand((M, N), 7, cuda)
    out = res.clone()
    pkg.ops.linear(a, b, out, bias=bias, gamma=gamma, residual=out, act=act, alpha=0.5)
    z = _act(0.5 * (a @ b.t()) + bias, act)
    ref = res + gamma * z
    assert float((out - ref).abs().max()) < 2e-4
    outb = torch.zeros(M, N, device=cuda, dtype=torch.bfloat16)
    pkg.ops.linear(a, b, outb, bias=bias, act=act)
    z = _act((a @ b.t()) + bias, act)
    assert float((outb.float() - z).abs().max()) < 0.05


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 200, 192), (513, 257, 64), (256, 512, 1024), (70, 1100, 320)])
def test_gemm_every_variant(pkg, cuda, variant, dtype, M, N, K):
    """Every kernel behind marship_gemm on ragged tiles, with the full epilogue, both output types."""
    lib = pkg.hip.lib()
    a = _rand((M, K), 11, cuda).to(dtype)
    b = (_rand((N, K), 12, cuda) * 0.2).to(dtype)
    bias, gamma = _rand((N,), 13, cuda), _rand((N,), 14, cuda)
    res = _rand((M, N), 15, cuda)
    z = 0.5 * (a.double() @ b.double().t()) + bias.double()
    z = z * torch.sigmoid(1.702 * z)
    old = lib.marship_set_gemm_variant(variant)
    try:
        out = res.clone()
        pkg.ops.linear(a, b, out, bias=bias, gamma=gamma, residual=out, act=2, alpha=0.5)
        # bf16 inputs take the fast exp / reciprocal in the QuickGELU of the ring kernels
        tol = 2e-4 if dtype == torch.float32 else 2e-3
        assert float((out.double() - (res.double() + gamma.double() * z)).abs().max()) < tol
        outb = torch.full((M, N), float("nan"), device=cuda, dtype=torch.bfloat16)
        pkg.ops.linear(a, b, outb, bias=bias, act=2, alpha=0.5)
        assert float((outb.double() - z).abs().max()) < 0.06
        # output rows wider than N keep their padding untouched (ldc > N, unaligned N)
        wide = torch.full((M, N + 3), -7.0, device=cuda)
        pkg.ops.linear(a, b, wide[:, :N])
        assert float((wide[:, :N].double() - a.double() @ b.double().t()).abs().max()) < 2e-3
        assert float((wide[:, N:] + 7.0).abs().max()) == 0.0
    finally:
        lib.marship_set_gemm_variant(old)


@pytest.mark.parametrize("slots", [1, 3, 8, 24])
def test_gemm_persistent_workgroups(pkg, cuda, slots):
    """Variant 12 - workgroups that walk several tiles and request the next tile's first steps before they store
    the one in hand - equals variant 8 (same tile, same summation order, one workgroup per tile) bit for bit:
    register epilogues (plain, QuickGELU, f32 residual, folded-LayerNorm forms), the staged per-element epilogue
    of a ragged product, and tiles with rows beyond M."""
    lib, ops = pkg.hip.lib(), pkg.ops
    bf = torch.bfloat16

    def both(fn):
        outs = []
        for v in (8, 12):
            old = lib.marship_set_gemm_variant(v)
            old_slots = lib.marship_set_gemm_slots(slots)
            try:
                outs.append(fn())
            finally:
                lib.marship_set_gemm_variant(old)
                lib.marship_set_gemm_slots(old_slots)
        torch.cuda.synchronize()
        for x, y in zip(outs[0], outs[1]):
            assert torch.equal(x, y)
        return outs[0]

    # aligned product, 5 x 6 tiles with a ragged last row tile
    M, N, K = 1100, 1536, 512
    a = _rand((M, K), 51, cuda).to(bf)
    b = (_rand((N, K), 52, cuda) * 0.2).to(bf)
    bias, gamma, res = _rand((N,), 53, cuda), _rand((N,), 54, cuda), _rand((M, N), 55, cuda)

    def aligned():
        o1 = torch.full((M, N), float("nan"), device=cuda, dtype=bf)
        ops.linear(a, b, o1, bias=bias)
        o2 = torch.full((M, N), float("nan"), device=cuda, dtype=bf)
        ops.linear(a, b, o2, bias=bias, act=2, gamma=gamma)
        o3 = res.clone()
        ops.linear(a, b, o3, bias=bias, gamma=gamma, residual=o3)
        o4 = torch.full((M, N), float("nan"), device=cuda)
        ops.linear(a, b, o4)
        return o1, o2, o3, o4

    o1, o2, o3, o4 = both(aligned)
    ref = a.double() @ b.double().t()
    assert float((o4.double() - ref).abs().max()) < 2e-3
    assert float((o3.double() - (res.double() + gamma.double() * (ref + bias.double()))).abs().max()) < 2e-3
    y1 = ref + bias.double()
    assert float(((o1.double() - y1).abs() - 2.0 ** -8 * y1.abs()).max()) < 1e-2       # one bf16 rounding

    # ragged product: N edge tiles and unaligned rows go through the staged epilogue, with an activation and a residual
    M2, N2, K2 = 700, 1100, 320
    a2 = _rand((M2, K2), 56, cuda).to(bf)
    b2 = (_rand((N2, K2), 57, cuda) * 0.2).to(bf)
    bias2, res2 = _rand((N2,), 58, cuda), _rand((M2, N2), 59, cuda)

    def ragged():
        o1 = res2.clone()
        ops.linear(a2, b2, o1, bias=bias2, residual=o1, act=2, alpha=0.5)
        wide = torch.full((M2, N2 + 3), -7.0, device=cuda)
        ops.linear(a2, b2, wide[:, :N2])
        return o1, wide

    r1, wide = both(ragged)
    z = 0.5 * (a2.double() @ b2.double().t()) + bias2.double()
    assert float((r1.double() - (res2.double() + z * torch.sigmoid(1.702 * z))).abs().max()) < 2e-3
    assert float((wide[:, N2:] + 7.0).abs().max()) == 0.0

    # folded-LayerNorm forms: the stream product (planes in, planes and row sums out) and its consumer
    M3, D, K3 = 1000, 512, 1024
    x0 = _rand((M3, D), 60, cuda) * 0.7 + 0.3
    h = _rand((M3, K3), 61, cuda).to(bf)
    w2 = (_rand((D, K3), 62, cuda) * 0.05).to(bf)
    b2_, ls = _rand((D,), 63, cuda), _rand((D,), 64, cuda) * 0.5 + 1.0
    wf = (_rand((1536, D), 65, cuda) * 0.05).to(bf)
    colsum = wf.float().sum(1).contiguous()
    bf_ = _rand((1536,), 66, cuda) * 0.1

    def folded():
        hi = torch.empty(M3, D, device=cuda, dtype=bf)
        lo = torch.empty(M3, D, device=cuda, dtype=bf)
        stat = torch.empty(M3, 2, device=cuda)
        ops.row_stats(x0, 1e-5, hi, lo, stat)
        part = torch.full((M3, D // 64, 2), float("nan"), device=cuda)
        dummy = torch.zeros(M3, D, device=cuda)
        ops.linear(h, w2, dummy, bias=b2_, gamma=ls, ln_out=(hi, lo, part))
        ops.ln_finish(part, M3, D, 1e-5, stat)
        out = torch.full((M3, 1536), float("nan"), device=cuda, dtype=bf)
        ops.linear(hi, wf, out, bias=bf_, act=2, ln_in=(stat, colsum))
        return hi, lo, part, stat, out

    both(folded)


def test_gemm_split_launch(pkg, cuda):
    """A tall product the launch policy cuts into two launches (2.27 rounds of 256 x 256 tiles) equals the same
    product kept in one launch, bit for bit, and the split point is a whole number of row tiles."""
    lib = pkg.hip.lib()
    M, N, K = 36928, 1024, 2048
    a = _rand((M, K), 41, cuda).to(torch.bfloat16)
    b = (_rand((N, K), 42, cuda) * 0.2).to(torch.bfloat16)
    bias = _rand((N,), 43, cuda)
    x0 = _rand((M, N), 44, cuda)
    g = pkg.hip.MarsGemm()
    g.M, g.N, g.K, g.nb0, g.nb1, g.in_dtype, g.out_dtype, g.act = M, N, K, 1, 1, 1, 0, 0
    import ctypes
    rows = lib.marship_gemm_split_rows(ctypes.byref(g))
    assert 0 < rows < M and rows % 256 == 0
    x1 = x0.clone()
    pkg.ops.linear(a, b, x1, bias=bias, residual=x1)
    old = lib.marship_set_gemm_split(0)
    try:
        assert lib.marship_gemm_split_rows(ctypes.byref(g)) == M
        x2 = x0.clone()
        pkg.ops.linear(a, b, x2, bias=bias, residual=x2)
    finally:
        lib.marship_set_gemm_split(old)
    ref = x0[rows - 2:rows + 2] + a[rows - 2:rows + 2].float() @ b.float().t() + bias
    assert float((x1[rows - 2:rows + 2] - ref).abs().max()) < 5e-3
    assert float((x1 - x2).abs().max()) < 1e-3      # different tile shapes sum K in different orders
    assert torch.equal(x1[:rows], x2[:rows])


@pytest.mark.parametrize("D", [384, 512, 768, 1024, 1280, 1536, 2048])
def test_layernorm_does_not_depend_on_the_number_of_rows(pkg, cuda, D):
    """The LayerNorm form (a workgroup or a wave per row) is chosen by width and output type alone: a row gives the same
    bits in a small launch and inside a large one, and matches torch."""
    ops = pkg.ops
    x = _rand((9000, D), 71, cuda) * 3 + 0.5
    w, b = _rand((D,), 72, cuda), _rand((D,), 73, cuda)
    for odt in (torch.float32, torch.bfloat16):
        big = torch.empty(9000, D, device=cuda, dtype=odt)
        ops.layernorm(x, w, b, big, 1e-6)
        small = torch.empty(3000, D, device=cuda, dtype=odt)
        ops.layernorm(x[:3000], w, b, small, 1e-6)
        assert torch.equal(small, big[:3000])
        ref = torch.nn.functional.layer_norm(x[:3000], (D,), w, b, 1e-6)
        assert float((small.float() - ref).abs().max()) < (1e-5 if odt == torch.float32 else 0.05)


def test_gemm_column_split(pkg, cuda):
    """A product of 1.25 rounds of 256 x 256 tiles (16 x 20: the fc1 of a ViT-H layer on 4096 tokens) goes out as one round
    of whole tile columns and the remaining columns as a launch of their own: equal to the product kept in one launch up to
    the summation order of another tile shape, and to f32 on the same operands, with bias, GELU and bf16 output or residual."""
    lib = pkg.hip.lib()
    M, N, K = 4096, 5120, 1024
    a = _rand((M, K), 51, cuda).to(torch.bfloat16)
    b = (_rand((N, K), 52, cuda) * 0.05).to(torch.bfloat16)
    bias = _rand((N,), 53, cuda)
    ref = a.float() @ b.float().t() + bias
    for act, odt in ((1, torch.bfloat16), (0, torch.float32)):
        outs = []
        for split in (1, 0):
            old = lib.marship_set_gemm_split(split)
            try:
                out = torch.full((M, N), float("nan"), device=cuda, dtype=odt) if act else _rand((M, N), 54, cuda)
                pkg.ops.linear(a, b, out, bias=bias, act=act, residual=None if act else out)
            finally:
                lib.marship_set_gemm_split(old)
            outs.append(out.float())
        want = torch.nn.functional.gelu(ref) if act else ref + _rand((M, N), 54, cuda)
        tol = (5e-3 if act else 1e-3) * want.abs().clamp(min=1.0)        # (bf16 output: half a unit in the last place)
        assert bool(((outs[0] - want).abs() <= tol).all()) and bool(((outs[1] - want).abs() <= tol).all())
        assert bool(((outs[0] - outs[1]).abs() <= 2 * tol).all())


@pytest.mark.parametrize("M,D,N2,K2", [(1000, 128, 384, 512), (4616, 1024, 4096, 1024), (36928, 1024, 3072, 4096)])
def test_gemm_folded_layernorm(pkg, cuda, M, D, N2, K2):
    """A LayerNorm between two products, folded into both: the product that writes the f32 stream leaves its
    rounded copy and the row sums, the product that would have read the normalised rows applies mean and rstd
    to its accumulators.  (The last shape runs the stream product as two launches.)"""
    ops = pkg.ops
    eps = 1e-5
    x0 = (_rand((M, D), 1, cuda) * 0.7 + 0.3)                      # stream before the product, mean != 0
    h = _rand((M, K2), 2, cuda).to(torch.bfloat16)
    w2 = (_rand((D, K2), 3, cuda) * 0.05).to(torch.bfloat16)
    b2 = _rand((D,), 4, cuda)
    ls = _rand((D,), 5, cuda) * 0.5 + 1.0
    nan16 = lambda: torch.full((M, D), float("nan"), device=cuda, dtype=torch.bfloat16)        # noqa: E731
    hi, lo, stat0 = nan16(), nan16(), torch.full((M, 2), float("nan"), device=cuda)
    # the stream as two bf16 planes, with the statistics of its rows
    ops.row_stats(x0, eps, hi, lo, stat0)
    assert torch.equal(hi, x0.to(torch.bfloat16)) and torch.equal(lo, (x0 - hi.float()).to(torch.bfloat16))
    xr = hi.float() + lo.float()                                   # what the planes hold: x0 to 16 mantissa bits
    assert float(((xr - x0).abs() - 2.0 ** -16 * x0.abs()).max()) <= 0
    xd0 = x0.double()
    assert float((stat0[:, 0].double() - xd0.mean(1)).abs().max()) < 1e-5
    assert float((stat0[:, 1].double() * torch.sqrt(xd0.var(1, unbiased=False) + eps) - 1).abs().max()) < 1e-4
    # a product that adds to the stream: same f32 values as the plain product on the reconstructed stream
    plain = xr.clone()
    ops.linear(h, w2, plain, bias=b2, gamma=ls, residual=plain)
    last = torch.full((M, D), float("nan"), device=cuda)
    ops.linear(h, w2, last, bias=b2, gamma=ls, ln_out=(hi, lo, None))          # last of a run: f32 out
    assert torch.equal(last, plain)
    part = torch.full((M, D // 64, 2), float("nan"), device=cuda)
    stat = torch.full((M, 2), float("nan"), device=cuda)
    untouched = torch.full((M, D), 7.0, device=cuda)
    ops.linear(h, w2, untouched, bias=b2, gamma=ls, ln_out=(hi, lo, part))      # inside a run: back into the planes
    assert bool((untouched == 7.0).all())
    xb = hi
    assert torch.equal(hi, plain.to(torch.bfloat16)) and torch.equal(lo, (plain - hi.float()).to(torch.bfloat16))
    ops.ln_finish(part, M, D, eps, stat)
    x = plain
    xd = x.double()
    assert float((part[:, :, 0].double().sum(1) - xd.sum(1)).abs().max()) < 1e-3 * D ** 0.5
    mean, var = xd.mean(1), xd.var(1, unbiased=False)
    assert float((stat[:, 0].double() - mean).abs().max()) < 1e-5
    assert float((stat[:, 1].double() * torch.sqrt(var + eps) - 1).abs().max()) < 1e-4
    # consumer: N2 outputs from the normalised rows, QuickGELU or none, with a per-column factor
    lnw, lnb = _rand((D,), 6, cuda) * 0.2 + 1.0, _rand((D,), 7, cuda) * 0.1
    w1, b1 = _rand((N2, D), 8, cuda) * 0.05, _rand((N2,), 9, cuda) * 0.1
    wf, bf = pkg.models._fold_norm(w1, b1, lnw, lnb)
    wf = wf.to(torch.bfloat16)
    colsum = wf.float().sum(1).contiguous()
    gam = _rand((N2,), 10, cuda) * 0.3 + 1.0
    for act in (0, 2):
        out = torch.full((M, N2), float("nan"), device=cuda, dtype=torch.bfloat16)
        ops.linear(xb, wf, out, bias=bf, act=act, gamma=gam, ln_in=(stat, colsum))
        rows = slice(0, M) if M <= 5000 else torch.cat([torch.arange(0, 700), torch.arange(32700, 33400), torch.arange(M - 300, M)]).to(cuda)
        # the arithmetic of the kernel, in f64 on the same operands
        t = xb[rows].double() @ wf.double().t()
        y = stat[rows, 1:2].double() * (t - stat[rows, 0:1].double() * colsum.double()[None]) + bf.double()[None]
        if act == 2:
            y = y * torch.sigmoid(1.702 * y)
        y = y * gam.double()[None]
        err = (out[rows].double() - y).abs()
        assert float((err - 2.0 ** -8 * y.abs()).max()) < 2e-3, float(err.max())
        # ...and what it stands for: the product of the normalised f32 rows
        ln = torch.nn.functional.layer_norm(x[rows], (D,), lnw, lnb, eps)
        z = ln.double() @ w1.double().t() + b1.double()[None]
        if act == 2:
            z = z * torch.sigmoid(1.702 * z)
        z = z * gam.double()[None]
        stand_alone = torch.zeros(ln.shape[0], N2, device=cuda, dtype=torch.bfloat16)
        ops.linear(ln.to(torch.bfloat16), w1.to(torch.bfloat16), stand_alone, bias=b1, act=act, gamma=gam)
        e_fold = float((out[rows].double() - z).abs().max())
        e_alone = float((stand_alone.double() - z).abs().max())
        assert e_fold < 1.5 * e_alone + 1e-3, (e_fold, e_alone)


@pytest.mark.parametrize("ratio", [0.4, 10.0, 30.0])
def test_gemm_folded_layernorm_rows_with_large_means(pkg, cuda, ratio):
    """The folded form rounds x to bf16, the stand-alone pass x - mean: on rows with a common offset of `ratio`
    standard deviations (and a few outlier channels on top) the folded product is further from the f64 result by
    about that factor, never by more than 1.5 * (1 + ratio) times the stand-alone error; the statistics kernels
    report the rows beyond a limit."""
    ops = pkg.ops
    M, D, N2, eps = 1024, 1024, 512, 1e-5
    x = _rand((M, D), 11, cuda)
    x[:, 5] *= 20.0                                                # outlier channels, as residual streams have
    x[:, 700] *= -12.0
    x = x / x.std(1, keepdim=True)
    x = x + ratio * torch.sign(_rand((M, 1), 12, cuda))            # rows at +-ratio standard deviations
    hi, lo = torch.zeros(M, D, device=cuda, dtype=torch.bfloat16), torch.zeros(M, D, device=cuda, dtype=torch.bfloat16)
    stat, watch = torch.zeros(M, 2, device=cuda), torch.zeros(1, device=cuda)
    ops.row_stats(x, eps, hi, lo, stat, watch, 8.0)
    got = float(watch[0])
    true_ratio = float((x.double().mean(1).abs() / x.double().std(1, unbiased=False)).max())
    if ratio < 8.0:
        assert got == 0.0
    else:
        assert abs(got - true_ratio) < 1e-3 * true_ratio
    # the same report from the sums a producing epilogue leaves
    part = torch.stack([x.view(M, D // 64, 64).sum(2), (x * x).view(M, D // 64, 64).sum(2)], dim=2).contiguous()
    stat2, watch2 = torch.zeros(M, 2, device=cuda), torch.zeros(1, device=cuda)
    ops.ln_finish(part, M, D, eps, stat2, watch2, 8.0)
    assert float((stat2 - stat).abs().max()) < 1e-3
    assert (float(watch2[0]) == 0.0) if ratio < 8.0 else abs(float(watch2[0]) - true_ratio) < 1e-2 * true_ratio
    lnw, lnb = _rand((D,), 6, cuda) * 0.2 + 1.0, _rand((D,), 7, cuda) * 0.1
    w1, b1 = _rand((N2, D), 8, cuda) * 0.05, _rand((N2,), 9, cuda) * 0.1
    wf, bf = pkg.models._fold_norm(w1, b1, lnw, lnb)
    wf = wf.to(torch.bfloat16)
    colsum = wf.float().sum(1).contiguous()
    out = torch.zeros(M, N2, device=cuda, dtype=torch.bfloat16)
    ops.linear(hi, wf, out, bias=bf, ln_in=(stat, colsum))
    ln = torch.nn.functional.layer_norm(x, (D,), lnw, lnb, eps)
    z = ln.double() @ w1.double().t() + b1.double()[None]
    alone = torch.zeros(M, N2, device=cuda, dtype=torch.bfloat16)
    ops.linear(ln.to(torch.bfloat16), w1.to(torch.bfloat16), alone, bias=b1)
    e_fold = float((out.double() - z).abs().mean())
    e_alone = float((alone.double() - z).abs().mean())
    assert e_fold < 1.5 * (1.0 + ratio) * e_alone, (ratio, e_fold, e_alone)
    if ratio >= 10.0:
        assert e_fold > 2.0 * e_alone, (ratio, e_fold, e_alone)   # the loss is real: hence the report


def test_embed_tokens_matches_torch(pkg, cuda):
    ops = pkg.ops
    g = torch.Generator().manual_seed(3)
    table, pos = torch.randn(300, 96, generator=g).to(cuda), torch.randn(11, 96, generator=g).to(cuda)
    ids = torch.randint(0, 300, (5, 11), generator=g).int().to(cuda)
    X = torch.full((5, 11, 96), float("nan"), device=cuda)
    ops.embed_tokens(table, ids, pos, X)
    assert torch.equal(X, table[ids.long()] + pos[None])
    with pytest.raises(Exception):
        ops.embed_tokens(table, ids.long(), pos, X)


def test_gemm_folded_layernorm_is_refused_where_no_kernel_takes_it(pkg, cuda):
    ops = pkg.ops
    M, D = 128, 128
    x = _rand((M, D), 1, cuda)
    h = _rand((M, 64), 2, cuda).to(torch.bfloat16)
    w = _rand((D, 64), 3, cuda).to(torch.bfloat16)
    xb, lo = torch.zeros(M, D, device=cuda, dtype=torch.bfloat16), torch.zeros(M, D, device=cuda, dtype=torch.bfloat16)
    part, stat, colsum = torch.zeros(M, D // 64, 2, device=cuda), torch.zeros(M, 2, device=cuda), torch.zeros(D, device=cuda)
    with pytest.raises(RuntimeError):                      # the residual of a stream product comes from the planes
        ops.linear(h, w, x, residual=x, ln_out=(xb, lo, part))
    with pytest.raises(RuntimeError):                      # ...and it has no activation
        ops.linear(h, w, x, act=2, ln_out=(xb, lo, part))
    with pytest.raises(RuntimeError):                      # erf GELU has no folded form
        ops.linear(h, w, xb, act=1, ln_in=(stat, colsum))
    with pytest.raises(RuntimeError):                      # f32 operands have no folded form
        ops.linear(h.float(), w.float(), xb, ln_in=(stat, colsum))
    w96 = _rand((96, 64), 3, cuda).to(torch.bfloat16)
    with pytest.raises((RuntimeError, ValueError)):        # N must be a multiple of 64
        ops.linear(h, w96, torch.zeros(M, 96, device=cuda, dtype=torch.bfloat16), ln_in=(stat, torch.zeros(96, device=cuda)))


def test_gemm_large_m(pkg, cuda):
    """Default launch policy on a tall product (the shape of the AlphaCLIP batch): in-place residual,
    ragged last row tile, both output types."""
    M, N, K = 36928, 1024, 128
    a = _rand((M, K), 21, cuda).to(torch.bfloat16)
    b = (_rand((N, K), 22, cuda) * 0.2).to(torch.bfloat16)
    bias = _rand((N,), 23, cuda)
    x = _rand((M, N), 24, cuda)
    ref = x + (a.float() @ b.float().t() + bias)
    pkg.ops.linear(a, b, x, bias=bias, residual=x)
    assert float((x - ref).abs().max()) < 2e-3
    out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.bfloat16)
    pkg.ops.linear(a, b, out, bias=bias)
    assert float((out.float() - (ref - _rand((M, N), 24, cuda))).abs().max()) < 0.06


def test_gemm_batched_strided(pkg, cuda):
    """The attention-shaped call: per (image, head) views into a packed qkv buffer."""
    B, H, T, D = 2, 3, 70, 192
    Tp = 128
    qkv = _rand((B * T, 3 * D), 8, cuda)
    S = torch.zeros(B * H, T, Tp, device=cuda)
    pkg.ops.gemm(qkv, qkv, S, T, T, 64, lda=3 * D, ldb=3 * D, ldc=Tp, a_off=0, b_off=D, nb0=B, nb1=H,
                 sA=(T * 3 * D, 64), sB=(T * 3 * D, 64), sC=(H * T * Tp, T * Tp), alpha=0.125)
    q = qkv.view(B, T, 3, H, 64)
    ref = torch.einsum("bthd,bshd->bhts", q[:, :, 0], q[:, :, 1]) * 0.125
    assert float((S.view(B, H, T, Tp)[..., :T] - ref).abs().max()) < 1e-4
    assert float(S.view(B, H, T, Tp)[..., T:].abs().max()) == 0.0


def test_gemm_rejects_bad_shapes(pkg, cuda):
    a, b, c = torch.zeros(4, 40, device=cuda), torch.zeros(4, 40, device=cuda), torch.zeros(4, 4, device=cuda)
    with pytest.raises(ValueError):
        pkg.ops.linear(a, b, c)                      # K not a 128-byte multiple
    with pytest.raises(ValueError):
        pkg.ops.gemm(torch.zeros(4, 64, device=cuda), torch.zeros(4, 64, device=cuda), c, 8, 4, 64, lda=64, ldb=64, ldc=4)


# ------------------------------------------------------------------ ViT glue
@pytest.mark.parametrize("D", [128, 384, 1024, 1536])
def test_layernorm(pkg, cuda, D):
    x = _rand((37, D), 1, cuda, 3.0) + 1.5
    w, b = _rand((D,), 2, cuda), _rand((D,), 3, cuda)
    y = torch.empty_like(x)
    pkg.ops.layernorm(x, w, b, y, 1e-6)
    assert float((y - F.layer_norm(x, (D,), w, b, 1e-6)).abs().max()) < 2e-5
    yb = torch.empty(37, D, device=cuda, dtype=torch.bfloat16)
    pkg.ops.layernorm(x, w, b, yb, 1e-5)
    assert float((yb.float() - F.layer_norm(x, (D,), w, b, 1e-5)).abs().max()) < 0.05
    # strided rows (class-token gather) and in-place
    big = _rand((5, 7, D), 4, cuda)
    rows = big[:, 0, :]
    out = torch.empty(5, D, device=cuda)
    pkg.ops.layernorm(rows, w, b, out, 1e-5)
    assert float((out - F.layer_norm(rows, (D,), w, b, 1e-5)).abs().max()) < 2e-5
    ref = F.layer_norm(x, (D,), w, b, 1e-5)
    pkg.ops.layernorm(x, w, b, x, 1e-5)
    assert float((x - ref).abs().max()) < 2e-5


def test_im2col_matches_conv(pkg, cuda):
    B, p, G, D = 2, 14, 5, 32
    img = _rand((B, 3, p * G, p * G), 1, cuda)
    wgt = _rand((D, 3, p, p), 2, cuda, 0.05)
    mean = torch.tensor([0.485, 0.456, 0.406], device=cuda)
    std = torch.tensor([0.229, 0.224, 0.225], device=cuda)
    Kp = 640
    cols = torch.zeros(B * G * G, Kp, device=cuda)
    pkg.ops.im2col(img, p, cols, mean=mean, std=std)
    wp = torch.zeros(D, Kp, device=cuda)
    wp[:, : 3 * p * p] = wgt.reshape(D, -1)
    out = torch.empty(B * G * G, D, device=cuda)
    pkg.ops.linear(cols, wp, out)
    ref = F.conv2d((img - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1), wgt, stride=p).flatten(2).transpose(1, 2).reshape(-1, D)
    assert float((out - ref).abs().max()) < 1e-4
    # 8-bit input with the 1/255 prescale
    u8 = (torch.rand(1, 3, 28, 28, device=cuda) * 255).to(torch.uint8)
    cols8 = torch.zeros(4, Kp, device=cuda)
    pkg.ops.im2col(u8, p, cols8, mean=mean, std=std, prescale=1.0 / 255.0)
    ref8 = F.unfold((u8.float() / 255.0 - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1), p, stride=p)[0].t()
    assert float((cols8[:, : 3 * p * p] - ref8).abs().max()) < 1e-6


def test_softmax_transpose_attnmean(pkg, cuda):
    nb, T, ld = 5, 70, 128
    S = _rand((nb, T, ld), 1, cuda, 3.0)
    P = torch.full((nb, T, ld), float("nan"), device=cuda)
    pkg.ops.softmax_rows(S, P, nb, T, ld, False)
    assert float((P[..., :T] - S[..., :T].softmax(-1)).abs().max()) < 1e-6 and float(P[..., T:].abs().max()) == 0
    pkg.ops.softmax_rows(S, P, nb, T, ld, True)
    mask = torch.full((T, T), float("-inf"), device=cuda).triu_(1)
    assert float((P[..., :T] - (S[..., :T] + mask).softmax(-1)).abs().max()) < 1e-6
    Pb = torch.zeros(nb, T, ld, device=cuda, dtype=torch.bfloat16)
    pkg.ops.softmax_rows(S, Pb, nb, T, ld, False)
    assert float((Pb[..., :T].float() - S[..., :T].softmax(-1)).abs().max()) < 5e-3
    # transpose, both element sizes
    for dt in (torch.float32, torch.bfloat16):
        x = _rand((3, 45, 70), 2, cuda).to(dt)
        y = torch.zeros(3, 70, 64, device=cuda, dtype=dt)
        pkg.ops.transpose_raw(x, 0, 70, (45 * 70, 0), y, 0, 64, (70 * 64, 0), 45, 70, 3, 1)
        assert torch.equal(y[:, :, :45], x.transpose(1, 2)) and float(y[:, :, 45:].abs().max()) == 0
    # head/layer mean accumulate
    H, off, Pn = 4, 5, 60
    probs = S[:H, :, :T].softmax(-1).contiguous()
    acc = torch.zeros(Pn, 64, device=cuda)
    pkg.ops.attn_mean_accum(probs, 0, T, T * T, H, off, Pn, 0.25, acc)
    pkg.ops.attn_mean_accum(probs, 0, T, T * T, H, off, Pn, 0.25, acc)
    assert float((acc[:, :Pn] - 2 * probs[:, off:off + Pn, off:off + Pn].mean(0)).abs().max()) < 1e-6


def test_swiglu_cast_l2norm(pkg, cuda):
    x = _rand((9, 2 * 344), 1, cuda)
    out = torch.zeros(9, 384, device=cuda)
    pkg.ops.swiglu(x, out)
    ref = F.silu(x[:, :344]) * x[:, 344:]
    assert float((out[:, :344] - ref).abs().max()) < 1e-6 and float(out[:, 344:].abs().max()) == 0
    xb = torch.empty(9, 688, device=cuda, dtype=torch.bfloat16)
    pkg.ops.cast(x, xb)
    assert torch.equal(xb, x.to(torch.bfloat16))
    X = _rand((2, 11, 192), 2, cuda)
    y = torch.empty(2 * 6, 192, device=cuda)
    pkg.ops.l2norm_rows(X, 5, 6, y)
    assert float((y - F.normalize(X[:, 5:].reshape(-1, 192), dim=1)).abs().max()) < 1e-6


# ------------------------------------------------------------------ fused attention
@pytest.mark.parametrize("scaled_queries", [True, False])
@pytest.mark.parametrize("B,T,H,causal", [(1, 64, 1, False), (2, 70, 2, False), (1, 577, 3, False), (3, 77, 2, True), (1, 1374, 2, False)])
def test_flash_attention_matches_reference(pkg, cuda, B, T, H, causal, scaled_queries):
    """``scaled_queries``: the stored queries already carry log2(e)/8 (as the towers' qkv epilogue leaves them) and
    the kernel is told so; otherwise it scales its query fragments itself, which rounds them to bf16 a second time."""
    D = H * 64
    c2 = pkg.ops.RAW_QUERY_SCALE_LOG2
    x = _rand((B * T, 3 * D), 5, cuda) * 1.5
    if scaled_queries:
        x[:, :D] *= c2
    qkv = x.to(torch.bfloat16)
    O = torch.full((B * T, D), float("nan"), device=cuda, dtype=torch.bfloat16)
    lse = torch.zeros(B * H, T, device=cuda)
    scale_arg = 1.0 if scaled_queries else c2
    pkg.ops.flash_attn(qkv, O, lse, B, T, H, causal, score_scale_log2=scale_arg)
    q = qkv.float().view(B, T, 3, H, 64)
    # natural-log logits: q.k/8 for raw queries, q'.k ln 2 for scaled ones
    s = torch.einsum("bthd,bshd->bhts", q[:, :, 0], q[:, :, 1]) * (0.6931471805599453 if scaled_queries else 0.125)
    if causal:
        s = s + torch.full((T, T), float("-inf"), device=cuda).triu_(1)
    ref = torch.einsum("bhts,bshd->bthd", s.softmax(-1), q[:, :, 2]).reshape(B * T, D)
    assert float((O.float() - ref).abs().max()) < 3e-2
    # the row sum is accumulated from the bf16-rounded probabilities (the same values the PV product
    # uses), so the LSE carries their rounding: 2^-9 relative per term, a few 1e-3 in the worst row; the
    # second rounding of kernel-scaled queries moves every score by up to 2^-9 of its terms on top of that
    assert float((lse.view(B, H, T) - torch.logsumexp(s, -1)).abs().max()) < (4e-3 if scaled_queries else 1.2e-2)
    if not causal:
        off, Pn = 1, T - 1
        ld = (Pn + 63) // 64 * 64
        acc = torch.zeros(Pn, ld, device=cuda)
        pkg.ops.flash_attn_mean(qkv, lse, T, H, off, Pn, 1.0 / H, acc, score_scale_log2=scale_arg)
        want = s[0].softmax(-1)[:, off:, off:].mean(0)
        assert float((acc[:, :Pn] - want).abs().max()) < (2e-3 if scaled_queries else 4e-3)
        # both kernels form the same scores, so the mean map's rows sum to what the full rows leave for the patch block
        rows = acc[:, :Pn].sum(1)
        assert float((rows - s[0].softmax(-1)[:, off:, off:].sum(-1).mean(0)).abs().max()) < 5e-3


@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 70, 2), (2, 577, 3), (1, 33, 2), (1, 1374, 1)])
def test_fp8_attention_close_to_reference(pkg, cuda, B, T, H):
    """q, k, v and the probabilities as 8-bit floats (e4m3: three mantissa bits, a relative step of 1/16), f32 softmax
    and accumulation: the output stays within 15 % of the largest value of the f32 attention at its worst element and 1.5 % on
    average, no further from it than the same quantisation emulated in torch explains, and a reused workspace holds nothing stale."""
    D = H * 64
    qkv = (_rand((B * T, 3 * D), 5, cuda) * 1.5).to(torch.bfloat16)
    O = torch.full((B * T, D), float("nan"), device=cuda, dtype=torch.bfloat16)
    ws = {}
    pkg.ops.flash_attn_fp8(qkv, O, B, T, H, ws)
    q = qkv.float().view(B, T, 3, H, 64)
    s = torch.einsum("bthd,bshd->bhts", q[:, :, 0], q[:, :, 1]) * 0.125
    ref = torch.einsum("bhts,bshd->bthd", s.softmax(-1), q[:, :, 2]).reshape(B * T, D)
    err = (O.float() - ref).abs()
    scale = float(ref.abs().max())
    assert bool(torch.isfinite(O.float()).all())
    assert float(err.max()) < 0.15 * scale and float(err.mean()) < 0.015 * scale, (float(err.max()), float(err.mean()), scale)
    # the same quantisation emulated in torch (per image, head and operand: 448 / amax, e4m3; probabilities as
    # 256 p relative to the row maximum): the kernel's error against the f32 attention is the error this explains
    f8 = torch.float8_e4m3fn
    x = q.permute(0, 3, 2, 1, 4)                                      # [B, H, 3, T, 64]
    amax = x.abs().amax(dim=(3, 4), keepdim=True).clamp_min(1e-20)
    xq = (x * (448.0 / amax)).to(f8).float() * (amax / 448.0)
    se = torch.einsum("bhtd,bhsd->bhts", xq[:, :, 0], xq[:, :, 1]) * 0.125
    pe = torch.exp(se - se.amax(-1, keepdim=True))
    pq = (pe * 256.0).to(f8).float() / 256.0
    emu = (torch.einsum("bhts,bhsd->bthd", pq, xq[:, :, 2]) / pe.sum(-1).permute(0, 2, 1)[..., None]).reshape(B * T, D)
    emu_err = float((emu - ref).abs().mean())
    assert float(err.mean()) < 1.5 * emu_err + 1e-4, (float(err.mean()), emu_err)
    assert float((O.float() - emu).abs().mean()) < 1.5 * emu_err + 2e-3 * scale
    # a second call through the same workspace with other data gives that data's result (nothing stale is read)
    qkv2 = (_rand((B * T, 3 * D), 6, cuda) * 0.7).to(torch.bfloat16)
    O2 = torch.empty_like(O)
    pkg.ops.flash_attn_fp8(qkv2, O2, B, T, H, ws)
    q2 = qkv2.float().view(B, T, 3, H, 64)
    s2 = torch.einsum("bthd,bshd->bhts", q2[:, :, 0], q2[:, :, 1]) * 0.125
    ref2 = torch.einsum("bhts,bshd->bthd", s2.softmax(-1), q2[:, :, 2]).reshape(B * T, D)
    assert float((O2.float() - ref2).abs().max()) < 0.15 * float(ref2.abs().max())


def _attention_reference(qkv, B, T, H):
    q = qkv.float().view(B, T, 3, H, 64)
    s = torch.einsum("bthd,bshd->bhts", q[:, :, 0], q[:, :, 1]) * 0.6931471805599453          # stored queries carry log2(e)/8
    return torch.einsum("bhts,bshd->bthd", s.softmax(-1), q[:, :, 2]).reshape(B * T, H * 64), torch.logsumexp(s, -1).reshape(B * H, T)


@pytest.mark.parametrize("B,T,H", [(8, 577, 16), (16, 577, 16), (12, 600, 12), (9, 321, 13)])
def test_flash_attention_walking_workgroups(pkg, cuda, B, T, H):
    """Launches with more (image-head, query block) items than the chip holds workgroups: every workgroup walks
    several items, the next one's queries and first tiles brought in under the current one, and - from twice as many
    items as workgroups - takes its third and later items from the launch's counters.  Against the f32 softmax; a
    second launch on the same stream (it finds the counters as the first left them) gives the same bits, and so do two
    launches side by side on streams of their own."""
    ops = pkg.ops
    D = H * 64
    x = _rand((B * T, 3 * D), 41, cuda) * 1.2
    x[:, :D] *= ops.RAW_QUERY_SCALE_LOG2
    qkv = x.to(torch.bfloat16)
    ref, want_lse = _attention_reference(qkv, B, T, H)

    def run(stream=None):
        O = torch.full((B * T, D), float("nan"), device=cuda, dtype=torch.bfloat16)
        lse = torch.zeros(B * H, T, device=cuda)
        if stream is None:
            ops.flash_attn(qkv, O, lse, B, T, H, False, score_scale_log2=1.0)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                ops.flash_attn(qkv, O, lse, B, T, H, False, score_scale_log2=1.0)
        return O, lse

    O1, l1 = run()
    assert bool(torch.isfinite(O1.float()).all())
    assert float((O1.float() - ref).abs().max()) < 3e-2
    assert float((l1 - want_lse).abs().max()) < 4e-3
    O2, l2 = run()
    assert torch.equal(O1, O2) and torch.equal(l1, l2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    (Oa, la), (Ob, lb) = run(sa), run(sb)
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    torch.cuda.synchronize()
    assert torch.equal(Oa, O1) and torch.equal(Ob, O1) and torch.equal(la, l1) and torch.equal(lb, l1)
    O3, _ = run()                                  # and the default stream's counters are still in order
    assert torch.equal(O3, O1)


def test_flash_attention_walking_workgroups_second_pass(pkg, cuda):
    """A few items of a walking launch leave the absolute scale's range (one key lined up with one query at +150 nats,
    another at key 0): their workgroups work them again with the running maximum - the rings set up afresh in the
    middle of a walk, the next item's queries staged twice - and go on to their next items."""
    ops = pkg.ops
    B, T, H = 10, 577, 16
    D = H * 64
    x = _rand((B * T, 3 * D), 43, cuda) * 0.8
    for b, t, key, head, nats in [(0, 5, 300, 0, 150.0), (3, 200, 0, 7, 120.0), (9, 576, 575, 15, 150.0), (4, 130, 64, 2, -150.0)]:
        q = x[b * T + t, head * 64:head * 64 + 64]
        x[b * T + key, D + head * 64:D + head * 64 + 64] = q / q.norm() ** 2 * (8.0 * nats)       # q . k / 8 = nats
    x[:, :D] *= ops.RAW_QUERY_SCALE_LOG2
    qkv = x.to(torch.bfloat16)
    ref, want_lse = _attention_reference(qkv, B, T, H)
    O = torch.full((B * T, D), float("nan"), device=cuda, dtype=torch.bfloat16)
    lse = torch.zeros(B * H, T, device=cuda)
    ops.flash_attn(qkv, O, lse, B, T, H, False, score_scale_log2=1.0)
    assert bool(torch.isfinite(O.float()).all())
    assert float((O.float() - ref).abs().max()) < 3e-2
    assert float(((lse - want_lse).abs() / want_lse.abs().clamp_min(1.0)).max()) < 1.2e-2
    O2 = torch.empty_like(O)
    ops.flash_attn(qkv, O2, lse, B, T, H, False, score_scale_log2=1.0)
    assert torch.equal(O, O2)


def test_flash_attention_spiked_scores(pkg, cuda):
    """A key that dominates a row late in the sequence forces the online-softmax rescale."""
    T, H = 200, 1
    qkv = (_rand((T, 192), 9, cuda) * 0.5)
    qkv[150, 64:128] = qkv[3, 0:64] * 40.0     # key 150 aligned with query 3
    qkv = qkv.to(torch.bfloat16)
    O = torch.zeros(T, 64, device=cuda, dtype=torch.bfloat16)
    pkg.ops.flash_attn(qkv, O, None, 1, T, H, False)
    q = qkv.float()
    s = (q[:, 0:64] @ q[:, 64:128].t()) * 0.125
    ref = s.softmax(-1) @ q[:, 128:192]
    assert float((O.float() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("case", ["moderate_spike", "late_huge_spike", "large_first_tile", "very_negative_rows", "two_regimes_in_one_wave"])
def test_flash_attention_score_ranges(pkg, cuda, case):
    """The kernel works full key tiles after the first on an absolute exp2 scale when the first tile's row maxima
    are moderate, and falls back to the running maximum when a score leaves the head room: every regime, and the
    switch between them in the middle of a sequence, against the f32 softmax (output and log-sum-exp)."""
    T, H = 330, 1                                   # six key tiles: one classic, four full, one ragged
    qkv = _rand((T, 192), 21, cuda) * 0.5
    unit = qkv[7, 0:64] / qkv[7, 0:64].norm()
    if case == "moderate_spike":                    # +40 nats on one key of a later tile: stays on the absolute scale
        qkv[200, 64:128] = unit * (8 * 40.0 / float(qkv[7, 0:64].norm()))
    elif case == "late_huge_spike":                 # +120 nats: beyond the head room, classic from that tile on
        qkv[200, 64:128] = unit * (8 * 120.0 / float(qkv[7, 0:64].norm()))
    elif case == "large_first_tile":                # a first-tile maximum of +60 nats: never leaves the classic path
        qkv[5, 64:128] = unit * (8 * 60.0 / float(qkv[7, 0:64].norm()))
    elif case == "very_negative_rows":              # every score of the queries far below zero
        qkv[:, 64:128] = 0.05 * qkv[:, 64:128] - 30.0 * unit
        qkv[:, 0:64] = 0.05 * qkv[:, 0:64] + 30.0 * unit     # q.k/8 about -112 for all pairs
    else:                                           # query 40 (second block of the first wave) alone has a late huge spike
        u2 = qkv[40, 0:64] / qkv[40, 0:64].norm()
        qkv[250, 64:128] = u2 * (8 * 150.0 / float(qkv[40, 0:64].norm()))
    qkv = qkv.to(torch.bfloat16)
    O = torch.zeros(T, 64, device=cuda, dtype=torch.bfloat16)
    lse = torch.zeros(H, T, device=cuda)
    pkg.ops.flash_attn(qkv, O, lse, 1, T, H, False)
    q = qkv.float()
    s = (q[:, 0:64] @ q[:, 64:128].t()) * 0.125
    ref = s.softmax(-1) @ q[:, 128:192]
    assert bool(torch.isfinite(O.float()).all())
    assert float((O.float() - ref).abs().max()) < 3e-2
    want = torch.logsumexp(s, -1)
    assert float(((lse[0] - want).abs() / want.abs().clamp_min(1.0)).max()) < 1.2e-2      # raw queries: scaled in the kernel


@pytest.mark.parametrize("case", ["plain", "huge_key0", "very_negative_key0", "late_huge_spike", "key0_dominates", "two_heads_batch",
                                  "two_tiles", "three_tiles"])
def test_flash_attention_class_token_sequences(pkg, cuda, case):
    """Sequences of 64 k + 1 tokens: key 0 is the initial state of the online softmax (worked on the vector ALU) and
    the tiles cover keys 1 .. T-1.  Every regime of that initial state - moderate (absolute scale), out of range either
    way (running maximum from the start), a later score beyond the head room - against the f32 softmax."""
    T, H, B = 321, 1, 1                             # key 0 + five full tiles
    if case == "two_heads_batch":
        T, H, B = 577, 2, 2
    elif case == "two_tiles":                       # the shortest sequence this form takes
        T, H, B = 129, 2, 2
    elif case == "three_tiles":
        T, H, B = 193, 1, 3
    qkv = _rand((B * T, 192 * H), 31, cuda) * 0.5
    if case in ("plain", "huge_key0", "very_negative_key0", "late_huge_spike", "key0_dominates"):
        unit = qkv[7, 0:64] / qkv[7, 0:64].norm()
        n7 = float(qkv[7, 0:64].norm())
        if case == "huge_key0":                     # q7 . k0 / 8 = +60 nats: beyond the absolute range
            qkv[0, 64:128] = unit * (8 * 60.0 / n7)
        elif case == "very_negative_key0":
            qkv[0, 64:128] = -unit * (8 * 60.0 / n7)
        elif case == "late_huge_spike":             # +120 nats on a key of the fourth tile
            qkv[200, 64:128] = unit * (8 * 120.0 / n7)
        elif case == "key0_dominates":              # +30 nats: inside the range, every other probability tiny
            qkv[0, 64:128] = unit * (8 * 30.0 / n7)
    qkv = qkv.to(torch.bfloat16)
    D = 64 * H
    O = torch.zeros(B * T, D, device=cuda, dtype=torch.bfloat16)
    lse = torch.zeros(B * H, T, device=cuda)
    pkg.ops.flash_attn(qkv, O, lse, B, T, H, False)
    q = qkv.float().view(B, T, 3, H, 64)
    s = torch.einsum("bthd,bshd->bhts", q[:, :, 0], q[:, :, 1]) * 0.125
    ref = torch.einsum("bhts,bshd->bthd", s.softmax(-1), q[:, :, 2]).reshape(B * T, D)
    assert bool(torch.isfinite(O.float()).all())
    assert float((O.float() - ref).abs().max()) < 3e-2
    want = torch.logsumexp(s, -1).reshape(B * H, T)
    assert float(((lse - want).abs() / want.abs().clamp_min(1.0)).max()) < 1.2e-2


# ------------------------------------------------------------------ alignment / refinement
def test_vva_prior_and_minmax(pkg, cuda):
    R, P, ld = 162, 81, 128
    S = _rand((R, ld), 1, cuda)
    fg = (torch.rand(R, generator=torch.Generator().manual_seed(2)) < 0.3).to(torch.uint8).to(cuda)
    out = torch.empty(P, device=cuda)
    pkg.ops.vva_prior(S, fg, P, torch.zeros(4 * 32 * P, device=cuda), 32, out)
    Sf, Sb = S[fg.bool(), :P], S[~fg.bool(), :P]
    v = Sf.mean(0) * Sf.max(0).values - Sb.mean(0) * Sb.max(0).values
    assert float((out - opipe.minmax(v)).abs().max()) < 1e-5
    ones = torch.ones(R, dtype=torch.uint8, device=cuda)    # no background rows: foreground term only
    pkg.ops.vva_prior(S, ones, P, torch.zeros(4 * 32 * P, device=cuda), 32, out)
    v = S[:, :P].mean(0) * S[:, :P].max(0).values
    assert float((out - opipe.minmax(v)).abs().max()) < 1e-5
    x = _rand((1089,), 3, cuda)
    y = torch.empty_like(x)
    pkg.ops.minmax(x, y)
    assert float((y - opipe.minmax(x)).abs().max()) < 1e-6


@pytest.mark.parametrize("G,thr", [(9, 0.8), (33, 0.4), (37, 0.8), (37, 0.4)])
def test_box_mask_matches_oracle(pkg, cuda, G, thr):
    from scipy import ndimage
    rng = np.random.default_rng(G)
    for trial in range(6):
        prior = ndimage.uniform_filter(rng.random((G, G)).astype(np.float32), 3 + trial % 3)
        prior = (prior - prior.min()) / (1e-7 + prior.max() - prior.min())
        prior = ((np.floor(prior * 254) + 0.5) / 255).astype(np.float32)   # mid-bin values: trunc(x*255) is robust
        want = opipe.box_mask(torch.from_numpy(prior.astype(np.float32)), thr)
        got = torch.empty(G * G, device=cuda)
        pkg.ops.box_mask(torch.from_numpy(prior.astype(np.float32)).to(cuda), thr, got)
        assert torch.equal(got.view(G, G).cpu(), want)
    got = torch.ones(G * G, device=cuda)
    pkg.ops.box_mask(torch.zeros(G, G, device=cuda), thr, got)
    assert float(got.sum()) == 0.0


def test_refinement_chain_matches_oracle(pkg, cuda):
    import importlib
    PIR = importlib.import_module(pkg.__name__ + ".mars.components.PriorInformationRefinementModule")
    for G, thr in ((9, 0.8), (33, 0.4)):
        P = G * G
        A = torch.rand(P, P, generator=torch.Generator().manual_seed(G)) ** 4 + 1e-3
        A = A / A.sum(1, keepdim=True)
        prior = opipe.minmax(torch.rand(G, G, generator=torch.Generator().manual_seed(G + 1)))
        want = opipe.pir_refine(prior, A, thr)
        mod = PIR.PriorInformationRefinementModule(thr, 8, cuda)
        acc = mod.new_accumulator(P)
        acc[:, :P].copy_(A.to(cuda))
        got = mod.refine(prior.to(cuda), acc)
        assert float((got.cpu() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
        # reference signature: list of per-block [1, h, T, T] maps with a class token
        T = P + 1
        maps = [torch.rand(1, 2, T, T, generator=torch.Generator().manual_seed(i)).softmax(-1) for i in range(3)]
        want2 = opipe.pir_refine(prior, opipe.mean_attention(maps, 2, 0), thr)
        mod2 = PIR.PriorInformationRefinementModule(thr, 2, cuda)
        got2 = mod2.compute(prior.to(cuda), [m.to(cuda) for m in maps])
        assert float((got2.cpu() - want2).abs().max()) < 1e-5 * max(1.0, float(want2.abs().max()))


# ------------------------------------------------------------------ resampling
def test_resample_kernels(pkg, cuda):
    import importlib
    from PIL import Image
    rs = importlib.import_module(pkg.__name__ + ".resample")
    dev = lambda t: tuple(torch.from_numpy(a).to(cuda).contiguous() for a in t)
    g = torch.Generator().manual_seed(0)
    for n_in, n_out, mode in [(126, 128, "bicubic"), (126, 56, "bilinear"), (518, 336, "bilinear"), (518, 528, "bicubic")]:
        x = torch.rand(3, n_in, n_in, generator=g)
        ref = F.interpolate(x[None], size=(n_out, n_out), mode=mode, align_corners=False, antialias=True)[0]
        tab = dev(rs.torch_aa_table(n_in, n_out, mode))
        tmp = torch.empty(3, n_in, n_out, device=cuda)
        out = torch.empty(3, n_out, n_out, device=cuda)
        pkg.ops.resample_axis(x.to(cuda), 3, n_in, n_in, 1, tab, tmp)
        pkg.ops.resample_axis(tmp, 3, n_in, n_out, 0, tab, out)
        assert float((out.cpu() - ref).abs().max()) < 3e-6
    # mask branch: u8 quantisation in front, (x - 0.5) / 0.26 behind
    m = (torch.rand(2, 126, 126, generator=g) > 0.5).float()
    tab = dev(rs.torch_aa_table(126, 56, "bilinear"))
    tmp, out = torch.empty(2, 126, 56, device=cuda), torch.empty(2, 56, 56, device=cuda)
    pkg.ops.resample_axis(m.to(cuda), 2, 126, 126, 1, tab, tmp, quant_u8=True)
    pkg.ops.resample_axis(tmp, 2, 126, 56, 0, tab, out, shift=0.5, scale=0.26)
    want = torch.stack([opipe.alphaclip_mask_transform(mm, 56)[0] for mm in m])
    assert float((out.cpu() - want).abs().max()) < 1e-5
    # Pillow 8-bit path: bit exact
    img = torch.rand(3, 126, 126, generator=g)
    u8 = torch.empty(3, 126, 126, dtype=torch.uint8, device=cuda)
    pkg.ops.f32_to_u8(img.to(cuda), u8)
    assert torch.equal(u8.cpu(), (img * 255).to(torch.uint8))
    tab = dev(rs.pil_table_u8(126, 56))
    t8, o8 = torch.empty(3, 126, 56, dtype=torch.uint8, device=cuda), torch.empty(3, 56, 56, dtype=torch.uint8, device=cuda)
    pkg.ops.resample_axis_u8(u8, 3, 126, 126, 1, tab, t8)
    pkg.ops.resample_axis_u8(t8, 3, 126, 56, 0, tab, o8)
    pil = Image.fromarray((img.permute(1, 2, 0).numpy() * 255).astype(np.uint8), "RGB").resize((56, 56), Image.BICUBIC)
    assert torch.equal(o8.cpu(), torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1))


# ------------------------------------------------------------------ pooling / ranking / merge / IoU
@pytest.mark.parametrize("N,S,G", [(6, 126, 9), (3, 518, 37), (4, 100, 7)])
def test_pool_and_stats(pkg, cuda, N, S, G):
    g = torch.Generator().manual_seed(S)
    masks = (torch.rand(N, S, S, generator=g) > 0.9995).float()
    masks[0] = 0
    masks[0, 3:40, 5:60] = 1
    bits = torch.zeros(N, G * G, dtype=torch.uint8, device=cuda)
    pkg.ops.pool_masks(masks.to(cuda), G, bits)
    want = (F.adaptive_max_pool2d(masks[:, None], (G, G))[:, 0] > 0).reshape(N, -1)
    assert torch.equal(bits.cpu().bool(), want)
    vva, vta = torch.rand(G * G, generator=g), torch.rand(G * G, generator=g)
    cnt = torch.zeros(N, dtype=torch.int32, device=cuda); ucnt = torch.zeros(1, dtype=torch.int32, device=cuda)
    svv, svt = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda)
    pkg.ops.proposal_stats(bits, N, G * G, vva.to(cuda), vta.to(cuda), cnt, svv, svt, ucnt)
    assert torch.equal(cnt.cpu().long(), want.sum(1))
    assert int(ucnt) == int(want.any(0).sum())
    assert float((svv.cpu() - (want * vva).sum(1)).abs().max()) < 1e-4
    assert float((svt.cpu() - (want * vta).sum(1)).abs().max()) < 1e-4


def test_rank_merge_matches_oracle_arithmetic(pkg, cuda):
    rng = np.random.default_rng(0)
    for trial, N in enumerate((6, 64, 256)):
        emd = 0.9 + 0.1 * rng.random(N)
        aclip = rng.normal(0.1, 0.05, N).astype(np.float32)
        cnt = rng.integers(1, 80, N).astype(np.int32)
        ucnt = np.array([int(cnt.max()) + 5], np.int32)
        svv = (rng.random(N) * cnt).astype(np.float32)
        svt = (rng.random(N) * cnt).astype(np.float32)
        if trial == 1:
            svv[5], svt[5], cnt[5], emd[5], aclip[5] = svv[4], svt[4], cnt[4], emd[4], aclip[4]   # an exact tie
        t = lambda a, d=None: torch.from_numpy(a).to(cuda)
        score = torch.zeros(N, dtype=torch.float64, device=cuda)
        order = torch.zeros(N, dtype=torch.int32, device=cuda)
        keep = torch.zeros(N, dtype=torch.uint8, device=cuda)
        pkg.ops.rank(t(emd), t(aclip), t(cnt), t(svv), t(svt), t(ucnt), N, 0.85, 0.55, 0.95, score, order, keep)
        cov = cnt / (1e-7 + ucnt[0])
        pvv = 0.85 * (svv.astype(np.float64) / (1e-7 + cnt)) + 0.15 * cov
        pvt = 0.85 * (svt.astype(np.float64) / (1e-7 + cnt)) + 0.15 * cov
        es = (emd - emd.min()) / (1e-7 + emd.max() - emd.min())
        # the AlphaCLIP min-max stays float32, as the reference's numpy expression does (float32 entries,
        # Python-float epsilon: FilteringMergingModule.py:128-132)
        as_ = (aclip - aclip.min()) / (np.float32(1e-7) + aclip.max() - aclip.min())
        assert as_.dtype == np.float32
        want = (es + as_.astype(np.float64) + pvv + pvt) / 4
        assert np.abs(score.cpu().numpy() - want).max() < 1e-12
        want_order = sorted(range(N), key=lambda i: score.cpu().numpy()[i], reverse=True)
        assert order.cpu().tolist() == want_order
        top = score.cpu().numpy()[want_order[0]]
        lb = 0.95 * top if top < 0.55 else 0.55
        assert keep.cpu().numpy().astype(bool).tolist() == (score.cpu().numpy() >= lb).tolist()
        masks = (torch.rand(N, 50, 50, generator=torch.Generator().manual_seed(N)) > 0.98).float()
        merged = torch.empty(50, 50, device=cuda)
        pkg.ops.merge_masks(masks.to(cuda), keep, merged)
        wantm = (masks[keep.cpu().bool()].sum(0) > 0).float()
        assert torch.equal(merged.cpu(), wantm)


def test_cosine_scores(pkg, cuda):
    f, t = _rand((7, 768), 1, cuda), _rand((1, 768), 2, cuda)
    s = torch.empty(7, device=cuda)
    pkg.ops.cosine_scores(f, t, s)
    want = (F.normalize(f, dim=1) @ F.normalize(t, dim=1).t())[:, 0]
    assert float((s - want).abs().max()) < 1e-6


def test_iou_counts_match_evaluator(pkg, cuda):
    import importlib
    g = torch.Generator().manual_seed(1)
    pred = (torch.rand(1, 126, 126, generator=g) > 0.5).float()
    gt = (torch.rand(1, 126, 126, generator=g) > 0.6).float()
    ign = ((torch.rand(1, 126, 126, generator=g) > 0.9) & (gt == 0)).float()
    ev = importlib.import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
    ev.initialize()
    for ignore in (None, ign):
        batch = {"query_mask": gt.to(cuda)}
        if ignore is not None:
            batch["query_ignore_idx"] = ignore.to(cuda)
        ai, au = ev.classify_prediction(pred.to(cuda), batch)
        wi, wu = opipe.classify_prediction(pred[0], gt[0], None if ignore is None else ignore[0])
        assert torch.equal(ai[:, 0].cpu(), wi) and torch.equal(au[:, 0].cpu(), wu)


# ------------------------------------------------------------------ GradCAM backward pieces
def test_backward_pieces_against_autograd(pkg, cuda):
    T, D = 19, 192
    x = _rand((T, D), 1, cuda).requires_grad_(True)
    w, b = _rand((D,), 2, cuda), _rand((D,), 3, cuda)
    dy = _rand((T, D), 4, cuda)
    (gx,) = torch.autograd.grad((F.layer_norm(x, (D,), w, b, 1e-5) * dy).sum(), x)
    dx = torch.empty(T, D, device=cuda)
    add = _rand((T, D), 5, cuda)
    pkg.ops.layernorm_bwd(x.detach(), w, dy, D, 1.0, 0, add, dx, 1e-5)
    assert float((dx - (gx + add)).abs().max()) < 2e-5
    vec = _rand((D,), 6, cuda)
    dyb = (vec / (T - 1)).expand(T, D).clone()
    dyb[0] = 0
    (gx2,) = torch.autograd.grad((F.layer_norm(x, (D,), w, b, 1e-5) * dyb).sum(), x)
    pkg.ops.layernorm_bwd(x.detach(), w, vec, 0, 1.0 / (T - 1), 1, None, dx, 1e-5)
    assert float((dx - gx2).abs().max()) < 2e-5
    z = _rand((T, 4 * D), 7, cuda).requires_grad_(True)
    dh = _rand((T, 4 * D), 8, cuda)
    (gz,) = torch.autograd.grad(((z * torch.sigmoid(1.702 * z)) * dh).sum(), z)
    out = torch.empty_like(dh)
    pkg.ops.quickgelu_bwd(dh, z.detach(), out)
    assert float((out - gz).abs().max()) < 1e-5
    s = _rand((3, T, 64), 9, cuda).requires_grad_(True)
    dP = _rand((3, T, 64), 10, cuda)
    p = torch.zeros(3, T, 64, device=cuda)
    pr = s[..., :T].softmax(-1)
    (gs,) = torch.autograd.grad((pr * dP[..., :T]).sum(), s)
    p[..., :T] = pr.detach()
    dS = torch.empty(3, T, 64, device=cuda)
    pkg.ops.softmax_bwd(p, dP, dS, 64, 3, T, 0.125)
    assert float((dS[..., :T] - gs[..., :T] * 0.125).abs().max()) < 1e-6 and float(dS[..., T:].abs().max()) == 0
    m = torch.empty(D, device=cuda)
    pkg.ops.col_mean(dy, 1, T, m)
    assert float((m - dy[1:].mean(0)).abs().max()) < 1e-6
    cam = torch.empty(T - 1, device=cuda)
    pkg.ops.cam_rows(dy, m, 1, T - 1, cam)
    assert float((cam - torch.relu(dy[1:] @ m)).abs().max()) < 1e-5
    # head: logits and d logits[0] / d f
    E = 64
    f = _rand((D,), 11, cuda).requires_grad_(True)
    proj, text = _rand((D, E), 12, cuda, 0.1), _rand((2, E), 13, cuda)
    gg = f @ proj
    logits = (14.285714 * (gg / gg.norm()) @ (text / text.norm(dim=1, keepdim=True)).t()).softmax(-1)
    (gf,) = torch.autograd.grad(logits[0], f)
    lo, df = torch.empty(2, device=cuda), torch.empty(D, device=cuda)
    pkg.ops.cam_head(f.detach(), proj, text, 14.285714, lo, df)
    assert float((lo - logits.detach()).abs().max()) < 1e-5
    assert float((df - gf).abs().max()) < 1e-5 * max(1.0, float(gf.abs().max()))


# ------------------------------------------------------------------ kernels of the proposal stage
def test_gather_rows_and_casts(pkg, cuda):
    ops = pkg.ops
    src = _rand((50, 96), 41, cuda)
    idx = torch.tensor([3, -1, 49, 0, 7, 7, -1, 20], dtype=torch.int32, device=cuda)
    for dt in (torch.float32, torch.bfloat16):
        s = src.to(dt)
        dst = torch.full((8, 130), 5.0, device=cuda, dtype=dt)
        ops.gather_rows(s, idx, dst, src_col=16, dst_col=32, ncols=64)
        ref = torch.full((8, 130), 5.0, device=cuda, dtype=dt)
        for r, i in enumerate(idx.tolist()):
            ref[r, 32:96] = s[i, 16:80] if i >= 0 else 0
        assert torch.equal(dst, ref)
    a, b = _rand((12, 40), 42, cuda), _rand((3, 40), 43, cuda)
    out = torch.empty(12, 40, device=cuda, dtype=torch.bfloat16)
    ops.add_cast(a, b, out)
    assert torch.equal(out, (a + b.repeat(4, 1)).to(torch.bfloat16))
    out32 = torch.empty(12, 40, device=cuda)
    ops.add_cast(a, None, out32)
    assert torch.equal(out32, a)
    for act in (1, 3):
        o = torch.empty(12, 40, device=cuda)
        ops.act_cast(a, o, act)
        assert float((o - _act(a, act)).abs().max()) < 1e-6


@pytest.mark.parametrize("B,H,gw,hd,hdp", [(2, 3, 5, 16, 64), (1, 2, 20, 80, 128), (3, 1, 14, 64, 64)])
def test_softmax_bias_rows_and_relpos(pkg, cuda, B, H, gw, hd, hdp):
    ops = pkg.ops
    T = gw * gw
    qkv = _rand((B * T, 3 * H * hdp), 44, cuda)
    Rh, Rw = _rand((gw, gw, hd), 45, cuda), _rand((gw, gw, hd), 46, cuda)
    rel_h = torch.empty(B * H * T, gw, device=cuda)
    rel_w = torch.empty(B * H * T, gw, device=cuda)
    ops.relpos_bias(qkv, B, T, H, hd, hdp, gw, Rh, Rw, rel_h, rel_w)
    q = qkv[:, : H * hdp].view(B, T, H, hdp)[..., :hd].permute(0, 2, 1, 3).reshape(B * H, gw, gw, hd)
    ref_h = torch.einsum("bhwc,hkc->bhwk", q, Rh).reshape(-1, gw)
    ref_w = torch.einsum("bhwc,wkc->bhwk", q, Rw).reshape(-1, gw)
    assert float((rel_h - ref_h).abs().max()) < 1e-4 and float((rel_w - ref_w).abs().max()) < 1e-4
    Tp = -(-T // 64) * 64                                      # rows of 25 / 196 (one wave each) and of 400 entries
    S = _rand((B * H * T, Tp), 47, cuda)
    for dt in (torch.float32, torch.bfloat16):
        P = torch.full((B * H * T, Tp), 3.0, device=cuda, dtype=dt)
        ops.softmax_bias_rows(S.view(B * H, T, Tp), P.view(B * H, T, Tp), B * H * T, T, rel_h, rel_w, gw, gw)
        bias = (ref_h[:, :, None] + ref_w[:, None, :]).reshape(-1, T)
        ref = torch.softmax(S[:, :T] + bias, dim=-1)
        assert float((P[:, :T].float() - ref).abs().max()) < (1e-5 if dt == torch.float32 else 4e-3)
        assert float(P[:, T:].float().abs().max()) == 0.0           # the K padding of the next product
    P = torch.empty(B * H * T, Tp, device=cuda)
    ops.softmax_bias_rows(S.view(B * H, T, Tp), P.view(B * H, T, Tp), B * H * T, T)
    assert float((P[:, :T] - torch.softmax(S[:, :T], -1)).abs().max()) < 1e-5


@pytest.mark.parametrize("B,H,gh,gw,hd,hdp", [(3, 2, 14, 14, 80, 80), (2, 3, 14, 14, 64, 64), (1, 2, 16, 16, 80, 128), (2, 1, 5, 13, 48, 64),
                                               (1, 1, 9, 9, 96, 128), (5, 4, 7, 3, 16, 64), (1, 2, 20, 20, 80, 80)])
def test_relpos_bias_of_bf16_queries(pkg, cuda, B, H, gh, gw, hd, hdp):
    """bf16 queries against f32 tables: grids up to 16 x 16 go through the matrix pipe (tables as three bf16 terms), larger
    ones through the vector form; both against f64 on the same operands."""
    ops = pkg.ops
    T = gh * gw
    qkv = _rand((B * T, 3 * H * hdp), 61, cuda).to(torch.bfloat16)
    Rh, Rw = _rand((gh, gh, hd), 62, cuda), _rand((gw, gw, hd), 63, cuda)
    rel_h = torch.full((B * H * T, gh), float("nan"), device=cuda)
    rel_w = torch.full((B * H * T, gw), float("nan"), device=cuda)
    ops.relpos_bias(qkv, B, T, H, hd, hdp, gw, Rh, Rw, rel_h, rel_w)
    q = qkv[:, : H * hdp].view(B, T, H, hdp)[..., :hd].permute(0, 2, 1, 3).reshape(B * H, gh, gw, hd).double()
    ref_h = torch.einsum("bhwc,hkc->bhwk", q, Rh.double()).reshape(-1, gh)
    ref_w = torch.einsum("bhwc,wkc->bhwk", q, Rw.double()).reshape(-1, gw)
    assert float((rel_h.double() - ref_h).abs().max()) < 2e-5 and float((rel_w.double() - ref_w).abs().max()) < 2e-5


# ------------------------------------------------------------------ mask decoder: small-head attention and fused row passes
@pytest.mark.parametrize("B,H,hd,Tq,Tk", [(5, 8, 32, 7, 7), (3, 8, 16, 7, 300), (3, 8, 16, 300, 7), (2, 4, 32, 9, 1000), (2, 4, 32, 1000, 9),
                                          (1, 1, 16, 1, 16), (70, 8, 16, 16, 4096)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_small_matches_torch(pkg, cuda, B, H, hd, Tq, Tk, dtype):
    """softmax(q k^T * scale) v on 16- and 32-wide heads with at most 16 tokens on one side, row-strided operands."""
    W = H * hd
    q = _rand((B * Tq, W), 1, cuda).to(dtype)
    k = _rand((B * Tk, W), 2, cuda).to(dtype)
    v = _rand((B * Tk, W), 3, cuda).to(dtype)
    out = torch.full((B * Tq, W + 64), float("nan"), device=cuda, dtype=dtype)
    scale = 1.0 / math.sqrt(hd)
    pkg.ops.attn_small(q, k, v, out, B, H, hd, Tq, Tk, scale)
    qh = q.double().view(B, Tq, H, hd).transpose(1, 2)
    kh = k.double().view(B, Tk, H, hd).transpose(1, 2)
    vh = v.double().view(B, Tk, H, hd).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(B * Tq, W)
    got = out[:, :W].double()
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2          # bf16: one rounding of the result (2^-8 relative, |out| <~ 3)
    assert float((got - ref).abs().max()) <= tol
    assert bool(torch.isnan(out[:, W:].float()).all())        # columns beyond the heads are not written
    # one set of queries, or of keys and values, for every prompt
    for qs, ks in ((True, False), (False, True)):
        out2 = torch.empty(B * Tq, W, device=cuda, dtype=dtype)
        pkg.ops.attn_small(q[:Tq] if qs else q, k[:Tk] if ks else k, v[:Tk] if ks else v, out2, B, H, hd, Tq, Tk, scale, q_shared=qs, kv_shared=ks)
        q2 = qh[:1].expand(B, -1, -1, -1) if qs else qh
        k2, v2 = (kh[:1].expand(B, -1, -1, -1), vh[:1].expand(B, -1, -1, -1)) if ks else (kh, vh)
        ref2 = (torch.softmax(q2 @ k2.transpose(-1, -2) * scale, -1) @ v2).transpose(1, 2).reshape(B * Tq, W)
        assert float((out2.double() - ref2).abs().max()) <= tol


@pytest.mark.parametrize("rows,D,add_rows", [(12, 256, 4), (4096 * 3, 256, 4096), (7, 64, 7), (10, 1024, 5), (9, 2048, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("norm", [False, True])
def test_stream_copies(pkg, cuda, rows, D, add_rows, dtype, norm):
    x = _rand((rows, D), 4, cuda) + 0.5
    add = _rand((add_rows, D), 5, cuda)
    w, b = _rand((D,), 6, cuda) + 1.0, _rand((D,), 7, cuda)
    y = F.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5) if norm else x.double()
    c0 = torch.empty(rows, D, device=cuda, dtype=dtype); c1 = torch.empty_like(c0)
    x_in = x.clone()
    pkg.ops.stream_copies(x_in, c0, c1, add=add, norm=(w, b) if norm else None, eps=1e-5)
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    ref0 = y + add.double().repeat(rows // add_rows, 1)
    assert float((c1.double() - y).abs().max()) <= tol
    assert float((c0.double() - ref0).abs().max()) <= tol
    assert float((x_in.double() - y).abs().max()) <= 1e-5     # the f32 stream holds the normalised rows (or is untouched)
    # one copy only, no write-back
    x2 = x.clone(); only = torch.empty_like(c0)
    pkg.ops.stream_copies(x2, None, only, norm=(w, b) if norm else None, write_back=False)
    assert torch.equal(x2, x) and torch.equal(only, c1)


@pytest.mark.parametrize("rows,D,ldo", [(100, 64, 64), (16385, 64, 64), (33, 40, 64), (5, 256, 256), (17, 128, 192)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1])
def test_layernorm_act_rows(pkg, cuda, rows, D, ldo, dtype, act):
    x = _rand((rows, D), 8, cuda) * 2 + 0.3
    w, b = _rand((D,), 9, cuda) + 1.0, _rand((D,), 10, cuda)
    out = torch.full((rows, ldo), 7.0, device=cuda, dtype=dtype)
    pkg.ops.layernorm_act_rows(x, w, b, out, 1e-6, act)
    ref = F.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
    if act:
        ref = F.gelu(ref)
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    assert float((out[:, :D].double() - ref).abs().max()) <= tol
    assert bool((out[:, D:] == 7.0).all())


@pytest.mark.parametrize("B,P,C,ldh,ldu", [(3, 1000, 32, 32, 32), (2, 65536, 32, 32, 32), (5, 257, 64, 64, 64), (4, 300, 20, 32, 24)])
def test_hyper_product(pkg, cuda, B, P, C, ldh, ldu):
    hyper = _rand((B, 4, ldh), 11, cuda)
    up = _rand((B * P, ldu), 12, cuda)
    masks = torch.full((B, 4, P), float("nan"), device=cuda)
    pkg.ops.hyper_product(hyper, up, masks, B, P, C)
    ref = hyper[:, :, :C].double() @ up[:, :C].double().view(B, P, C).transpose(1, 2)
    assert float((masks.double() - ref).abs().max()) <= 1e-4


@pytest.mark.parametrize("prompts,G", [(3, 32), (1, 8), (4, 12), (2, 64)])
def test_sam_upscale_ln_gelu_matches_torch(pkg, cuda, prompts, G):
    """ConvTranspose2d(256, 64, 2, 2) + LayerNorm2d + GELU as one kernel against the torch layers on the same bf16 operands."""
    x = _rand((prompts * G * G, 256), 21, cuda).to(torch.bfloat16)
    wt = (_rand((256, 64, 2, 2), 22, cuda) * 0.08).to(torch.bfloat16)
    cb, lw, lb = _rand((64,), 23, cuda) * 0.2, _rand((64,), 24, cuda) * 0.3 + 1.0, _rand((64,), 25, cuda) * 0.2
    rows = wt.float().permute(2, 3, 1, 0).reshape(256, 256)
    runs = rows.view(256, 32, 8).permute(1, 0, 2).contiguous().to(torch.bfloat16)
    out = torch.full((prompts * 4 * G * G, 64), float("nan"), device=cuda, dtype=torch.bfloat16)
    pkg.ops.sam_upscale_ln_gelu(x, runs, torch.stack([cb, lw, lb]).contiguous(), out, prompts, G, 1e-6)
    xi = x.double().view(prompts, G, G, 256).permute(0, 3, 1, 2)
    y = F.conv_transpose2d(xi, wt.double(), cb.double(), stride=2)                      # [B, 64, 2G, 2G]
    y = F.layer_norm(y.permute(0, 2, 3, 1), (64,), lw.double(), lb.double(), 1e-6)
    ref = F.gelu(y).reshape(prompts * 4 * G * G, 64)
    err = (out.double() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) <= 5e-3                                # one bf16 rounding of the result


@pytest.mark.parametrize("prompts,G2", [(3, 64), (2, 32), (1, 128)])
def test_sam_upscale_masks_matches_torch(pkg, cuda, prompts, G2):
    """ConvTranspose2d(64, 32, 2, 2) + GELU + the product with the four hypernetwork vectors as one kernel."""
    u = _rand((prompts * G2 * G2, 64), 31, cuda).to(torch.bfloat16)
    wt = (_rand((64, 32, 2, 2), 32, cuda) * 0.15).to(torch.bfloat16)
    cb = _rand((32,), 33, cuda) * 0.2
    hyper = _rand((prompts, 4, 32), 34, cuda)
    masks = torch.full((prompts, 4, 2 * G2, 2 * G2), float("nan"), device=cuda)
    pkg.ops.sam_upscale_masks(u, wt.float().permute(2, 3, 1, 0).reshape(128, 64).contiguous().to(torch.bfloat16), cb, hyper, masks, prompts, G2)
    ui = u.double().view(prompts, G2, G2, 64).permute(0, 3, 1, 2)
    up = F.gelu(F.conv_transpose2d(ui, wt.double(), cb.double(), stride=2))             # [B, 32, 2 G2, 2 G2]
    ref = torch.einsum("bmc,bchw->bmhw", hyper.double(), up)
    assert float((masks.double() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


@pytest.mark.parametrize("prompts,Ni,Tk,shared,write_y", [(3, 64, 7, False, True), (2, 256, 8, True, True), (5, 32, 3, False, False), (1, 4096, 7, True, True)])
def test_sam_i2t_block_matches_torch(pkg, cuda, prompts, Ni, Tk, shared, write_y):
    """Image -> token attention + output projection + residual + LayerNorm + the bf16 copy as one kernel, against the same
    steps in f64 on the same bf16 operands (q, the probabilities and o are rounded to bf16 inside the kernel)."""
    bf = torch.bfloat16
    rows = prompts * Ni
    qin = _rand((Ni if shared else rows, 256), 41, cuda).to(bf)
    resid = _rand((Ni if shared else rows, 256), 42, cuda)
    wq = (_rand((128, 256), 43, cuda) * 0.08).to(bf)
    qbias = _rand((Ni, 128), 44, cuda) * 0.5                      # key_pe Wq^T + bq of the caller
    wo, bo = (_rand((256, 128), 45, cuda) * 0.1).to(bf), _rand((256,), 46, cuda) * 0.2
    lw, lb = _rand((256,), 47, cuda) * 0.3 + 1.0, _rand((256,), 48, cuda) * 0.2
    k, v = _rand((prompts * Tk, 128), 49, cuda).to(bf), _rand((prompts * Tk, 128), 50, cuda).to(bf)
    slot = torch.tensor([16 * h + 8 * (e // 4) + 4 * g + e % 4 for h in range(8) for g in range(2) for e in range(8)], device=cuda)
    wq_runs = wq.view(128, 32, 8).permute(1, 0, 2).contiguous()
    wo_runs = wo[:, slot].view(256, 16, 8).permute(1, 0, 2).contiguous()
    prm = torch.cat([bo, lw, lb]).contiguous()
    kvo = torch.empty(prompts, 8, 64, 2, 8, device=cuda, dtype=bf)
    pkg.ops.sam_i2t_kv_operands(k, v, prompts, Tk, kvo)
    y = torch.full((rows, 256), float("nan"), device=cuda) if write_y else None
    c1 = torch.empty(rows, 256, device=cuda, dtype=bf)
    pkg.ops.sam_i2t_block(qin, wq_runs, wo_runs, qbias, prm, kvo, Tk, resid, prompts, Ni, 1e-5, y, c1)
    # reference
    x = qin.double().repeat(prompts if shared else 1, 1)
    r = resid.double().repeat(prompts if shared else 1, 1)
    q = (x @ wq.double().t() + qbias.double().repeat(prompts, 1)).to(bf).double().view(prompts, Ni, 8, 16).transpose(1, 2)        # [B, 8, Ni, 16]
    kh = k.double().view(prompts, Tk, 8, 16).transpose(1, 2)
    vh = v.double().view(prompts, Tk, 8, 16).transpose(1, 2)
    o = (torch.softmax(q @ kh.transpose(-1, -2) * 0.25, -1) @ vh).transpose(1, 2).reshape(rows, 128)
    z = r + o @ wo.double().t() + bo.double()
    ref = F.layer_norm(z, (256,), lw.double(), lb.double(), 1e-5)
    # bf16 roundings of the probabilities and of o ahead of a K = 128 product: a few 1e-3 of the unit-variance result
    if write_y:
        assert float((y.double() - ref).abs().max()) <= 2e-2
        assert torch.equal(c1, y.to(bf))                          # the copy is the rounding of what was written
    assert float((c1.double() - ref).abs().max()) <= 5e-2


@pytest.mark.parametrize("prompts,Ni,Tq", [(5, 256, 7), (2, 4096, 7), (9, 64, 8), (3, 32, 1), (4, 96, 5)])
def test_sam_t2i_block_matches_torch(pkg, cuda, prompts, Ni, Tq):
    """Token -> image attention with the key / value projections of the stream inside the kernel, against the separate steps
    in f64 on the same bf16 operands (k, v and the probabilities are rounded to bf16 inside the kernel)."""
    bf = torch.bfloat16
    x = _rand((prompts * Ni, 256), 61, cuda).to(bf)
    pe = _rand((Ni, 256), 62, cuda)
    wk, bk = (_rand((128, 256), 63, cuda) * 0.06).to(bf), _rand((128,), 64, cuda)
    wv, bv = (_rand((128, 256), 65, cuda) * 0.06).to(bf), _rand((128,), 66, cuda) * 0.3
    q = (_rand((prompts * Tq, 128), 67, cuda) * 1.5).to(bf)
    runs = lambda w: w.view(128, 32, 8).permute(1, 0, 2).contiguous()
    kpe = pe.to(bf).double() @ wk.double().t() + bk.double()                # (with the bias: it cancels)
    kpe_ops = pkg.ops.sam_t2i_kpe_operands(kpe.float())
    q_ops = torch.empty(prompts, 8, 64, 8, device=cuda, dtype=bf)
    pkg.ops.sam_t2i_q_operands(q, prompts, Tq, q_ops)
    out = torch.full((prompts * Tq, 128), float("nan"), device=cuda, dtype=bf)
    pkg.ops.sam_t2i_block(x, runs(wk), runs(wv), kpe_ops, bv, q_ops, prompts, Ni, Tq, out)
    k = (x.double() @ wk.double().t()).view(prompts, Ni, 8, 16) + kpe.to(bf).double().view(1, Ni, 8, 16)
    v = (x.double() @ wv.double().t() + bv.double()).view(prompts, Ni, 8, 16)
    qh = q.double().view(prompts, Tq, 8, 16)
    s = torch.einsum("bqhd,bnhd->bhqn", qh, k) * 0.25
    ref = torch.einsum("bhqn,bnhd->bqhd", torch.softmax(s, -1), v).reshape(prompts * Tq, 128)
    assert float((out.double() - ref).abs().max()) <= 3e-2                   # |v| ~ 1: roundings of k, p, v to bf16 and of the result
    if prompts > 1:                                                          # one stream for every prompt == that stream repeated
        out_a, out_b = torch.empty_like(out), torch.empty_like(out)
        pkg.ops.sam_t2i_block(x[:Ni].contiguous(), runs(wk), runs(wv), kpe_ops, bv, q_ops, prompts, Ni, Tq, out_a)
        pkg.ops.sam_t2i_block(x[:Ni].repeat(prompts, 1), runs(wk), runs(wv), kpe_ops, bv, q_ops, prompts, Ni, Tq, out_b)
        assert torch.equal(out_a, out_b)


def _tile_bf16(x):
    """Row-major [tokens, 256] -> the lane order of the fused mask-decoder kernels: entry (16 t + ks) 64 + 32 g + n = channels 16 ks + 8 g .. of token 32 t + n."""
    T = x.shape[0] // 32
    return x.view(T, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1, 256)


def _tile_f32(y):
    """Row-major [tokens, 256] f32 -> entry ((8 t + mb) 4 + i) 64 + 32 g + n = channels 32 mb + 8 i + 4 g .. of token 32 t + n."""
    T = y.shape[0] // 32
    return y.view(T, 32, 8, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().view(-1, 256)


def test_sam_fused_blocks_in_lane_order_equal_row_major(pkg, cuda):
    """The three fused kernels read and write the per-prompt image stream either row-major or in their own lane order:
    same values, bit for bit, only the addresses differ."""
    bf = torch.bfloat16
    prompts, Ni, Tk = 3, 128, 7
    rows = prompts * Ni
    slot = torch.tensor([16 * h + 8 * (e // 4) + 4 * g + e % 4 for h in range(8) for g in range(2) for e in range(8)], device=cuda)
    runs = lambda w: w.view(128, 32, 8).permute(1, 0, 2).contiguous()
    wq, wo = (_rand((128, 256), 71, cuda) * 0.08).to(bf), (_rand((256, 128), 72, cuda) * 0.1).to(bf)
    wo_runs = wo[:, slot].view(256, 16, 8).permute(1, 0, 2).contiguous()
    qbias, prm = _rand((Ni, 128), 73, cuda) * 0.5, torch.cat([_rand((256,), 74, cuda) * 0.2, _rand((256,), 75, cuda) * 0.3 + 1, _rand((256,), 76, cuda) * 0.2]).contiguous()
    kvo = torch.empty(prompts, 8, 64, 2, 8, device=cuda, dtype=bf)
    pkg.ops.sam_i2t_kv_operands(_rand((prompts * Tk, 128), 77, cuda).to(bf), _rand((prompts * Tk, 128), 78, cuda).to(bf), prompts, Tk, kvo)
    x, r = _rand((rows, 256), 79, cuda).to(bf), _rand((rows, 256), 80, cuda)
    # image -> token block: row-major in and out; then tiled out; then tiled in and out (in place, as the decoder runs it)
    y0, c0 = torch.empty(rows, 256, device=cuda), torch.empty(rows, 256, device=cuda, dtype=bf)
    pkg.ops.sam_i2t_block(x, runs(wq), wo_runs, qbias, prm, kvo, Tk, r, prompts, Ni, 1e-5, y0, c0)
    y1, c1 = torch.empty_like(y0), torch.empty_like(c0)
    pkg.ops.sam_i2t_block(x, runs(wq), wo_runs, qbias, prm, kvo, Tk, r, prompts, Ni, 1e-5, y1, c1, tiled=4)
    assert torch.equal(y1, _tile_f32(y0)) and torch.equal(c1, _tile_bf16(c0))
    xt, rt = _tile_bf16(x), _tile_f32(r)
    pkg.ops.sam_i2t_block(xt, runs(wq), wo_runs, qbias, prm, kvo, Tk, rt, prompts, Ni, 1e-5, rt, xt, tiled=7)
    assert torch.equal(rt, _tile_f32(y0)) and torch.equal(xt, _tile_bf16(c0))
    # token -> image block
    Tq = 7
    wk, wv, bv = (_rand((128, 256), 81, cuda) * 0.06).to(bf), (_rand((128, 256), 82, cuda) * 0.06).to(bf), _rand((128,), 83, cuda) * 0.3
    kpe_ops = pkg.ops.sam_t2i_kpe_operands(_rand((Ni, 128), 84, cuda))
    q_ops = torch.empty(prompts, 8, 64, 8, device=cuda, dtype=bf)
    pkg.ops.sam_t2i_q_operands((_rand((prompts * Tq, 128), 85, cuda) * 1.5).to(bf), prompts, Tq, q_ops)
    o0, o1 = (torch.empty(prompts * Tq, 128, device=cuda, dtype=bf) for _ in range(2))
    pkg.ops.sam_t2i_block(x, runs(wk), runs(wv), kpe_ops, bv, q_ops, prompts, Ni, Tq, o0)
    pkg.ops.sam_t2i_block(_tile_bf16(x), runs(wk), runs(wv), kpe_ops, bv, q_ops, prompts, Ni, Tq, o1, x_tiled=True)
    assert torch.equal(o0, o1)
    # first upscaling kernel (Ni = 128 is not a square grid: 2 prompts of an 8 x 8 grid)
    w0 = ((_rand((256, 256), 86, cuda) * 0.08).to(bf)).view(256, 32, 8).permute(1, 0, 2).contiguous()
    up_prm = torch.stack([_rand((64,), 87, cuda) * 0.2, _rand((64,), 88, cuda) * 0.3 + 1.0, _rand((64,), 89, cuda) * 0.2]).contiguous()
    xs = x[:2 * 64].contiguous()
    u0, u1 = (torch.empty(2 * 4 * 64, 64, device=cuda, dtype=bf) for _ in range(2))
    pkg.ops.sam_upscale_ln_gelu(xs, w0, up_prm, u0, 2, 8, 1e-6)
    pkg.ops.sam_upscale_ln_gelu(_tile_bf16(xs), w0, up_prm, u1, 2, 8, 1e-6, x_tiled=True)
    assert torch.equal(u0, u1)
