"""Host-side logic that needs no GPU: library symbols, resampling tables, box extraction,
synthetic data determinism, ranking arithmetic of the oracle."""
import importlib
import os

import numpy as np
import torch
import torch.nn.functional as F

from tests.helpers import load_package
from oracle import pipeline as opipe

pkg = load_package(require_native=False)


def test_library_exports_every_declared_symbol(pkg):
    names = pkg.hip.declared_symbols()
    assert len(names) >= 35
    assert pkg.hip.missing_symbols() == []
    assert pkg.hip.lib().marship_version() >= 1


def test_missing_library_fails_loudly(monkeypatch):
    hip = importlib.import_module(pkg.__name__ + ".hip")
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libmars_hip.so")
    try:
        hip.lib()
    except hip.MarsHipError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("a missing library must raise")


def test_ops_refuse_cpu_tensors(pkg):
    x = torch.zeros(4, 64)
    try:
        pkg.ops.layernorm(x, torch.ones(64), torch.zeros(64), torch.zeros(4, 64), 1e-5)
    except ValueError as e:
        assert "GPU" in str(e)
    else:
        raise AssertionError("CPU tensors must be rejected, not silently computed")


def _apply(img, tab, axis):
    xmin, xsize, w = tab
    img = np.moveaxis(img, axis, -1)
    out = np.zeros(img.shape[:-1] + (len(xmin),), np.float32)
    for o in range(len(xmin)):
        acc = img[..., xmin[o]] * w[o, 0]
        for k in range(1, xsize[o]):
            acc = (acc + img[..., xmin[o] + k] * w[o, k]).astype(np.float32)
        out[..., o] = acc
    return np.moveaxis(out, -1, axis)


def test_torch_antialias_tables_match_interpolate():
    rs = importlib.import_module(pkg.__name__ + ".resample")
    g = torch.Generator().manual_seed(0)
    for n_in, n_out, mode in [(126, 128, "bicubic"), (126, 56, "bilinear"), (70, 80, "bicubic"), (518, 336, "bilinear")]:
        x = torch.rand(1, n_in, n_in, generator=g)
        ref = F.interpolate(x[None], size=(n_out, n_out), mode=mode, align_corners=False, antialias=True)[0].numpy()
        tab = rs.torch_aa_table(n_in, n_out, mode)
        got = _apply(_apply(x.numpy(), tab, 2), tab, 1)
        assert np.abs(got - ref).max() < 2e-6


def test_pillow_tables_are_bit_exact():
    from PIL import Image
    rs = importlib.import_module(pkg.__name__ + ".resample")
    rng = np.random.default_rng(0)
    for n_in, n_out in [(126, 56), (200, 336), (518, 336)]:
        a = rng.integers(0, 256, (n_in, n_in, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(a, "RGB").resize((n_out, n_out), Image.BICUBIC))
        xmin, xsize, kk = rs.pil_table_u8(n_in, n_out)

        def run(img, axis):
            img = np.moveaxis(img, axis, -1).astype(np.int64)
            out = np.zeros(img.shape[:-1] + (n_out,), np.uint8)
            for o in range(n_out):
                ss = np.full(img.shape[:-1], 1 << 21, np.int64)
                for k in range(xsize[o]):
                    ss = ss + img[..., xmin[o] + k] * int(kk[o, k])
                out[..., o] = np.clip(ss >> 22, 0, 255)
            return np.moveaxis(out, -1, axis)

        assert np.array_equal(run(run(a, 1), 0), ref)


def test_oracle_boxes_match_scipy_label():
    from scipy import ndimage
    rng = np.random.default_rng(3)
    for trial in range(20):
        G = int(rng.integers(5, 38))
        prior = rng.random((G, G)).astype(np.float32)
        prior = ndimage.uniform_filter(prior, 3)
        prior = (prior - prior.min()) / (1e-7 + prior.max() - prior.min())
        thr = float(rng.choice([0.4, 0.8]))
        boxes = opipe.scoremap_boxes(prior, thr)
        img = (prior * 255).astype(np.uint8)
        binary = img > int(thr * img.max())
        lab, n = ndimage.label(binary, structure=np.ones((3, 3)))
        want = set()
        for sl in ndimage.find_objects(lab):
            ys, xs = sl
            want.add((xs.start, ys.start, min(xs.stop, G - 1), min(ys.stop, G - 1)))
        assert {tuple(int(v) for v in b) for b in boxes} == want


def test_oracle_no_component_gives_empty_box():
    assert opipe.scoremap_boxes(np.zeros((5, 5), np.float32), 0.8).tolist() == [[0, 0, 0, 0]]
    assert float(opipe.box_mask(torch.zeros(5, 5), 0.8).sum()) == 0.0


def test_synth_is_deterministic_and_wellformed():
    synth = pkg.synth
    a = synth.make_episode(5, size=126, nshot=2, nprop=8)
    b = synth.make_episode(5, size=126, nshot=2, nprop=8)
    for k in ("query_img", "support_imgs", "support_masks", "mask_proposals", "query_mask"):
        assert torch.equal(a[k], b[k])
    assert a["support_imgs"].shape == (1, 2, 3, 126, 126) and a["mask_proposals"].shape == (8, 126, 126)
    assert set(a["mask_proposals"].unique().tolist()) <= {0.0, 1.0}
    assert float(a["mask_proposals"].sum(dim=(1, 2)).min()) >= 14 * 14   # every proposal covers >= 1 patch
    sd1 = synth.dinov2_state_dict(synth.dino_cfg("vit_tiny2", 70), 1)
    sd2 = synth.dinov2_state_dict(synth.dino_cfg("vit_tiny2", 70), 1)
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)


def test_oracle_merge_thresholds():
    masks = torch.zeros(4, 4, 4)
    for i in range(4):
        masks[i, i, :] = 1
    score = np.array([0.9, 0.56, 0.54, 0.1]); order = np.argsort(-score, kind="stable")
    m, keep = opipe.fmm_merge(masks, score, order)
    assert keep == [0, 1] and float(m.sum()) == 8
    score = np.array([0.50, 0.48, 0.47, 0.1]); order = np.argsort(-score, kind="stable")
    m, keep = opipe.fmm_merge(masks, score, order)          # top < 0.55: keep >= 0.95*top = 0.475
    assert keep == [0, 1]


def test_oracle_compute_iou():
    inter = torch.zeros(2, 80); union = torch.zeros(2, 80)
    inter[1, 0], union[1, 0] = 50, 100
    inter[1, 4], union[1, 4] = 30, 40
    inter[0, 0], union[0, 0] = 900, 950
    inter[0, 4], union[0, 4] = 960, 970
    miou, fb = opipe.compute_iou(inter, union, [0, 4, 8])
    assert abs(miou - (0.5 + 0.75 + 0.0) / 3 * 100) < 1e-4
    assert abs(fb - ((900 + 960) / (950 + 970) + 80 / 140) / 2 * 100) < 1e-4


def test_pillow_bilinear_table_is_bit_exact():
    """The 8-bit BILINEAR coefficients reproduce PIL.Image.resize exactly (SAM's ResizeLongestSide)."""
    from PIL import Image
    resample = importlib.import_module(load_package(require_native=False).__name__ + ".resample")
    rng = np.random.default_rng(2)
    for (h, w), (nh, nw) in (((37, 53), (90, 128)), ((100, 126), (102, 128)), ((64, 40), (33, 21))):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        want = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))

        def axis_pass(a, n_out, axis):
            xmin, xsize, kk = resample.pil_table_u8(a.shape[axis], n_out, "bilinear")
            a = np.moveaxis(a.astype(np.int64), axis, -1)
            out = np.zeros(a.shape[:-1] + (n_out,), np.int64)
            for o in range(n_out):
                out[..., o] = (a[..., xmin[o]: xmin[o] + xsize[o]] * kk[o, : xsize[o]]).sum(-1)
            out = np.clip((out + (1 << 21)) >> 22, 0, 255)
            return np.moveaxis(out, -1, axis).astype(np.uint8)
        got = axis_pass(axis_pass(img, nw, 1), nh, 0)
        assert np.array_equal(got, want)


def test_bilinear_and_composite_tables_match_interpolate():
    resample = importlib.import_module(load_package(require_native=False).__name__ + ".resample")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 2, 32, 32, generator=g)
    M = resample.torch_bilinear_matrix(32, 75)
    want = F.interpolate(x, (75, 75), mode="bilinear", align_corners=False)
    got = torch.einsum("oi,bcij,pj->bcop", torch.from_numpy(M).float(), x, torch.from_numpy(M).float())
    assert float((got - want).abs().max()) < 1e-5
    # Sam.postprocess_masks on one axis pair: 32 -> 128, crop to (102, 128), -> (100, 126)
    up = F.interpolate(x, (128, 128), mode="bilinear", align_corners=False)[..., :102, :128]
    want = F.interpolate(up, (100, 126), mode="bilinear", align_corners=False)
    th = resample.sam_postprocess_table(32, 128, 102, 100)
    tw = resample.sam_postprocess_table(32, 128, 128, 126)

    def dense(t, n_in):
        xmin, xsize, w = t
        D = np.zeros((len(xmin), n_in), np.float32)
        for o in range(len(xmin)):
            D[o, xmin[o]: xmin[o] + xsize[o]] = w[o, : xsize[o]]
        return torch.from_numpy(D)
    got = torch.einsum("oi,bcij,pj->bcop", dense(th, 32), x, dense(tw, 32))
    assert float((got - want).abs().max()) < 1e-5
