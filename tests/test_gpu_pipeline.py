"""Whole ranking path on the GPU against the CPU oracle on the same seeded episode, and the
size-independent properties that must hold at any configuration."""
import importlib
import os

import numpy as np
import pytest
import torch

from tests.helpers import maxdiff, run_tiny_episode_parity, tiny_config
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _build(pkg, cfg):
    MARS = importlib.import_module(pkg.__name__ + ".mars.MARS")
    return MARS.build_MARS_fss(cfg["args"], state_dicts=cfg["sds"])


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_tiny_episode_fp32_parity(pkg, cuda, seed):
    rep = run_tiny_episode_parity(pkg, device="cuda:0", seed=seed, dtype=torch.float32, tol=TOL)
    assert rep["order_equal"], rep
    assert rep["mask_mismatch"] == 0, rep


def test_components_against_oracle(pkg, cuda):
    cfg = tiny_config(pkg)
    ep = pkg.synth.make_episode(7, size=cfg["size"], nshot=cfg["nshot"], nprop=cfg["nprop"])
    mars = _build(pkg, cfg)
    dev = cuda
    vva_mod = mars.visual_visual_alignment_component
    vva = vva_mod.compute(ep["support_imgs"].to(dev), ep["support_masks"].to(dev), ep["query_img"].to(dev))
    ref = opipe.vva_compute(cfg["oracle_models"]["dino_sd"], cfg["oracle_models"]["dino_kw"], ep["support_imgs"],
                            ep["support_masks"], ep["query_img"])
    assert maxdiff(vva_mod.similarity_matrix, ref["similarity"]) < TOL
    assert maxdiff(vva_mod.cost_matrix, ref["cost"]) < TOL
    assert maxdiff(vva_mod.vva_raw, ref["vva_raw"]) < TOL
    P = ref["mean_attn"].shape[0]
    assert maxdiff(vva_mod.mean_attn[:, :P], ref["mean_attn"]) < TOL
    assert maxdiff(vva, ref["vva"]) < TOL
    vta_mod = mars.visual_text_alignment_component
    vta = vta_mod.compute(ep["query_img"].to(dev), "object", [])
    rt = opipe.vta_compute(cfg["oracle_models"]["clip_sd"], cfg["oracle_models"]["clip_kw"], ep["query_img"],
                           cfg["tokens"]["fg"], cfg["tokens"]["bg"])
    assert maxdiff(vta_mod.logits, rt["logits"]) < TOL
    assert maxdiff(vta_mod.cam, rt["cam"]) < 5e-4
    Pv = rt["mean_attn"].shape[0]
    assert maxdiff(vta_mod.mean_attn[:, :Pv], rt["mean_attn"]) < TOL
    assert maxdiff(vta, rt["vta"]) < 5e-4 * max(1.0, float(rt["vta"].abs().max()))


def test_five_shot_episode(pkg, cuda):
    cfg = tiny_config(pkg, nshot=5, nprop=8)
    ep = pkg.synth.make_episode(21, size=cfg["size"], nshot=5, nprop=8)
    mars = _build(pkg, cfg)
    pred = mars.predict(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda), ep["query_img"].to(cuda),
                        ep["mask_proposals"].to(cuda))
    ref = opipe.mars_predict(cfg["oracle_models"], ep, cfg["tokens"])
    got = mars.filtering_merging_component.scores()
    assert maxdiff(got["emd"], ref["scores"]["emd"]) < TOL
    assert maxdiff(got["score"], ref["scores"]["score"]) < TOL
    assert got["order"].tolist() == ref["scores"]["order"].tolist()
    assert int((pred.cpu() != ref["mask"]).sum()) == 0


def test_bf16_run_ranks_like_fp32(pkg, cuda):
    """bf16 cannot meet 1e-4; it is judged on rank agreement and mask overlap with the fp32 run."""
    rep32 = run_tiny_episode_parity(pkg, device="cuda:0", seed=3, dtype=torch.float32)
    rep16 = run_tiny_episode_parity(pkg, device="cuda:0", seed=3, dtype=torch.bfloat16)
    print("bf16 vs oracle:", rep16, "fp32 vs oracle:", rep32)
    assert all(np.isfinite(v) for k, v in rep16.items() if isinstance(v, float)), rep16
    assert rep16["emd"] < 0.05 and rep16["score"] < 0.25, rep16


def test_properties_merge_and_iou(pkg, cuda):
    """Idempotence and order independence: merging the merged mask with itself changes nothing, the
    kept set does not depend on proposal order, IoU counts add up to the pixel count."""
    cfg = tiny_config(pkg)
    ep = pkg.synth.make_episode(9, size=cfg["size"], nshot=1, nprop=6)
    mars = _build(pkg, cfg)
    args = [ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda), ep["query_img"].to(cuda)]
    pred = mars.predict(*args, ep["mask_proposals"].to(cuda)).clone()
    perm = torch.tensor([3, 0, 5, 1, 4, 2])
    pred_p = mars.predict(*args, ep["mask_proposals"][perm].to(cuda)).clone()
    assert torch.equal(pred, pred_p)
    keep = torch.ones(2, dtype=torch.uint8, device=cuda)
    again = torch.empty_like(pred)
    pkg.ops.merge_masks(torch.stack([pred, pred]).contiguous(), keep, again)
    assert torch.equal(again, pred)
    ev = importlib.import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
    ev.initialize()
    ai, au = ev.classify_prediction(pred[None], {"query_mask": ep["query_mask"].to(cuda)})
    n = pred.numel()
    assert float(au.sum() + ai.sum()) == float(2 * n)       # |pred_k| + |gt_k| summed over both bins
    meter = importlib.import_module(pkg.__name__ + ".mars.utils.logger").AverageMeter
    ids = sorted({0, 4, int(ep["class_id"])})
    ds = type("DS", (), {"benchmark": "coco", "class_ids": ids})()
    am = meter(ds, device=cuda)
    am.update(ai, au, ep["class_id"].to(cuda))
    miou, fb, _ = am.compute_iou()
    wi, wu = opipe.classify_prediction(pred.cpu(), ep["query_mask"][0])
    inter = torch.zeros(2, 80); union = torch.zeros(2, 80)
    inter[:, int(ep["class_id"])] = wi; union[:, int(ep["class_id"])] = wu
    m_ref, fb_ref = opipe.compute_iou(inter, union, ids)
    assert abs(float(miou) - m_ref) < 1e-4 and abs(float(fb) - fb_ref) < 1e-4


def test_reference_signature_with_explicit_cost_matrix(pkg, cuda):
    """FilteringMergingModule.compute called the reference way: cost_matrix as a CPU tensor."""
    cfg = tiny_config(pkg)
    ep = pkg.synth.make_episode(3, size=cfg["size"], nshot=1, nprop=6)
    mars = _build(pkg, cfg)
    pred = mars.predict(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda), ep["query_img"].to(cuda),
                        ep["mask_proposals"].to(cuda)).clone()
    s1 = mars.filtering_merging_component.scores()["score"].clone()
    vva_mod = mars.visual_visual_alignment_component
    out = mars.filtering_merging_component.compute(
        query_img=ep["query_img"].to(cuda), mask_proposals=ep["mask_proposals"].to(cuda), support_mask=ep["support_masks"].to(cuda),
        cost_matrix=vva_mod.cost_matrix, patch_features_spatial_dimension=vva_mod.model_embedding_spatial_dimensions,
        vva=mars.vva, vta=mars.vta, text=["a object."])
    assert torch.equal(out, pred)
    assert maxdiff(mars.filtering_merging_component.scores()["score"], s1) < 1e-12


def test_order_of_work_does_not_change_the_result(pkg, cuda):
    """The reference's order of work (everything on one stream, stage after stage), the proposal batch and the
    transport solver started right behind the visual-visual tower, and the default - the three stages side by
    side on streams of their own - give the same prediction and the same scores, bit for bit."""
    cfg = tiny_config(pkg)
    ep = pkg.synth.make_episode(9, size=cfg["size"], nshot=cfg["nshot"], nprop=cfg["nprop"])
    args = [ep[k].to(cuda) for k in ("support_imgs", "support_masks", "query_img", "mask_proposals")]
    out = {}
    for mode in ("reference order", "early batch", "concurrent"):
        mars = _build(pkg, cfg)
        mars.concurrent = mode == "concurrent"
        mars.early_batch = mode != "reference order"
        for _ in range(3):                      # later calls reuse the streams and the buffers of the first
            pred = mars.predict(*args)
        out[mode] = (pred.clone(), {k: v.clone() for k, v in mars.filtering_merging_component.scores().items()},
                     mars.vva.clone(), mars.vta.clone())
    for mode in ("early batch", "concurrent"):
        assert torch.equal(out["reference order"][0], out[mode][0]), mode
        for k in ("emd", "aclip", "score", "order"):
            assert torch.equal(out["reference order"][1][k], out[mode][1][k]), (mode, k)
        assert torch.equal(out["reference order"][2], out[mode][2]) and torch.equal(out["reference order"][3], out[mode][3]), mode


two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")


@two_gpus
def test_operands_of_another_gpu_are_refused(pkg, cuda):
    """Kernels go to the current device's stream: an operand that lives on another GPU is an error, not a launch."""
    torch.cuda.set_device(0)
    a = torch.zeros(64, 64, device="cuda:1")
    with pytest.raises(ValueError, match="current device"):
        pkg.ops.linear(a, a, torch.zeros(64, 64, device="cuda:1"))
    with torch.cuda.device(1):
        out = torch.full((64, 64), float("nan"), device="cuda:1")
        pkg.ops.linear(torch.eye(64, device="cuda:1"), torch.eye(64, device="cuda:1"), out)
        assert torch.equal(out, torch.eye(64, device="cuda:1"))


@two_gpus
def test_episode_on_a_gpu_that_is_not_current(pkg, cuda):
    """``args.device = cuda:1`` while cuda:0 is current: the builders and ``predict`` select the device themselves."""
    torch.cuda.set_device(0)
    rep = run_tiny_episode_parity(pkg, device="cuda:1", seed=3, dtype=torch.float32, tol=TOL)
    assert rep["order_equal"] and rep["mask_mismatch"] == 0, rep
    assert torch.cuda.current_device() == 0


_NOCACHE_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from tests.helpers import tiny_config
from __graft_entry__ import load_package
from importlib import import_module
pkg = load_package(require_native=True)
MARS = import_module(pkg.__name__ + ".mars.MARS")
cfg = tiny_config(pkg)
dev = torch.device("cuda:0")
mars = MARS.build_MARS_fss(cfg["args"], state_dicts=cfg["sds"])
mars.concurrent = %(concurrent)r
out = {}
for e in (21, 22, 21):                        # two different episodes back to back, then the first again
    ep = pkg.synth.make_episode(e, size=cfg["size"], nshot=cfg["nshot"], nprop=cfg["nprop"])
    args = [ep[k].to(dev) for k in ("support_imgs", "support_masks", "query_img", "mask_proposals")]
    pred = mars.predict(*args)
    sc = mars.filtering_merging_component.scores()
    out.setdefault(e, []).append(dict(pred=pred.cpu(), score=sc["score"], order=sc["order"], emd=sc["emd"], aclip=sc["aclip"],
                                      vva=mars.vva.cpu(), vta=mars.vta.cpu()))
    mars.clear()
torch.save(out, %(path)r)
"""


def test_streams_without_the_caching_allocator(tmp_path):
    """Side by side on four streams with the caching allocator OFF (every tensor that dies is freed at once and the
    next allocation may get its address): two different episodes back to back, then the first again, give bit for
    bit what the one-stream order gives, and the repeated episode gives what it gave the first time.  A result
    read on a stream other than the one it was made on must therefore be kept alive by the hand-over itself."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in (True, False):
        path = str(tmp_path / f"out_{int(mode)}.pt")
        env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1", PYTORCH_NO_HIP_MEMORY_CACHING="1")
        r = subprocess.run([sys.executable, "-c", _NOCACHE_SCRIPT % dict(root=root, concurrent=mode, path=path)], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        got[mode] = torch.load(path)
    for e in (21, 22):
        for i, (a, b) in enumerate(zip(got[True][e], got[False][e])):
            for k in a:
                assert torch.equal(a[k], b[k]), (e, i, k)
    first, again = got[True][21]
    for k in first:
        assert torch.equal(first[k], again[k]), k
