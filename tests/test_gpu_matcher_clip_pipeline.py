"""Matcher scoring pieces and CLIPPipeline.process_masks on the GPU against their CPU restatement
(oracle/matcher.py; parity unpinned: cv2 / POT / torchvision are absent, see the oracle header)."""
import importlib

import numpy as np
import pytest
import torch

from tests.helpers import maxdiff, tiny_config
from oracle import matcher as omatch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _points(ep, n=40, seed=0):
    rng = np.random.default_rng(seed)
    S = ep["query_img"].shape[-1]
    ys, xs = np.nonzero(ep["query_mask"][0].numpy() > 0) if "query_mask" in ep else (np.array([S // 2]), np.array([S // 2]))
    pick = rng.integers(0, len(ys), size=n // 2)
    inside = np.stack([xs[pick], ys[pick]], axis=1).astype(np.float64) + rng.random((n // 2, 2))
    anywhere = rng.random((n - n // 2, 2)) * (S + 6) - 3          # a few outside the image: they are clipped
    return np.concatenate([inside, anywhere], axis=0)


@pytest.mark.parametrize("nshot", [1, 2])
def test_matcher_features_similarity_and_mask_scores(pkg, cuda, nshot):
    cfg = tiny_config(pkg, nshot=nshot, nprop=7)
    cfg["args"].device = "cuda:0"
    ep = pkg.synth.make_episode(31 + nshot, size=cfg["size"], nshot=nshot, nprop=7)
    M = importlib.import_module(pkg.__name__ + ".matcher.Matcher")
    cfg["args"].alpha, cfg["args"].beta, cfg["args"].exp = 0.8, 0.3, 0.5
    m = M.build_matcher_oss(cfg["args"], state_dict=cfg["sds"]["dino"])
    m.set_reference(ep["support_imgs"].to(cuda), ep["support_masks"].to(cuda))
    m.set_target(ep["query_img"].to(cuda))
    G = cfg["size"] // 14
    om = cfg["oracle_models"]
    pool_ref = omatch.set_reference_pool(ep["support_masks"], 14)
    assert torch.equal(m.ref_masks_pool.cpu(), pool_ref)
    ref_f, tar_f = m.extract_img_feats()
    o_ref, o_tar = omatch.extract_img_feats(om["dino_sd"], om["dino_kw"], ep["support_imgs"][0], ep["query_img"])
    assert maxdiff(ref_f, o_ref) < TOL and maxdiff(tar_f, o_tar) < TOL
    S, C = m.similarity()
    oS, oC = omatch.similarity_and_cost(o_ref, o_tar)
    assert maxdiff(S, oS) < TOL and maxdiff(C, oC) < TOL
    masks = ep["mask_proposals"].clone()
    masks[3] = 0                                                   # an empty proposal: selects every patch
    pts = _points(ep)
    got = m.score_masks(masks[:, None].to(cuda), pts)
    want = omatch.matcher_scores(masks[:, None].numpy(), pts, oC, pool_ref, G, 0.8, 0.3, 0.5)
    for k in ("purity", "coverage", "emd", "scores"):
        assert maxdiff(got[k], want[k]) < TOL, k
    assert torch.equal(m.get_unfiltered_generated_masks().cpu(), (masks > 0).float())
    with pytest.raises(NotImplementedError):
        m.predict()
    # an all-zero reference mask gets the centred square (lines 150-153)
    m.clear()
    m.set_reference(ep["support_imgs"].to(cuda), torch.zeros_like(ep["support_masks"]).to(cuda))
    assert torch.equal(m.ref_masks_pool.cpu(), omatch.set_reference_pool(torch.zeros_like(ep["support_masks"]), 14))


def test_points_in_masks_and_cosine_table_kernels(pkg, cuda):
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(5, 37, 41, generator=g) > 0.6).float().to(cuda)
    pts = (torch.rand(300, 2, generator=g) * torch.tensor([47.0, 43.0]) - 3.0)
    cnt = torch.zeros(5, dtype=torch.int32, device=cuda)
    pkg.ops.points_in_masks(masks, pts.to(cuda).contiguous(), cnt)
    x = pts[:, 0].long().clamp(0, 40)
    y = pts[:, 1].long().clamp(0, 36)
    want = torch.stack([(masks[k].cpu()[y, x] != 0).sum() for k in range(5)]).int()
    assert torch.equal(cnt.cpu(), want)
    f = torch.randn(9, 48, generator=g).to(cuda)
    t = torch.randn(4, 48, generator=g).to(cuda)
    out = torch.zeros(9, 4, device=cuda)
    pkg.ops.cosine_table(f, t, out, scale=100.0)
    ref = 100.0 * torch.nn.functional.normalize(f.double(), dim=1) @ torch.nn.functional.normalize(t.double(), dim=1).t()
    assert float((out.double() - ref).abs().max()) < 1e-4


class _FixedGenerator:
    def __init__(self, masks):
        self.masks = masks
        self.resets = 0

    def generate(self, image_np):
        assert image_np.ndim == 3 and image_np.shape[2] == 3
        return [{"segmentation": m} for m in self.masks]

    def reset_stored_features(self):
        self.resets += 1


def test_clip_pipeline_process_masks(pkg, cuda):
    cfg = tiny_config(pkg, nprop=5)
    ep = pkg.synth.make_episode(41, size=cfg["size"], nshot=1, nprop=5)
    loader = importlib.import_module(pkg.__name__ + ".utils.backbone_loader")
    model, transforms = loader.BackboneLoader.load_backbone(
        backbone_name="alphaclip", backbone_size="tiny/14", device="cuda:0", backbone_weights_path=None,
        encoder_kwargs={"device": "cuda:0", "state_dict": cfg["sds"]["alphaclip"], "dtype": torch.float32})
    CP = importlib.import_module(pkg.__name__ + ".mars.ClipPipeline")
    seg = [m.numpy() > 0 for m in ep["mask_proposals"][:4]]
    gen = _FixedGenerator(seg)
    pipe = CP.CLIPPipeline(model, transforms, generator=gen, input_size=cfg["size"], device="cuda:0")
    pipe.set_reference(ep["support_imgs"], ep["support_masks"])
    pipe.set_target(ep["query_img"])
    pipe.set_candidate_mask([ep["mask_proposals"][4].numpy() > 0])
    prompts = ["a object.", "a photo of object.", "a photo without object."]
    masks, sims = pipe.process_masks(prompts)
    assert len(masks) == 5 and masks[-1]["type"] == "matcher" and sorted(sims) == [0, 1, 2, 3, 4]
    om = cfg["oracle_models"]
    want = omatch.clip_pipeline_process_masks(om["aclip_sd"], om["aclip_kw"], ep["query_img"][0],
                                              seg + [ep["mask_proposals"][4].numpy() > 0], pkg.synth.tokenize(prompts))
    for i in range(5):
        assert sims[i].shape == (1, 3)
        assert maxdiff(sims[i], want[i]) < 100 * TOL, i          # the table is 100 x cosine
    with pytest.raises(ValueError):
        pipe.predict()                                            # no class name and no VLM
    pipe.reset_clip_pipeline()
    assert gen.resets >= 1 and pipe.target_img is None
