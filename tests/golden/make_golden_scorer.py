"""Golden vectors for the scorer, produced by the REFERENCE's own lines.

Run in the build container only (it reads /root/reference):

    python tests/golden/make_golden_scorer.py

What is executed from the reference, unmodified, loaded from its files:
  * mars/components/FilteringMergingModule.py   (_score_proposals :59-140, _merge_masks :209-221, compute :35-57)
  * mars/components/VisualVisualAlignmentModule.py (compute :42-111, _extract_patch_features :113-127)
  * mars/components/PriorInformationRefinementModule.py (compute :20-89)
  * mars/MARS.py (predict :33-104)
  * dinov2.models.vision_transformer (the tower VisualVisualAlignmentModule drives)

The modules import wheels that are absent here (torchvision, cv2, POT ``ot``, alpha_clip's
tokenizer dependencies).  They get NAME-ONLY entries in ``sys.modules`` - no function of
theirs is written here, and no code path that would dereference one is driven:
  * ``_compute_emd`` (the ``ot.emd2`` call) and the two ``_compute_alphaclip_*`` methods are
    overridden in a subclass that returns preset numbers, so _score_proposals' own numpy
    arithmetic (dtype promotions, the 1e-7 terms, ``sorted``) is what is recorded;
  * ``_scoremap2bbox`` (the cv2 calls) is overridden to return boxes given by hand for priors
    built from known rectangles, so the attention slicing, the Sinkhorn step, R = max(D, DD^T),
    R.R and the exclusive [y0:y1, x0:x1] fill are what is recorded;
  * VisualVisualAlignmentModule gets a plain-tensor normalisation callable as its transform and
    a pass-through object in place of ``self.pir``.
cv2 / POT / torchvision arithmetic itself stays "parity unpinned".

numpy is unpinned in the reference's requirements; the vectors record what its lines give under
the numpy of this container (2.2: the AlphaCLIP min-max stays float32, the mean of four is float64).
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from __graft_entry__ import load_package  # noqa: E402

pkg = load_package(require_native=False)
synth = pkg.synth
from oracle import pipeline as opipe  # noqa: E402

IMAGENET_MEAN = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
IMAGENET_STD = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """The reference's scorer modules with name-only entries for the absent wheels."""
    _stub("ot")
    _stub("cv2", __version__="4.10.0")
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Compose=object)
    _stub("alpha_clip", tokenize=None)
    ut = _stub("utils")
    ut.backbone_loader = _stub("utils.backbone_loader", BackboneLoader=object)
    # the VLM text retriever and the CLIP GradCAM component are not driven here (the first is out of the
    # path, the second is pinned by make_golden.py); MARS.py only needs their names to import
    _stub("mars.components.TextRetrieverModule", build_text_retriever_component=None, TextRetrieverModule=object)
    _stub("mars.components.VisualTextAlignmentModule", build_visual_text_alignment_component=None, VisualTextAlignmentModule=object)
    pir = importlib.import_module("mars.components.PriorInformationRefinementModule")
    vva = importlib.import_module("mars.components.VisualVisualAlignmentModule")
    fmm = importlib.import_module("mars.components.FilteringMergingModule")
    mars = importlib.import_module("mars.MARS")
    for m in (pir, vva, fmm, mars):
        assert m.__file__.startswith(REF), m.__file__
    return pir, vva, fmm, mars


def rect_masks(seed, name, n, size, dup=()):
    """n binary masks of random rectangles; ``dup`` = pairs (i, j): mask i is a copy of mask j."""
    g = torch.Generator().manual_seed(synth.stable_seed(seed, name))
    out = torch.zeros(n, size, size)
    for i in range(n):
        h, w = [int(x) for x in torch.randint(max(2, size // 8), size // 2, (2,), generator=g)]
        y, x = int(torch.randint(0, size - h, (1,), generator=g)), int(torch.randint(0, size - w, (1,), generator=g))
        out[i, y:y + h, x:x + w] = 1
    for i, j in dup:
        out[i] = out[j]
    return out


# --------------------------------------------------------------------------- _merge_masks
def golden_merge(fmm_mod, out):
    cases = [
        # (scores, static, dynamic)
        ([0.81, 0.40, 0.56, 0.55, 0.5499999, 0.10], 0.55, 0.95),      # static branch, one score equal to the threshold
        ([0.50, 0.48, 0.475, 0.4749, 0.2, 0.47], 0.55, 0.95),        # dynamic branch, 0.475 == 0.95 * 0.5
        ([0.3, 0.3, 0.3], 0.55, 0.95),                               # all tied below the static threshold
        ([0.55], 0.55, 0.95),                                        # one proposal exactly at the threshold
        ([0.2, 0.9, 0.9, 0.1], 0.9, 0.5),                            # top == static threshold -> static branch
    ]
    for k, (scores, st, dy) in enumerate(cases):
        masks = rect_masks(21, f"merge{k}", len(scores), 24)
        mod = fmm_mod.FilteringMergingModule(None, None, None, alpha=0.85, static_threshold=st, dynamic_threshold=dy, device="cpu")
        ranked = sorted([(m, s) for m, s in zip(masks, scores)], key=lambda x: x[1], reverse=True)   # as line 138
        merged = mod._merge_masks(ranked)
        sc = np.asarray(scores, dtype=np.float64)
        order = np.asarray(sorted(range(len(scores)), key=lambda i: scores[i], reverse=True))
        ours, _ = opipe.fmm_merge(masks, sc, order, static_threshold=st, dynamic_threshold=dy)
        assert torch.equal(ours, merged), k
        out[f"merge{k}_masks"] = masks.numpy().astype(np.uint8)
        out[f"merge{k}_scores"] = sc
        out[f"merge{k}_thresholds"] = np.asarray([st, dy])
        out[f"merge{k}_merged"] = merged.numpy().astype(np.uint8)
    out["merge_cases"] = np.asarray(len(cases))
    print(f"[merge] {len(cases)} cases, oracle fmm_merge identical")


# --------------------------------------------------------------------------- _score_proposals
def make_preset_fmm(fmm_mod, emd, feats, text, **kw):
    class PresetFMM(fmm_mod.FilteringMergingModule):
        """The reference class with its three model / solver hooks returning preset values."""

        def __init__(self):
            super().__init__(None, None, None, device="cpu", **kw)
            self.calls = 0
            self.seen = {}

        def _compute_emd(self, support_mask, mask_proposal, cost_matrix):
            # same slicing as line 160, so that a wrongly shaped argument still fails here
            pool = cost_matrix[support_mask.flatten().bool(), :][:, mask_proposal.flatten().bool()]
            assert pool.shape[0] > 0 and pool.shape[1] > 0
            v = emd[self.calls]
            self.calls += 1
            return v

        def _compute_alphaclip_text_feats(self, text_list):
            self.seen["text"] = list(text_list)
            return text

        def _compute_alphaclip_vis_feats(self, image, masks):
            return feats

        def _score_proposals(self, **kwargs):
            self.seen.update(vva=kwargs["vva"].clone(), vta=kwargs["vta"].clone(), cost=kwargs["cost_matrix"])
            self.ranked = super()._score_proposals(**kwargs)
            return self.ranked
    return PresetFMM()


def score_inputs(case, G, N, ns, dup=(), E=16):
    S = G * 14
    masks = rect_masks(31, f"score{case}", N, S, dup)
    g = torch.Generator().manual_seed(synth.stable_seed(31, f"score{case}_rest"))
    support = torch.zeros(1, ns, S, S)
    for s in range(ns):
        support[0, s, 10 + 5 * s: S // 2 + 9 * s, 14: S - 20] = 1
    P = G * G
    cost = torch.rand(ns * P, P, generator=g)
    vva = torch.rand(G, G, generator=g)
    vta = torch.rand(G, G, generator=g)
    emd = [np.float64(x) for x in (0.5 + 0.4 * torch.rand(N, generator=g, dtype=torch.float64)).numpy()]   # 1 - emd2(...): numpy float64
    feats = torch.nn.functional.normalize(torch.randn(N, E, generator=g), dim=1)
    text = torch.nn.functional.normalize(torch.randn(1, E, generator=g), dim=1)
    for i, j in dup:
        emd[i] = emd[j]
        feats[i] = feats[j]
    return dict(masks=masks, support=support, cost=cost, vva=vva, vta=vta, emd=emd, feats=feats, text=text,
                query=torch.rand(1, 3, S, S, generator=g))


def ranked_to_arrays(ranked, masks):
    step = masks[0].numel() * masks.element_size()
    order = np.asarray([(m.data_ptr() - masks.data_ptr()) // step for m, _ in ranked], dtype=np.int64)
    score = np.zeros(len(ranked), dtype=np.float64)
    for (m, s), i in zip(ranked, order):
        score[i] = np.asarray(s, dtype=np.float64).reshape(-1)[0]
    return order, score


def golden_score(fmm_mod, out):
    cases = [
        # name, G, N, ns, duplicates, static, dynamic
        ("a", 5, 7, 1, (), 0.55, 0.95),
        ("b", 9, 12, 2, ((5, 2), (9, 2)), 0.55, 0.95),     # three identical proposals: ties keep input order
        ("c", 9, 12, 2, ((5, 2), (9, 2)), 0.95, 0.90),     # same inputs, top below the static threshold
        ("d", 5, 1, 1, (), 0.55, 0.95),                    # one proposal: both min-max terms are 0 / 1e-7
    ]
    for name, G, N, ns, dup, st, dy in cases:
        seedname = "b" if name == "c" else name
        x = score_inputs(seedname, G, N, ns, dup)
        mod = make_preset_fmm(fmm_mod, x["emd"], x["feats"], x["text"], alpha=0.85, static_threshold=st, dynamic_threshold=dy)
        merged = mod.compute(query_img=x["query"], mask_proposals=x["masks"], support_mask=x["support"], cost_matrix=x["cost"],
                             patch_features_spatial_dimension=G, vva=x["vva"], vta=x["vta"], text=["a object."])
        order, score = ranked_to_arrays(mod.ranked, x["masks"])
        sdt = {type(s).__name__ + ":" + str(getattr(s, "dtype", "")) for _, s in mod.ranked}
        print(f"[score {name}] G={G} N={N} ns={ns} top={score[order[0]]:.6f} kept pixels={int(merged.sum())} score type {sdt}")
        # the oracle's combination on the same presets
        o = preset_oracle_scores(x, G)
        d = np.abs(o["score"] - score).max()
        assert d < 1e-6 and (o["order"] == order).all(), (name, d)
        om, _ = opipe.fmm_merge(x["masks"], o["score"], o["order"], static_threshold=st, dynamic_threshold=dy)
        assert torch.equal(om, merged), name
        for k in ("masks", "support"):
            out[f"score_{name}_{k}"] = x[k].numpy().astype(np.uint8)
        for k in ("cost", "vva", "vta", "feats", "text"):
            out[f"score_{name}_{k}"] = x[k].numpy()
        out[f"score_{name}_emd"] = np.asarray(x["emd"], dtype=np.float64)
        out[f"score_{name}_thresholds"] = np.asarray([0.85, st, dy])
        out[f"score_{name}_order"] = order
        out[f"score_{name}_score"] = score
        out[f"score_{name}_merged"] = merged.numpy().astype(np.uint8)
    out["score_cases"] = np.asarray([c[0] for c in cases])


def preset_oracle_scores(x, G, alpha=0.85):
    return opipe.combine_scores(x["masks"], G, x["vva"], x["vta"], np.asarray(x["emd"]),
                                (x["feats"] @ x["text"].t())[:, 0].numpy(), alpha=alpha)


# --------------------------------------------------------------------------- PriorInformationRefinementModule.compute
def blob_prior(G, rects, seed, name):
    """Background below 0.2, the given (y0, y1, x0, x1) inclusive rectangles filled with 0.85..1.0."""
    g = torch.Generator().manual_seed(synth.stable_seed(seed, name))
    p = 0.2 * torch.rand(G, G, generator=g)
    for (y0, y1, x0, x1) in rects:
        p[y0:y1 + 1, x0:x1 + 1] = 0.85 + 0.15 * torch.rand(y1 - y0 + 1, x1 - x0 + 1, generator=g)
    p[rects[0][0], rects[0][2]] = 1.0 if rects else p[0, 0]
    return p


def hand_boxes(G, rects):
    """What lines 115-120 give for a rectangle x..x+w-1, y..y+h-1: [x, y, min(x+w, W-1), min(y+h, H-1)]."""
    return np.asarray([[x0, y0, min(x1 + 1, G - 1), min(y1 + 1, G - 1)] for (y0, y1, x0, x1) in rects])


def attn_list(seed, name, L, heads, T, four_d):
    g = torch.Generator().manual_seed(synth.stable_seed(seed, name))
    maps = []
    for _ in range(L):
        a = torch.softmax(2.0 * torch.randn(1, heads, T, T, generator=g), dim=-1)
        maps.append(a if four_d else a.mean(dim=1))     # [1,h,T,T] (DINOv2) or head-averaged [1,T,T] (CLIP)
    return maps


def golden_pir(pir_mod, out):
    cases = [
        # name, G, regs, L, last_n, heads, 4-D maps, threshold, rectangles (y0, y1, x0, x1)
        ("vva", 9, 4, 5, 3, 2, True, 0.8, [(1, 2, 1, 3), (5, 7, 6, 8)]),        # second blob touches the right edge: x1 clipped to W-1
        ("vta", 7, 0, 4, 8, 3, False, 0.4, [(0, 1, 0, 1), (4, 6, 2, 3), (6, 6, 6, 6)]),   # last_n > L; corner pixel -> empty box
        ("zero", 5, 0, 2, 2, 1, False, 0.4, []),                               # no contour: box [0,0,0,0], count 1
    ]
    for name, G, regs, L, last_n, heads, four_d, thr, rects in cases:
        prior = blob_prior(G, rects, 41, f"pir_{name}") if rects else torch.zeros(G, G)
        boxes = hand_boxes(G, rects) if rects else np.asarray([[0, 0, 0, 0]])

        class HandBoxes(pir_mod.PriorInformationRefinementModule):
            def _scoremap2bbox(self, scoremap, multi_contour_eval=False):
                return boxes, len(boxes)

        T = 1 + regs + G * G
        maps = attn_list(41, f"pir_attn_{name}", L, heads, T, four_d)
        mod = HandBoxes(box_threshold=thr, last_n_attention_maps_for_refinement=last_n, device="cpu", num_regs=regs)
        refined = mod.compute(prior=prior.clone(), attn_maps=maps)
        # the oracle finds its own boxes: they must be the hand-given ones, and the chain must agree
        ob = opipe.scoremap_boxes(prior.numpy(), thr)
        assert sorted(map(tuple, ob.tolist())) == sorted(map(tuple, boxes.tolist())), (name, ob, boxes)
        A = opipe.mean_attention(maps, last_n, regs)
        ours = opipe.pir_refine(prior, A, thr)
        d = float((ours - refined).abs().max())
        print(f"[pir {name}] G={G} boxes={boxes.tolist()} max|ref-oracle| = {d:.2e} (|ref|max {float(refined.abs().max()):.3e})")
        assert d <= 1e-6 * max(1.0, float(refined.abs().max())), (name, d)
        out[f"pir_{name}_prior"] = prior.numpy()
        out[f"pir_{name}_attn"] = torch.stack(maps).numpy()
        out[f"pir_{name}_boxes"] = boxes.astype(np.int64)
        out[f"pir_{name}_params"] = np.asarray([G, regs, L, last_n, heads, int(four_d)], dtype=np.int64)
        out[f"pir_{name}_threshold"] = np.asarray(thr)
        out[f"pir_{name}_refined"] = refined.numpy()
    out["pir_cases"] = np.asarray([c[0] for c in cases])


# --------------------------------------------------------------------------- VisualVisualAlignmentModule.compute
class PassThroughPIR:
    """Stands where ``self.pir`` is: hands the prior back and keeps what it was given."""

    def compute(self, prior, attn_maps):
        self.prior = prior.clone()
        self.attn = [a.detach() for a in attn_maps]
        return prior


def reference_dino(arch, img):
    import dinov2.models.vision_transformer as vits
    cfg = synth.dino_cfg(arch, img_size=img)
    sd = synth.dinov2_state_dict(cfg, seed=1)
    model = vits.DinoVisionTransformer(
        img_size=img, patch_size=14, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4,
        init_values=1e-5, ffn_layer="mlp", block_chunks=0, num_register_tokens=cfg.num_regs,
        block_fn=vits.partial(vits.Block, attn_class=vits.MemEffAttention))
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model, cfg, sd


def golden_vva(vva_mod, out):
    img = 84
    model, cfg, sd = reference_dino("vit_tiny3", img)
    G = img // 14
    normalise = lambda t: (t - IMAGENET_MEAN) / IMAGENET_STD            # noqa: E731  plain-tensor transform
    cases = [("fgbg", 1, "part"), ("allfg", 1, "full"), ("two", 2, "part")]
    for name, ns, kind in cases:
        mod = vva_mod.VisualVisualAlignmentModule(
            model=model, model_transforms=normalise, model_patch_size=14, model_embedding_spatial_dimensions=G,
            model_num_regs=cfg.num_regs, vva_refinement_box_threshold=0.8, last_n_attention_maps_for_refinement=cfg.depth, device="cpu")
        mod.pir = PassThroughPIR()
        sup = synth.rand(51, f"vva_sup_{name}", (1, ns, 3, img, img))
        qry = synth.rand(51, f"vva_qry_{name}", (1, 3, img, img))
        masks = torch.zeros(1, ns, img, img)
        if kind == "full":
            masks[:] = 1                                                 # every support patch foreground: lines 100-101
        else:
            for s in range(ns):
                masks[0, s, 15 + 20 * s: 50 + 10 * s, 20: 66] = 1
        with torch.no_grad():
            res = mod.compute(support_imgs=sup, support_masks=masks, query_img=qry)
        o = opipe.vva_compute(sd, dict(depth=cfg.depth, heads=cfg.num_heads, patch=14, num_regs=cfg.num_regs, ffn="mlp"),
                              sup, masks, qry, last_n=cfg.depth)
        d_s = float((o["similarity"] - mod.similarity_matrix).abs().max())
        d_p = float((o["vva_raw"] - mod.pir.prior).abs().max())
        A = opipe.mean_attention(mod.pir.attn, cfg.depth, cfg.num_regs)
        d_a = float((o["mean_attn"] - A).abs().max())
        print(f"[vva {name}] ns={ns}: similarity {d_s:.2e}, prior handed to the refinement {d_p:.2e}, mean attention {d_a:.2e}")
        assert d_s < 1e-5 and d_p < 2e-5 and d_a < 1e-6, name
        out[f"vva_{name}_support"] = sup.numpy()
        out[f"vva_{name}_query"] = qry.numpy()
        out[f"vva_{name}_masks"] = masks.numpy().astype(np.uint8)
        out[f"vva_{name}_similarity"] = mod.similarity_matrix.numpy()
        out[f"vva_{name}_cost"] = mod.cost_matrix.numpy()
        out[f"vva_{name}_prior"] = mod.pir.prior.numpy()               # what line 105 hands to the refinement
        out[f"vva_{name}_mean_attn"] = A.numpy()
        out[f"vva_{name}_out"] = res.numpy()                           # line 109 on the pass-through
    out["vva_cases"] = np.asarray([c[0] for c in cases])
    out["vva_arch"] = np.asarray(["vit_tiny3", str(img)])


# --------------------------------------------------------------------------- MARS.predict
def golden_mars(mars_mod, fmm_mod, out):
    class Retriever:
        def __init__(self, name, desc):
            self.pair = (name, desc)

        def get_conceptual_information(self, support_images, support_masks):
            return self.pair

    class PresetVTA:
        def __init__(self, vta):
            self.vta = vta

        def compute(self, query_image, fg_label, bg_labels):
            self.seen = (fg_label, list(bg_labels))
            return self.vta

    class PresetVVA:
        def __init__(self, vva, cost, G):
            self.vva, self.cost_matrix, self.model_embedding_spatial_dimensions = vva, cost, G

        def compute(self, support_imgs, support_masks, query_img):
            return self.vva

        def clear(self):
            pass

    cases = [("up", 9, 7, "object", ""), ("down", 9, 11, "dog", "a four legged animal")]
    for name, G, Gt, cname, desc in cases:
        x = score_inputs(f"mars_{name}", G, 8, 1)
        g = torch.Generator().manual_seed(synth.stable_seed(61, f"mars_{name}"))
        vta_raw = 3.0 * torch.rand(Gt, Gt, generator=g) - 0.5          # the text prior arrives unscaled
        fmm = make_preset_fmm(fmm_mod, x["emd"], x["feats"], x["text"], alpha=0.85, static_threshold=0.55, dynamic_threshold=0.95)
        vta_c = PresetVTA(vta_raw)
        m = mars_mod.MARS(Retriever(cname, desc), vta_c, PresetVVA(x["vva"], x["cost"], G), fmm)
        pred = m.predict(x["support"][:, :, None].repeat(1, 1, 3, 1, 1), x["support"], x["query"], x["masks"])
        order, score = ranked_to_arrays(fmm.ranked, x["masks"])
        assert vta_c.seen == (cname, []) and fmm.seen["cost"] is x["cost"]
        # oracle glue (mars_predict lines after the two alignment calls)
        ovta = opipe.minmax(torch.nn.functional.interpolate(vta_raw[None, None], (G, G), mode="nearest")[0, 0])
        assert torch.equal(ovta, fmm.seen["vta"]), name
        print(f"[mars {name}] text {fmm.seen['text']} vta {Gt}->{G} kept pixels {int(pred.sum())}")
        for k in ("masks", "support"):
            out[f"mars_{name}_{k}"] = x[k].numpy().astype(np.uint8)
        for k in ("cost", "vva", "feats", "text", "query"):
            out[f"mars_{name}_{k}"] = x[k].numpy()
        out[f"mars_{name}_emd"] = np.asarray(x["emd"], dtype=np.float64)
        out[f"mars_{name}_vta_raw"] = vta_raw.numpy()
        out[f"mars_{name}_vta_scaled"] = fmm.seen["vta"].numpy()
        out[f"mars_{name}_prompt"] = np.asarray(fmm.seen["text"])
        out[f"mars_{name}_label"] = np.asarray([cname, desc])
        out[f"mars_{name}_order"] = order
        out[f"mars_{name}_score"] = score
        out[f"mars_{name}_pred"] = pred.numpy().astype(np.uint8)
    out["mars_cases"] = np.asarray([c[0] for c in cases])


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    pir_mod, vva_mod, fmm_mod, mars_mod = load_reference()
    out = {}
    golden_merge(fmm_mod, out)
    golden_score(fmm_mod, out)
    golden_pir(pir_mod, out)
    golden_vva(vva_mod, out)
    golden_mars(mars_mod, fmm_mod, out)
    path = os.path.join(HERE, "scorer_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.1f} KiB, numpy {np.__version__}")
