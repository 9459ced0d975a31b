"""Test loop of MARS on stored (or generated) mask proposals - the reference's entry script
(/root/reference/main_MARS.py:15-171) on the MI355X path: episodes -> MARS.predict -> Evaluator -> mIoU, with
the per-episode ranking times written to ``ranking_time.csv``.

    python main_MARS.py --benchmark coco --fold 0 --nshot 1 --dataset_path D --mask_proposals_path P --models_path M
    python main_MARS.py --synthetic 4                      # no files: seeded synthetic episodes and weights

Proposals come from ``{mask_proposals_path}/{fold}_{idx}.pt`` (or ``.mprop``, the bit-packed form), or are
generated in-process with ``--generate_proposals matcher|sam``.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time
from importlib import import_module

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from __graft_entry__ import load_package  # noqa: E402


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MARS few-shot segmentation test loop (MI355X path)")
    p.add_argument("--dataset_path", type=str, default=None)
    p.add_argument("--mask_proposals_path", type=str, default=None)
    p.add_argument("--benchmark", type=str, default="coco", choices=["fss", "coco", "pascal5i", "lvis"])
    p.add_argument("--nworker", type=int, default=0)
    p.add_argument("--nshot", type=int, default=1)
    p.add_argument("--fold", type=int, default=0)
    p.add_argument("--input_size", type=int, default=518)
    p.add_argument("--models_path", type=str, default=None)
    p.add_argument("--vta_backbone", type=str, default="ViT-B/16")
    p.add_argument("--vta_refinement_box_threshold", type=float, default=0.4)
    p.add_argument("--last_n_attn_for_vta_refinement", type=int, default=8)
    p.add_argument("--vva_backbone", type=str, default="dino")
    p.add_argument("--dino_backbone", type=str, default="vit_large")
    p.add_argument("--alphaclip_backbone", type=str, default="ViT-L/14@336px")
    p.add_argument("--num_regs", type=int, default=4)
    p.add_argument("--vva_refinement_box_threshold", type=float, default=0.8)
    p.add_argument("--last_n_attn_for_vva_refinement", type=int, default=24)
    p.add_argument("--static_threshold", type=float, default=0.55)
    p.add_argument("--dynamic_threshold", type=float, default=0.95)
    p.add_argument("--alpha_coverage", type=float, default=0.85)
    p.add_argument("--log_root_path", type=str, default="logs")
    p.add_argument("--exp_name", type=str, default="mars")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--fp8_attention", action="store_true",
                   help="8-bit float (e4m3) operands in the attention of the AlphaCLIP proposal tower (bf16 towers only)")
    # additions of this implementation
    p.add_argument("--class_name", type=str, default="object", help="text retriever stand-in: the class name used in the prompts")
    p.add_argument("--class_description", type=str, default="")
    p.add_argument("--synthetic", type=int, default=0, help="run this many seeded synthetic episodes with synthetic weights")
    p.add_argument("--proposals", type=int, default=64, help="proposals per synthetic episode")
    p.add_argument("--max_episodes", type=int, default=0)
    p.add_argument("--generate_proposals", type=str, default=None, choices=["matcher", "sam"])
    p.add_argument("--sam_size", type=str, default="vit_h")
    p.add_argument("--sam_weights", type=str, default=None)
    return p


def test_MARS(args) -> dict:
    pkg = load_package(require_native=True)
    MARS = import_module(pkg.__name__ + ".mars.MARS")
    Evaluator = import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
    AverageMeter = import_module(pkg.__name__ + ".mars.utils.logger").AverageMeter
    utils = import_module(pkg.__name__ + ".mars.utils.utils")
    data = import_module(pkg.__name__ + ".data")
    pio = import_module(pkg.__name__ + ".io")
    log_path = os.path.join(args.log_root_path, args.exp_name)
    os.makedirs(log_path, exist_ok=True)
    random.seed(0)
    utils.fix_randseed(0)
    args.dtype = torch.bfloat16 if args.dtype in ("bf16", torch.bfloat16) else torch.float32
    if torch.device(args.device).type == "cuda":
        torch.cuda.set_device(args.device)       # kernels launch on the current device: make it the one asked for

    state_dicts = None
    if args.synthetic:
        synth = pkg.synth
        if args.input_size == 518:
            dcfg, ccfg, acfg = synth.dino_cfg(args.dino_backbone, img_size=518), synth.CLIP_ARCHS[args.vta_backbone], synth.CLIP_ARCHS[args.alphaclip_backbone]
        else:                                            # small runs (tests): the tiny towers
            args.dino_backbone, args.vta_backbone, args.alphaclip_backbone = "vit_tiny2", "tiny/16", "tiny/14"
            dcfg, ccfg, acfg = synth.dino_cfg("vit_tiny2", img_size=args.input_size), synth.CLIP_ARCHS["tiny/16"], synth.CLIP_ARCHS["tiny/14"]
        state_dicts = dict(dino=synth.dinov2_state_dict(dcfg, 11), clip=synth.clip_state_dict(ccfg, 11),
                           alphaclip=synth.clip_state_dict(acfg, 11, alpha=True))
        dataset = data.DatasetSynthetic(args.synthetic, args.input_size, args.nshot, args.proposals, args.fold)
        loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0)
    else:
        data.FSSDataset.initialize(img_size=args.input_size, datapath=args.dataset_path, use_original_imgsize=False)
        loader = data.FSSDataset.build_dataloader(benchmark=args.benchmark, bsz=1, nworker=args.nworker, fold=args.fold,
                                                  split="test", shot=args.nshot)
    Evaluator.initialize()
    meter = AverageMeter(loader.dataset, device=args.device)
    mars = MARS.build_MARS_fss(args, state_dicts=state_dicts)
    if args.generate_proposals == "sam":
        sa = pkg.segment_anything
        sam = sa.sam_model_registry[args.sam_size](checkpoint=args.sam_weights, device=args.device, dtype=args.dtype)
        mars.mask_generator = sa.ProposalGenerator(sa.SamAutomaticMaskGenerator(sam), n_proposals=args.proposals)
    elif args.generate_proposals == "matcher":
        M = import_module(pkg.__name__ + ".matcher.Matcher")
        args.sam_dtype = args.dtype
        mars.mask_generator = M.MatcherProposals(M.build_matcher_oss(args, state_dict=(state_dicts or {}).get("dino")))

    t_start = time.time()
    per_batch, rows = [], []
    for idx, batch in enumerate(loader):
        if args.max_episodes and idx >= args.max_episodes:
            break
        t_batch = time.time()
        batch = utils.to_cuda(batch, args.device)
        proposals = None
        if mars.mask_generator is None:
            if "mask_proposals" in batch:
                proposals = batch["mask_proposals"][0]
            else:
                stem = os.path.join(args.mask_proposals_path, f"{args.fold}_{idx}")
                path = stem + (".mprop" if os.path.exists(stem + ".mprop") else ".pt")
                proposals = pio.load_proposals(path, args.device)
        pred = mars.predict(support_images=batch["support_imgs"], support_masks=batch["support_masks"],
                            query_image=batch["query_img"], mask_proposals=proposals).unsqueeze(0).to(args.device)
        inter, union = Evaluator.classify_prediction(pred.clone(), batch)
        meter.update(inter, union, batch["class_id"], loss=None)
        n_props = proposals.shape[0] if proposals is not None else -1
        rows.append((idx, mars.time_end_ranking - mars.time_start_ranking,
                     mars.time_end_ranking - mars.time_start_ranking_after_text_extraction, n_props))
        mars.clear()
        per_batch.append(time.time() - t_batch)
    with open(os.path.join(log_path, "ranking_time.csv"), "w") as f:
        f.write("idx,full_mars_prediction_time,ranking_time_after_text_extraction,number_of_proposals\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]:.4f},{r[2]:.4f},{r[3]}\n")
    miou, fb_iou, _ = meter.compute_iou()
    out = dict(miou=float(miou), fb_iou=float(fb_iou), episodes=len(rows), avg_time_per_image=utils.mean(per_batch),
               test_time=time.time() - t_start, log_path=log_path)
    print(f"mIoU: {out['miou']:.2f} - FB-IoU: {out['fb_iou']:.2f}")
    print(f"Average time per image: {out['avg_time_per_image']:.2f}")
    print(f"Test time: {out['test_time']:.2f}")
    return out


if __name__ == "__main__":
    test_MARS(build_parser().parse_args())
