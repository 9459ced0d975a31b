"""Times the Segment-Anything proposal stage at a real configuration on one GPU (synthetic weights):
image embedding, one decoder batch of 64 points with post-processing and statistics, and a whole
32x32-point generate()."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="vit_h")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--size", type=int, nargs=2, default=(480, 640))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-generate", action="store_true")
    a = ap.parse_args()
    pkg = load_package(require_native=True)
    sa = pkg.segment_anything
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    t0 = time.perf_counter()
    sam = sa.sam_model_registry[a.arch](device="cuda", dtype=dtype, seed=0)
    print(f"build {a.arch} {a.dtype}: {time.perf_counter() - t0:.1f} s", flush=True)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(a.size[0], a.size[1], 3), dtype=np.uint8)
    pred = sa.SamPredictor(sam)
    print(f"set_image: {timed(lambda: pred.set_image(img), a.reps):.2f} ms", flush=True)
    enc = sam.image_encoder
    if getattr(enc, "fused_attention", False) and enc.hdp % 64 == 0:
        enc.fused_attention = False
        print(f"set_image with scores / softmax / P.V as three launches: {timed(lambda: pred.set_image(img), a.reps):.2f} ms "
              f"(workspace {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB)", flush=True)
        enc.fused_attention = True
        pred.set_image(img)
    pts = torch.rand(64, 1, 2, device="cuda") * 1000
    lab = torch.ones(64, 1, dtype=torch.int32, device="cuda")
    print(f"predict_torch (64 points, 3 masks each, logits at {a.size}): "
          f"{timed(lambda: pred.predict_torch(pts, lab, return_logits=True), a.reps):.2f} ms", flush=True)
    if not a.no_generate:
        amg = sa.SamAutomaticMaskGenerator(sam, pred_iou_thresh=-10.0, stability_score_thresh=0.0)
        t0 = time.perf_counter()
        recs = amg.generate(img)
        torch.cuda.synchronize()
        print(f"generate (32x32 points), first call (buffers allocated, tables built): {(time.perf_counter() - t0) * 1e3:.1f} ms, {len(recs)} records", flush=True)
        print(f"generate (32x32 points), warm: {timed(lambda: amg.generate(img), 2):.1f} ms", flush=True)
        q = torch.as_tensor(img).permute(2, 0, 1)[None].float().div(255).cuda()
        gen = sa.ProposalGenerator(amg, 100)
        print(f"proposals [100, H, W]: {timed(lambda: gen.generate(None, None, q), 2):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
