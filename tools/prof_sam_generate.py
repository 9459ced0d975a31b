"""Three warm ProposalGenerator.generate() calls at ViT-H (for a kernel trace of the proposal stage alone)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package(require_native=True)
sa = pkg.segment_anything
sam = sa.sam_model_registry["vit_h"](device="cuda", dtype=torch.bfloat16, seed=0)
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
amg = sa.SamAutomaticMaskGenerator(sam, pred_iou_thresh=-10.0, stability_score_thresh=0.0)
q = torch.as_tensor(img).permute(2, 0, 1)[None].float().div(255).cuda()
gen = sa.ProposalGenerator(amg, 100)
for _ in range(int(os.environ.get("CALLS", "4"))):
    gen.generate(None, None, q)
torch.cuda.synchronize()
