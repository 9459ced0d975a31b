"""Visual-text alignment (Softmax-GradCAM on CLIP + attention refinement) on the GPU.

Interface of /root/reference/mars/components/VisualTextAlignmentModule.py:23-104 and of the
``SoftmaxGradCAM`` helper it drives (/root/reference/mars/components/SoftmaxGradCAM.py:43-175):
``compute(query_image, fg_label, bg_labels) -> [G, G]`` (unscaled refined CAM).

The reference runs the last CLIP block under autograd with hooks; here the gradient of
softmax logit 0 with respect to the ``ln_1`` output is an explicit chain of kernels
(``CLIP.gradcam``), and the attention maps of the last n blocks are averaged on the fly.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from ... import ops, resample
from .PriorInformationRefinementModule import PriorInformationRefinementModule

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class VisualTextAlignmentModule:
    def __init__(self, model, model_transforms, model_patch_size: int, model_embedding_spatial_dimensions,
                 model_num_regs: int, vta_refinement_box_threshold: float, last_n_attention_maps_for_refinement: int,
                 device, tokenizer: Optional[Callable] = None):
        self.model = model
        self.model_transforms = model_transforms  # API parity; resize + normalise run as kernels
        self.model_patch_size = model_patch_size
        self.model_embedding_spatial_dimensions = model_embedding_spatial_dimensions
        self.model_num_regs = model_num_regs
        self.device = torch.device(device)
        self.last_n = last_n_attention_maps_for_refinement
        self.pir = PriorInformationRefinementModule(
            box_threshold=vta_refinement_box_threshold,
            last_n_attention_maps_for_refinement=last_n_attention_maps_for_refinement, device=device, num_regs=model_num_regs)
        if tokenizer is None:
            from ...tokenizer import default_tokenizer
            tokenizer = default_tokenizer()   # merge-table BPE when MARS_BPE_PATH is set, else the bundled prompt ids
        self.tokenizer = tokenizer
        self._mean = torch.tensor(CLIP_MEAN, dtype=torch.float32, device=self.device)
        self._std = torch.tensor(CLIP_STD, dtype=torch.float32, device=self.device)
        self._tables: Dict = {}
        self.cam = None
        self.logits = None
        self.mean_attn = None

    def _table(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self._tables:
            xmin, xsize, wts = resample.torch_aa_table(n_in, n_out, "bicubic")
            self._tables[key] = (torch.from_numpy(xmin).to(self.device), torch.from_numpy(xsize).to(self.device),
                                 torch.from_numpy(wts).to(self.device).contiguous())
        return self._tables[key]

    def preprocess(self, image: torch.Tensor) -> torch.Tensor:
        """Resize((S', S'), BICUBIC) of lines 88-92 as two antialiased passes (width, then height);
        the CLIP normalisation is applied later inside the patch extraction."""
        C, H, W = image.shape
        p = self.model_patch_size
        size = int(math.ceil(H / p) * p)
        tmp = torch.empty(C, H, size, dtype=torch.float32, device=self.device)
        out = torch.empty(C, size, size, dtype=torch.float32, device=self.device)
        img = image.contiguous().float()
        ops.resample_axis(img, C, H, W, 1, self._table(W, size), tmp)
        ops.resample_axis(tmp, C, H, size, 0, self._table(H, size), out)
        return out

    def compute_text_feats(self, fg_label: str) -> torch.Tensor:
        """SoftmaxGradCAM.compute_text_feats (lines 63-109) for one label and one prompt per side:
        rows [fg, bg], each normalised, averaged over its single prompt, normalised again."""
        toks = self.tokenizer([f"a photo of {fg_label}.", f"a photo without {fg_label}."])
        e = self.model.encode_text(toks)                                   # [2, E]
        n, E = e.shape
        a = torch.empty(n, E, dtype=torch.float32, device=self.device)
        b = torch.empty(n, E, dtype=torch.float32, device=self.device)
        ops.l2norm_rows(e.view(1, n, E), 0, n, a)
        ops.l2norm_rows(a.view(1, n, E), 0, n, b)
        return b

    def prepare_text(self, fg_label: str, stream: "torch.cuda.Stream"):
        """Queues the prompt features of ``fg_label`` on ``stream`` (a chain of small launches that needs nothing
        from the image tower and can run beside it); ``compute`` for the same label waits for them there."""
        with torch.cuda.stream(stream):
            text = self.compute_text_feats(fg_label)
            done = torch.cuda.Event()
            done.record(stream)
        self._prepared = (fg_label, text, done)

    def compute(self, query_image: torch.Tensor, fg_label: str, bg_labels: List[str]) -> torch.Tensor:
        assert len(bg_labels) == 0, "the ranking path always calls with bg_labels=[] (mars/MARS.py:72)"
        image = self.preprocess(query_image[0].to(self.device))
        T = (image.shape[-1] // self.model_patch_size) ** 2 + 1
        acc = self.pir.new_accumulator(T - 1)
        X, _ = self.model.encode_image_tokens(image[None], self._mean, self._std, mean_attn=(acc, self.last_n))
        prepared, self._prepared = getattr(self, "_prepared", None), None        # taken once, never reused
        if prepared is not None and prepared[0] == fg_label:
            text = prepared[1]
            here = torch.cuda.current_stream(self.device)
            here.wait_event(prepared[2])
            text.record_stream(here)        # made on the prompt stream, read here: its block is not reused before this stream got there
        else:
            text = self.compute_text_feats(fg_label)
        self.text_features = text           # kept alive until the next call (it may live on another stream's pool)
        L, H = self.model.cfg.vision_layers, self.model.cfg.vision_heads
        last = (acc, 1, T - 1, 1.0 / (min(self.last_n, L) * H)) if self.last_n >= 1 else None
        cam, logits, _ = self.model.gradcam(X, text, mean_attn=last)
        self.cam, self.logits, self.mean_attn = cam, logits, acc
        return self.pir.refine(cam, acc)

    def clear(self):
        pass


def build_visual_text_alignment_component(args, state_dict=None):
    """Reference builder (lines 79-104) with weights injectable through ``state_dict``."""
    from ...utils.backbone_loader import BackboneLoader
    model, _ = BackboneLoader.load_backbone(
        backbone_name="clip", backbone_size=args.vta_backbone,
        encoder_kwargs={"device": args.device, "download_root": getattr(args, "models_path", None), "state_dict": state_dict,
                        "dtype": getattr(args, "dtype", torch.float32)})
    patch = model.visual.patch_size
    size = int(np.ceil(int(args.input_size) / patch) * patch)
    return VisualTextAlignmentModule(
        model=model, model_transforms=None, model_patch_size=patch, model_embedding_spatial_dimensions=size / patch,
        model_num_regs=0, vta_refinement_box_threshold=args.vta_refinement_box_threshold,
        last_n_attention_maps_for_refinement=args.last_n_attn_for_vta_refinement, device=args.device)
