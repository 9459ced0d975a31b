"""Per-proposal scoring, ranking and merging on the GPU (+ exact EMD on host threads).

Interface of /root/reference/mars/components/FilteringMergingModule.py:15-221:
``compute(query_img, mask_proposals, support_mask, cost_matrix, patch_features_spatial_dimension,
vva, vta, text) -> [H, W]``.

Data flow per query: proposals are read once by the pooling kernel (patch occupancy bits) and
once by the alpha-plane resampler; the N AlphaCLIP passes run as one batch while the host
solves the N transport problems; a single-block kernel combines the four scores in float64,
ranks them and decides the kept set; the merge kernel ORs the kept proposals.

Order of work: everything that needs neither prior (pooling, the AlphaCLIP batch, the transport
problems) can be started with ``begin`` as soon as the similarity matrix is on its way - ``MARS.predict``
does so right behind the visual-visual tower - so that the long GPU-bound batch is already queued while the
host issues the many small launches of the text tower and of the prompt encoders, and the transport solver
(a host thread of its own) has the whole batch to hide behind.  ``compute`` picks the started work up, or
starts it itself when nobody did.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Callable, Dict, List, Optional, Union

import torch

from ... import ops, resample
from .VisualVisualAlignmentModule import SimilarityHandle
from .VisualTextAlignmentModule import CLIP_MEAN, CLIP_STD


def default_emd_threads() -> int:
    """Host threads for the transport solver of one process: the cores this process may run on (at most
    16: a query has 64 problems and the solver stops scaling there), shared evenly between the ranks that
    a launcher placed on this node (LOCAL_WORLD_SIZE, else WORLD_SIZE) - unless the rank was already given a
    core range of its own (MARS_RANK_PINNED=1: the mask is this rank's share)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    ranks = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    if os.environ.get("MARS_RANK_PINNED", "0") == "1":
        ranks = 1
    return max(1, min(16, cores // max(ranks, 1)))


class FilteringMergingModule:
    def __init__(self, alpha_clip_model, img_transforms, mask_transforms, alpha: float, static_threshold: float,
                 dynamic_threshold: float, device, tokenizer: Optional[Callable] = None, emd_threads: int = 0):
        self.alpha_clip_model = alpha_clip_model
        self.img_transforms = img_transforms      # API parity; both transforms run as kernels
        self.mask_transforms = mask_transforms
        self.alpha = alpha
        self.static_threshold = static_threshold
        self.dynamic_threshold = dynamic_threshold
        self.device = torch.device(device)
        self.alpha_clip_batch_size = 128
        if tokenizer is None:
            from ...tokenizer import default_tokenizer
            tokenizer = default_tokenizer()   # merge-table BPE when MARS_BPE_PATH is set, else the bundled prompt ids
        self.tokenizer = tokenizer
        self.emd_threads = emd_threads or int(os.environ.get("MARS_EMD_THREADS", "0")) or default_emd_threads()
        self._mean = torch.tensor(CLIP_MEAN, dtype=torch.float32, device=self.device)
        self._std = torch.tensor(CLIP_STD, dtype=torch.float32, device=self.device)
        self._tables: Dict = {}
        self._ws: Dict = {}
        self.last = {}
        self.timing: dict = {}
        self._begun: Optional[dict] = None

    # ---------------------------------------------------------------- tables / buffers
    def _aa_table(self, n_in, n_out):
        key = ("aa", n_in, n_out)
        if key not in self._tables:
            t = resample.torch_aa_table(n_in, n_out, "bilinear")
            self._tables[key] = tuple(torch.from_numpy(a).to(self.device).contiguous() for a in t)
        return self._tables[key]

    def _pil_table(self, n_in, n_out):
        key = ("pil", n_in, n_out)
        if key not in self._tables:
            t = resample.pil_table_u8(n_in, n_out)
            self._tables[key] = tuple(torch.from_numpy(a).to(self.device).contiguous() for a in t)
        return self._tables[key]

    def _buffers(self, N: int, S: int, P: int, R: int):
        key = (N, S, P, R)
        w = self._ws.get(key)
        if w is None:
            dev = self.device
            res = self.alpha_clip_model.input_resolution
            f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            w = self._ws[key] = dict(
                bits=torch.zeros(N, P, dtype=torch.uint8, device=dev),
                bits_host=torch.zeros(N, P, dtype=torch.uint8).pin_memory(),
                fg_host=torch.zeros(R, dtype=torch.uint8).pin_memory(),
                sbits=torch.zeros(R, dtype=torch.uint8, device=dev),
                cnt=torch.zeros(N, dtype=torch.int32, device=dev), ucnt=torch.zeros(1, dtype=torch.int32, device=dev),
                svv=f(N), svt=f(N), aclip=f(N),
                emd=torch.zeros(N, dtype=torch.float64, device=dev),
                emd_host=torch.zeros(N, dtype=torch.float64).pin_memory(),
                score=torch.zeros(N, dtype=torch.float64, device=dev),
                order=torch.zeros(N, dtype=torch.int32, device=dev),
                keep=torch.zeros(N, dtype=torch.uint8, device=dev),
                img_u8=torch.zeros(3, S, S, dtype=torch.uint8, device=dev),
                img_tmp=torch.zeros(3, S, res, dtype=torch.uint8, device=dev),
                img_res=torch.zeros(1, 3, res, res, dtype=torch.uint8, device=dev),
                a_tmp=f(min(N, self.alpha_clip_batch_size), S, res),
                alpha=f(min(N, self.alpha_clip_batch_size), 1, res, res),
                feats=f(N, self.alpha_clip_model.cfg.embed_dim),
                merged=f(S, S),
            )
        return w

    # ---------------------------------------------------------------- reference API
    def compute(self, query_img, mask_proposals, support_mask, cost_matrix, patch_features_spatial_dimension: int,
                vva, vta, text: List[str]) -> torch.Tensor:
        self._score_proposals(query_img, mask_proposals, support_mask, cost_matrix, patch_features_spatial_dimension,
                              vva, vta, text)
        return self._merge_masks(mask_proposals)

    def begin(self, query_img, mask_proposals, support_mask, cost_matrix, patch_features_spatial_dimension: int, on_block=None):
        """Starts the part of ``compute`` that needs neither prior, on the current stream: patch occupancy of
        the support masks and of the proposals with their copies to the host, the AlphaCLIP batch over the
        proposals, and - on a host thread of its own - the transport problems, which start as soon as the
        occupancy bits and the similarity matrix have reached the host.  ``compute`` with the same proposals
        takes the result; anything else discards it.  ``on_block(i)`` is called when block i of the first
        AlphaCLIP chunk has been queued (``MARS.predict`` issues the visual-visual tower there)."""
        dev = self.device
        if self._begun is not None:             # started and never taken: its solver still reads the pinned buffers
            if self._begun["job"]["thread"] is not None:
                self._begun["job"]["thread"].join()
            self._begun = None
        masks = mask_proposals.to(dev).contiguous().float()
        N, S, _ = masks.shape
        P = patch_features_spatial_dimension * patch_features_spatial_dimension
        smask = support_mask.to(dev)[0].contiguous().float()       # [ns, S, S]
        R = smask.shape[0] * P
        w = self._buffers(N, S, P, R)
        # lines 73-81, 104-110: patch occupancy of the support masks and of every proposal
        ops.pool_masks(smask, patch_features_spatial_dimension, w["sbits"])
        ops.pool_masks(masks, patch_features_spatial_dimension, w["bits"])
        w["bits_host"].copy_(w["bits"], non_blocking=True)
        w["fg_host"].copy_(w["sbits"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        job = dict(result=None, error=None, thread=None, ready=ev, w=w, t_begin=time.perf_counter())
        if cost_matrix is not None:
            self._start_solver(job, cost_matrix)
        # lines 183-207: the AlphaCLIP visual features of every proposal
        hook = {"on_block": on_block} if on_block is not None else {}      # (subclasses may override without the hook)
        feats = self._compute_alphaclip_vis_feats(query_img[0].to(dev), masks, w, **hook)
        self._begun = dict(key=self._input_key(query_img, mask_proposals, support_mask, patch_features_spatial_dimension),
                           w=w, masks=masks, feats=feats, job=job, N=N, P=P)
        return self._begun

    @staticmethod
    def _input_key(query_img, mask_proposals, support_mask, G: int):
        """What ``compute`` must be given to take over the work ``begin`` started: the very same three tensor
        objects (held here, so their ids cannot be reused), unmodified since (their version counters), and the
        same grid.  Another tensor at the same address, or the same one written to in between, starts over."""
        return tuple((t, t._version) for t in (query_img, mask_proposals, support_mask)) + (G,)

    @staticmethod
    def _same_inputs(a, b) -> bool:
        return len(a) == len(b) and a[-1] == b[-1] and all(x[0] is y[0] and x[1] == y[1] for x, y in zip(a[:-1], b[:-1]))

    def start_solver(self, cost_matrix):
        """The transport problems of the work ``begin`` started without a similarity matrix (``cost_matrix=None``)."""
        if self._begun is not None and self._begun["job"]["thread"] is None:
            self._start_solver(self._begun["job"], cost_matrix)

    def _start_solver(self, job: dict, cost_matrix):
        """Lines 112-116, 142-169: the N transport problems, on a host thread of its own (the solver's worker threads
        hang below it) that first waits for the occupancy bits and the similarity matrix to reach the host."""
        if not isinstance(cost_matrix, SimilarityHandle):
            cost_matrix = cost_matrix.detach().cpu().float().contiguous()
        w = job["w"]

        def solve():
            try:
                job["ready"].synchronize()
                t0 = time.perf_counter()
                job["result"] = self._compute_emd_scores(cost_matrix, w["fg_host"], w["bits_host"])
                job["solve_s"] = time.perf_counter() - t0       # inputs on the host -> all problems solved
                job["end_s"] = time.perf_counter() - job["t_begin"]     # since the proposal stage was started
            except BaseException as exc:        # handed to the caller of compute()
                job["error"] = exc

        job["thread"] = threading.Thread(target=solve, name="mars-emd", daemon=True)
        job["thread"].start()

    def prepare_text(self, text: List[str], stream: "torch.cuda.Stream"):
        """Queues the AlphaCLIP prompt features on ``stream``; ``compute`` with the same prompt waits for them there."""
        with torch.cuda.stream(stream):
            tfeat = self._compute_alphaclip_text_feats(text)
            done = torch.cuda.Event()
            done.record(stream)
        self._prepared_text = (tuple(text), tfeat, done)

    def _compute_alphaclip_text_feats(self, text: List[str]) -> torch.Tensor:
        """Lines 171-181 (normalisation happens inside the cosine-score kernel)."""
        return self.alpha_clip_model.encode_text(self.tokenizer(text))

    def _image_u8(self, image: torch.Tensor, w: dict) -> torch.Tensor:
        """Image branch of the AlphaCLIP transform (utils/backbone_loader.py:172-180): float image ->
        8-bit (x255, truncated) -> Pillow bicubic resize -> 8-bit [1,3,R,R]; /255 and the CLIP
        normalisation are fused into the patch extraction."""
        C, S, S2 = image.shape
        assert S == S2, "the synthetic and benchmark episodes are square; centre crop is then the identity"
        res = self.alpha_clip_model.input_resolution
        ops.f32_to_u8(image.contiguous().float(), w["img_u8"])
        t = self._pil_table(S, res)
        ops.resample_axis_u8(w["img_u8"], C, S, S, 1, t, w["img_tmp"])
        ops.resample_axis_u8(w["img_tmp"], C, S, res, 0, t, w["img_res"])
        return w["img_res"]

    def _compute_alphaclip_vis_feats(self, image: torch.Tensor, masks: torch.Tensor, w: Optional[dict] = None, on_block=None) -> torch.Tensor:
        """Lines 183-207: one AlphaCLIP pass per proposal (batched), un-normalised features [N, E]."""
        N, S, _ = masks.shape
        if w is None:
            w = self._buffers(N, S, 1, 1)
        res = self.alpha_clip_model.input_resolution
        img = self._image_u8(image, w)
        t = self._aa_table(S, res)
        bs = self.alpha_clip_batch_size
        for i in range(0, N, bs):
            n = min(bs, N - i)
            chunk = masks[i:i + n].contiguous().float()
            # mask branch (backbone_loader.py:182-186): u8 quantise, bilinear antialiased resize, (x-0.5)/0.26
            ops.resample_axis(chunk, n, S, S, 1, t, w["a_tmp"], quant_u8=True)
            ops.resample_axis(w["a_tmp"], n, S, res, 0, t, w["alpha"], shift=0.5, scale=0.26)
            f = self.alpha_clip_model.visual(img, w["alpha"][:n], mean=self._mean, std=self._std, **({"on_block": on_block} if on_block is not None and i == 0 else {}))
            w["feats"][i:i + n].copy_(f)
        return w["feats"]

    def _score_proposals(self, query_img, mask_proposals, support_mask, cost_matrix, G: int, vva, vta, text):
        dev = self.device
        st, self._begun = self._begun, None
        if st is not None and not self._same_inputs(st["key"], self._input_key(query_img, mask_proposals, support_mask, G)):
            if st["job"]["thread"] is not None:
                st["job"]["thread"].join()      # started for other inputs: let it finish, then start over
            st = None
        if st is None:
            st = self.begin(query_img, mask_proposals, support_mask, cost_matrix, G)
            self._begun = None
        w, masks, N, P, feats = st["w"], st["masks"], st["N"], st["P"], st["feats"]
        ops.proposal_stats(w["bits"], N, P, vva.to(dev).contiguous().float().view(-1), vta.to(dev).contiguous().float().view(-1),
                           w["cnt"], w["svv"], w["svt"], w["ucnt"])
        # lines 90-97: the AlphaCLIP text features and the cosines (queued behind the batch)
        prepared, self._prepared_text = getattr(self, "_prepared_text", None), None      # taken once, never reused
        if prepared is not None and prepared[0] == tuple(text):
            tfeat = prepared[1]
            here = torch.cuda.current_stream(dev)
            here.wait_event(prepared[2])
            tfeat.record_stream(here)       # made on the prompt stream, read here: its block is not reused before this stream got there
        else:
            tfeat = self._compute_alphaclip_text_feats(text)
        ops.cosine_scores(feats, tfeat, w["aclip"])
        # the transport scores: through pinned memory, so that the copy and everything behind it are queued
        # while the batch is still running and the GPU goes from the batch to the ranking without the host
        if st["job"]["thread"] is None:
            self._start_solver(st["job"], cost_matrix)
        t_join = time.perf_counter()
        st["job"]["thread"].join()
        if st["job"]["error"] is not None:
            raise st["job"]["error"]
        # how long the solver took, when it was done (since begin(): compare with the step's length - the GPU needs
        # about a millisecond behind the batch for ranking and merge, so a solver done later than that held the
        # step up) and how long the host stood here for it after queueing everything else
        self.timing = dict(emd_ms=1e3 * st["job"].get("solve_s", 0.0), emd_end_ms=1e3 * st["job"].get("end_s", 0.0),
                           emd_wait_ms=1e3 * (time.perf_counter() - t_join), emd_threads=self.emd_threads)
        if w.get("emd_copied") is not None:
            w["emd_copied"].synchronize()       # the copy of the call before has read the pinned buffer
        w["emd_host"].copy_(st["job"]["result"])
        w["emd"].copy_(w["emd_host"], non_blocking=True)
        w["emd_copied"] = torch.cuda.Event()
        w["emd_copied"].record()
        # lines 118-138, 213-217: combine, rank, keep
        ops.rank(w["emd"], w["aclip"], w["cnt"], w["svv"], w["svt"], w["ucnt"], N, self.alpha, self.static_threshold,
                 self.dynamic_threshold, w["score"], w["order"], w["keep"])
        self.last = dict(w=w, masks=masks, N=N, tfeat=tfeat)     # tfeat: kept alive until the next call (it may live on another stream's pool)
        return w

    def _compute_emd_scores(self, cost_matrix, fg_host: torch.Tensor, bits_host: torch.Tensor) -> torch.Tensor:
        """Lines 112-116 / 142-169 for all proposals at once: float64 [N] of 1 - emd2 between the uniform
        distributions on the support's foreground patches (``fg_host`` u8 [ns*P]) and on each proposal's
        patches (``bits_host`` u8 [N, P]), solved exactly on host threads."""
        if isinstance(cost_matrix, SimilarityHandle):
            return ops.emd_scores_host(cost_matrix.wait(), fg_host, bits_host, self.emd_threads, is_cost=False)
        cm = cost_matrix.detach().cpu().float().contiguous()
        return ops.emd_scores_host(cm, fg_host, bits_host, self.emd_threads, is_cost=True)

    def _merge_masks(self, ranked_masks=None) -> torch.Tensor:
        """Lines 209-221 on the keep flags the ranking kernel produced."""
        w, masks = self.last["w"], self.last["masks"]
        ops.merge_masks(masks, w["keep"], w["merged"])
        return w["merged"]

    def scores(self) -> dict:
        """Host copies of the per-proposal quantities of the last call (for tests and logging)."""
        w, N = self.last["w"], self.last["N"]
        return {k: w[k][:N].cpu() for k in ("emd", "aclip", "cnt", "svv", "svt", "score", "order", "keep")} | {
            "ucnt": w["ucnt"].cpu(), "feats": w["feats"][:N].cpu()}


def build_filtering_and_merging_module(args, state_dict=None):
    """Reference builder (lines 223-246) with weights injectable through ``state_dict``."""
    from ...utils.backbone_loader import BackboneLoader
    model, transforms = BackboneLoader.load_backbone(
        backbone_name="alphaclip", backbone_size=getattr(args, "alphaclip_backbone", "ViT-L/14@336px"), device=args.device,
        backbone_weights_path=getattr(args, "alphaclip_weights", None),
        encoder_kwargs={"device": args.device, "download_root": getattr(args, "models_path", None), "state_dict": state_dict,
                        "dtype": getattr(args, "dtype", torch.float32)})
    # the proposal tower's attention with 8-bit float operands (bf16 towers only): args.fp8_attention or MARS_FP8_ATTENTION=1
    if (getattr(args, "fp8_attention", False) or os.environ.get("MARS_FP8_ATTENTION", "0") == "1") and hasattr(model, "vtower"):
        model.vtower.fp8_attention = True
    return FilteringMergingModule(
        alpha_clip_model=model, img_transforms=transforms[0], mask_transforms=transforms[1], alpha=args.alpha_coverage,
        static_threshold=args.static_threshold, dynamic_threshold=args.dynamic_threshold, device=args.device)
