"""Benchmark of the MARS alignment-and-ranking path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one query episode through the whole ranking path (the region the reference
timestamps as ranking_time_after_text_extraction, /root/reference/mars/MARS.py:59-102) plus
its IoU accounting.  Workload at every N: BASELINE.json configs[1] -- COCO-20i fold 0, 1-shot,
DINOv2-L/14 reg4 @518, CLIP ViT-B/16, AlphaCLIP ViT-L/14@336, 64 proposals, bf16 towers --
on seeded synthetic weights and episodes that are resident in HBM before the clock starts.
Episodes shard round-robin over ranks (weak scaling: K episodes per GPU); the only collective
is the sum of the [2, nclass] intersection/union buffers.

Launching: under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) this file
is one rank.  Called bare with --gpus N > 1 it starts the N ranks itself, as fresh child processes, before
anything touches the GPU, and exits with the first non-zero child status.  A world size that differs from
--gpus is an error, never a silently smaller run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types


def rank_cpu_slice(cpus, local_rank: int, local_world: int):
    """The cores rank ``local_rank`` of ``local_world`` ranks on this node keeps: the sorted core list cut into
    ``local_world`` runs of equal length (neighbouring core numbers share caches and a memory controller), the
    remainder left unused so that every rank has the same number.  Fewer cores than ranks: the list as it is."""
    cpus = sorted(cpus)
    per = len(cpus) // max(local_world, 1)
    if local_world <= 1 or per < 1:
        return cpus
    return cpus[local_rank * per:(local_rank + 1) * per]


def pin_rank_cpus():
    """Gives this rank a slice of the node's cores of its own (the transport solver's threads and torch's
    pools inherit it), before torch starts any thread.  MARS_BENCH_PIN=0 leaves the mask alone."""
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    lr = int(os.environ.get("LOCAL_RANK", "0") or 0)
    if lw <= 1 or os.environ.get("MARS_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    mine = rank_cpu_slice(os.sched_getaffinity(0), lr, lw)
    os.sched_setaffinity(0, mine)
    os.environ["MARS_RANK_PINNED"] = "1"          # default_emd_threads: the mask is already this rank's share
    return mine


if __name__ == "__main__":
    pin_rank_cpus()

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X
NPROP, SIZE, FOLD = 64, 518, 0
# Rehearsal of the N > 1 path on a one-GPU machine (tests only): every rank uses GPU 0 and the ranks meet over gloo.
# The line such a run prints says so ("shared_gpu": true); it is not a scaling measurement.
SHARE_GPU = os.environ.get("MARS_BENCH_SHARE_GPU", "0") == "1"
# kernel instantiation behind each launch-policy variant (csrc/gemm.hip, csrc/gemm3.hip)
RING_KERNELS = {3: "gemm_tn_ring_kernel<bf16, *, 2, 4, 8, 4, 2, 3> (256x256 tile)",
                5: "gemm_tn_ring_kernel<bf16, *, 2, 2, 8, 4, 2, 3> (256x128 tile)",
                6: "gemm_tn_ring_kernel<bf16, *, 2, 2, 4, 4, 3, 3> (128x128 tile)",
                7: "gemm_tn_ring_kernel<bf16, *, 2, 2, 2, 4, 4, 3> (64x128 tile)",
                8: "gemm_tn_ring_kernel<bf16, *, 2, 4, 8, 4, 2, 4, true> (256x256 tile, pair fills)",
                10: "gemm_tn_ring_kernel<bf16, *, 2, 2, 2, 4, 2, 6, true> (64x128 tile, pair fills)",
                11: "gemm_tn_ring_kernel<bf16, *, 2, 2, 4, 4, 2, 4, true> (128x128 tile, pair fills)"}


def rank_episodes(rank: int, world: int, per_rank: int):
    """Global indices of the episodes rank ``rank`` of ``world`` runs, in order: index k * world + rank
    for k < per_rank.  The ranks' lists partition range(per_rank * world) (weak scaling: per_rank each)."""
    if not (0 <= rank < world and per_rank >= 0):
        raise ValueError(f"bad shard: rank {rank} of {world}, {per_rank} episodes per rank")
    return [k * world + rank for k in range(per_rank)]


def _all_reduce(t, op):
    """all_reduce of a small tensor; a gloo group (CPU tests, the one-card rehearsal) reduces a host copy."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        t = t.cpu()
    dist.all_reduce(t, op=op)
    return t


def reduce_over_ranks(meter, elapsed: float, device, world: int) -> float:
    """The run's only collectives: SUM of the meter's [2, nclass] count buffers, MAX of the ranks' elapsed
    seconds (returned).  With one rank nothing is exchanged."""
    import torch.distributed as dist
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        el = _all_reduce(el, dist.ReduceOp.MAX)
        meter.all_reduce()
    return float(el)


def visible_gpus():
    """GPUs the ranks will see, found WITHOUT any HIP or torch.cuda call (the launching process must never
    initialise the runtime): the visibility lists a launcher or the box sets, else the compute nodes with
    SIMDs in the KFD topology.  None when neither can be read - then the ranks check for themselves."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            return None
        n += int(props.get("simd_count", "0")) > 0
    return n


def gather_host_side(host: dict, device, world: int):
    """Per rank: [mean solver ms, max solver ms, mean and max over steps of (step ms - ms at which the solver was
    done, both from the start of the step), mean step ms, cores in the rank's mask] - gathered (a SUM of one-hot
    rows: the same collective as the counts) so that rank 0 can say whether a rank's steps were held up by its
    host solver (slack of a millisecond or less: the GPU's ranking and merge behind the batch) or by its GPU."""
    import torch.distributed as dist
    n = max(len(host["step_ms"]), 1)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    slack = [a - b for a, b in zip(host["step_ms"], host["emd_end_ms"])] or [0.0]
    row = [sum(host["emd_ms"]) / n, max(host["emd_ms"] or [0.0]), sum(slack) / n, min(slack), sum(host["step_ms"]) / n, float(cores)]
    table = torch.zeros(world, len(row), dtype=torch.float64, device=device)
    table[int(os.environ.get("RANK", "0")) if world > 1 else 0] = torch.tensor(row, dtype=torch.float64, device=device)
    if world > 1:
        table = _all_reduce(table, dist.ReduceOp.SUM)
    return table.cpu().tolist()


def launch_ranks(n: int, argv, script=None, deadline_s=None, poll_s=0.05) -> int:
    """Start ``n`` ranks of this file (or of ``script``) as child processes - this process has not touched the
    GPU and never does - and watch them all at once: the first rank that exits non-zero ends the run (the
    others are terminated, then killed, and that status is returned), and so does ``deadline_s`` (status
    124).  The ranks meet through a file store in a directory of their own, so no port is picked here that
    another process could take before rank 0 binds it."""
    import shutil
    import subprocess
    import tempfile
    deadline_s = float(os.environ.get("MARS_BENCH_DEADLINE", "1500")) if deadline_s is None else deadline_s
    store = tempfile.mkdtemp(prefix="mars_bench_rdzv_")
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MARS_BENCH_STORE=os.path.join(store, "store"))
            # GPU buffers are shared between processes through dmabuf handles on this platform; the legacy IPC
            # mode fails in hipIpcGetMemHandle, and RCCL opens its peers' buffers that way
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
        t_end = time.monotonic() + deadline_s
        rc = 0
        while rc == 0 and any(p.poll() is None for p in procs):
            rc = next((p.returncode for p in procs if p.poll() not in (None, 0)), 0)
            if rc == 0 and time.monotonic() > t_end:
                print(f"bench: ranks still running after {deadline_s:.0f} s, stopping them", file=sys.stderr, flush=True)
                rc = 124
            if rc == 0:
                time.sleep(poll_s)
        rc = rc or next((p.returncode for p in procs if p.returncode), 0)
        return rc
    finally:
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in alive:
            try:
                p.wait(timeout=max(0.0, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        shutil.rmtree(store, ignore_errors=True)


def host_cores() -> int:
    """Cores this process may use: its affinity mask, capped by OMP_NUM_THREADS when the box sets one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(n, cap) if cap > 0 else n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_args(device, dtype):
    return types.SimpleNamespace(
        input_size=SIZE, num_regs=4, dino_backbone="vit_large", vta_backbone="ViT-B/16", alphaclip_backbone="ViT-L/14@336px",
        vva_refinement_box_threshold=0.8, last_n_attn_for_vva_refinement=24, vta_refinement_box_threshold=0.4,
        last_n_attn_for_vta_refinement=8, static_threshold=0.55, dynamic_threshold=0.95, alpha_coverage=0.85,
        device=device, dtype=dtype, class_name="object", class_description="")


def state_dicts(pkg, seed=0):
    synth = pkg.synth
    return dict(dino=synth.dinov2_state_dict(synth.dino_cfg("vit_large", SIZE), seed),
                clip=synth.clip_state_dict(synth.CLIP_ARCHS["ViT-B/16"], seed),
                alphaclip=synth.clip_state_dict(synth.CLIP_ARCHS["ViT-L/14@336px"], seed, alpha=True))


def cpu_baseline(pkg, sds, warm_props=2):
    """The CPU oracle on ONE whole episode of the same workload, measured, not scaled: both DINOv2-L passes,
    the CLIP pass with GradCAM and refinement, and all 64 proposals through AlphaCLIP + exact EMD + ranking, on
    every core this process may use.  A short pass (the towers and ``warm_props`` proposals) runs first so
    that thread pools and pages are warm.  Runs after the timed GPU region, on rank 0 of a one-GPU run only."""
    from oracle import pipeline as opipe
    synth = pkg.synth
    cores = host_cores()
    torch.set_num_threads(cores)
    d, c, a = synth.dino_cfg("vit_large", SIZE), synth.CLIP_ARCHS["ViT-B/16"], synth.CLIP_ARCHS["ViT-L/14@336px"]
    models = dict(
        dino_sd=sds["dino"], dino_kw=dict(depth=d.depth, heads=d.num_heads, patch=14, num_regs=4, ffn="mlp"),
        clip_sd=sds["clip"], clip_kw=dict(layers=c.vision_layers, heads=c.vision_heads, patch=16, text_layers=c.transformer_layers,
                                          text_heads=c.transformer_heads),
        aclip_sd=sds["alphaclip"], aclip_kw=dict(layers=a.vision_layers, heads=a.vision_heads, patch=14, resolution=336,
                                                 text_layers=a.transformer_layers, text_heads=a.transformer_heads))
    ep = synth.make_episode(1000 * FOLD, size=SIZE, nshot=1, nprop=NPROP, fold=FOLD)
    tok = dict(fg=synth.tokenize(["a photo of object."]), bg=synth.tokenize(["a photo without object."]),
               alphaclip=synth.tokenize(["a object."]))

    def one_pass(nprop):
        t0 = time.time()
        with torch.no_grad():
            v = opipe.vva_compute(models["dino_sd"], models["dino_kw"], ep["support_imgs"], ep["support_masks"], ep["query_img"])
        t1 = time.time()
        t = opipe.vta_compute(models["clip_sd"], models["clip_kw"], ep["query_img"], tok["fg"], tok["bg"])   # GradCAM needs autograd
        t2 = time.time()
        g = v["vva"].shape[0]
        vta = opipe.minmax(torch.nn.functional.interpolate(t["vta"][None, None], (g, g), mode="nearest")[0, 0])
        props = ep["mask_proposals"][:nprop]
        with torch.no_grad():
            s = opipe.fmm_scores(models["aclip_sd"], models["aclip_kw"], ep["query_img"], props, ep["support_masks"], v["cost"], g,
                                 v["vva"], vta, tok["alphaclip"], batch=min(8, nprop))
        mask, _ = opipe.fmm_merge(props, s["score"], s["order"])
        opipe.classify_prediction(mask, ep["query_mask"][0])
        t3 = time.time()
        return t1 - t0, t2 - t1, t3 - t2

    one_pass(warm_props)
    dino_s, clip_s, rank_s = one_pass(NPROP)
    per_query = dino_s + clip_s + rank_s
    return {"value": 1.0 / per_query, "unit": "query images/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "seconds_per_query": per_query,
            "sample": f"1 whole episode of the same workload after a warm-up pass, nothing scaled: DINOv2-L x2 {dino_s:.1f}s + "
                      f"CLIP-B/16 GradCAM and refinement {clip_s:.1f}s + all {NPROP} proposals through AlphaCLIP-L/14@336, exact EMD, "
                      f"ranking, merge and IoU {rank_s:.1f}s; fp32 torch CPU oracle on {cores} threads"}


def pmc_traffic():
    """HBM-side bytes per GEMM launch from the COMMITTED rocprofv3 counter passes under profiles/ (FETCH_SIZE and
    WRITE_SIZE, separate runs, FETCH_SIZE doubled for gfx950; see tools/pmc_traffic.py), or (None, None).  Counters
    cannot be read from inside the run that prints the line, so the figure is a stored one and the line says
    which file it came from (``traffic_source``)."""
    import glob
    import re
    tag = lambda p: [int(x) for x in re.findall(r"\d+", os.path.basename(p))]     # r1_v10 after r1_v9  # noqa: E731
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=tag, reverse=True):   # newest tag first
        try:
            return float(json.load(open(path))["bytes_per_launch"]), "stored profile " + os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="side measurement: 8-bit float operands in the AlphaCLIP attention (not the headline configuration)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks here; this process makes no HIP or torch.cuda call, before or after
        visible = visible_gpus()
        if visible is not None and visible < args.gpus and not SHARE_GPU:
            sys.exit(f"--gpus {args.gpus} but only {visible} GPU(s) are visible")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if SHARE_GPU else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    if local >= torch.cuda.device_count():
        sys.exit(f"rank {rank}: local rank {local} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    import torch.distributed as dist
    if world > 1:
        # ranks started by launch_ranks meet through its file store, ranks of a launcher through MASTER_ADDR / MASTER_PORT
        store = os.environ.get("MARS_BENCH_STORE")
        kw = dict(init_method="file://" + store) if store else {}
        if SHARE_GPU:       # rehearsal on one card: RCCL refuses two ranks on a device, gloo carries the reduction
            dist.init_process_group("gloo", rank=rank, world_size=world, **kw)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, **kw)

    pkg = load_package(require_native=True)
    from importlib import import_module
    MARS = import_module(pkg.__name__ + ".mars.MARS")
    Evaluator = import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
    AverageMeter = import_module(pkg.__name__ + ".mars.utils.logger").AverageMeter
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    sds = state_dicts(pkg)
    margs = build_args(dev, dtype)
    margs.fp8_attention = args.fp8_attention
    mars = MARS.build_MARS_fss(margs, state_dicts=sds)
    Evaluator.initialize()
    dataset = types.SimpleNamespace(benchmark="coco", class_ids=[FOLD + 4 * i for i in range(20)])
    meter = AverageMeter(dataset, device=dev)

    # this rank's episodes (seed = 1000*fold + global index), resident in HBM before the clock starts
    total = args.steps + args.warmup
    eps = []
    for e in rank_episodes(rank, world, total):
        ep = pkg.synth.make_episode(1000 * FOLD + e, size=SIZE, nshot=1, nprop=NPROP, fold=FOLD)
        eps.append({k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in ep.items()})

    host = {"emd_ms": [], "emd_end_ms": [], "step_ms": []}       # per timed step of this rank

    # warm-up steps go through everything a timed step does, the meter included (into one of their own): the runtime
    # loads a code object when one of its kernels is first launched, and the first `index_add_` of a process brings
    # in torch's indexing module (about 120 ms) - inside the timed region if the warm-up had skipped the meter
    warm_meter = AverageMeter(dataset, device=dev)

    def step(ep, count=True):
        t_s = time.perf_counter()
        pred = mars.predict(ep["support_imgs"], ep["support_masks"], ep["query_img"], ep["mask_proposals"])
        ai, au = Evaluator.classify_prediction(pred[None], ep)
        if not count:
            warm_meter.update(ai, au, ep["class_id"])
        if count:
            meter.update(ai, au, ep["class_id"])
            tm = getattr(mars.filtering_merging_component, "timing", {})
            host["emd_ms"].append(tm.get("emd_ms", 0.0))
            host["emd_end_ms"].append(tm.get("emd_end_ms", 0.0))
            host["step_ms"].append(1e3 * (time.perf_counter() - t_s))     # predict() ends with a device synchronize
        mars.clear()

    for k in range(args.warmup):
        step(eps[k], count=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # The GEMM launches of the LAST timed step are bracketed by events (two records per launch cost
    # about 5 % of a step when every step carries them, so the other steps run bare).
    pkg.ops.GEMM_PROFILE = None
    pkg.ops.GEMM_PROFILE_DTYPE = 1 if args.dtype == "bf16" else 0
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        if k == total - 1 and not os.environ.get("MARS_BENCH_NO_GEMM_EVENTS"):
            pkg.ops.GEMM_PROFILE = []
        step(eps[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof, pkg.ops.GEMM_PROFILE = pkg.ops.GEMM_PROFILE or [], None

    # One more step OUTSIDE the timed region with the stages one after the other on one stream: in the timed steps
    # the visual-visual tower, the text-alignment tower and the prompt encoders run beside the proposal batch, so
    # an event pair around a launch of the batch also spans the share of the chip those took meanwhile.
    prof_serial = []
    if rank == 0 and getattr(mars, "concurrent", False) and prof:
        mars.concurrent = False
        pkg.ops.GEMM_PROFILE = []
        step(eps[-1], count=False)
        torch.cuda.synchronize()
        prof_serial, pkg.ops.GEMM_PROFILE = pkg.ops.GEMM_PROFILE or [], None
        mars.concurrent = True

    elapsed = reduce_over_ranks(meter, elapsed, dev, world)
    host_all = gather_host_side(host, dev, world)
    ranks_joined = dist.get_world_size() if world > 1 else 1          # what RCCL itself reports
    miou, fb_iou, _ = meter.compute_iou()

    if rank == 0:
        # Every MFMA GEMM launch of a sampled step was timed with events on its stream.  The dominant
        # kernel is the ring shape with the largest summed time (the AlphaCLIP products run on three of them:
        # 256 x 256 with pair fills for QKV, FC1 and FC2, 128 x 128 for the projection); every launch,
        # the smaller shapes of the two alignment towers included, is summed under "all_gemm".
        def summarise(profile):
            timed = [(p[0].elapsed_time(p[1]), p[2], p[3]) for p in profile]
            by_variant = {}
            for t in timed:
                by_variant[t[2]] = by_variant.get(t[2], 0.0) + t[0]
            dom_v = max(by_variant, key=by_variant.get) if by_variant else None
            dom = [t for t in timed if t[2] == dom_v] or timed
            flops, ms = sum(t[1] for t in dom), sum(t[0] for t in dom)
            ms_all = sum(t[0] for t in timed)
            return dict(variant=dom_v, launches=len(dom), ms=ms, achieved=flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                        all_achieved=sum(t[1] for t in timed) / (ms_all * 1e-3) / 1e12 if ms_all > 0 else 0.0,
                        all_launches=len(timed), all_ms=ms_all)

        step_ms = elapsed / args.steps * 1e3
        in_step = summarise(prof)
        # In the timed steps the stages of a query run side by side, so an event pair around a launch of the batch also
        # spans the share of the chip that other streams' kernels took meanwhile.  The kernel's own rate is the one of
        # the extra step (same workload, same launches, nothing beside them): that is also what a kernel trace of this
        # command reports per launch.  Without such a step (stages already one after the other) both are the same.
        alone = summarise(prof_serial) if prof_serial else in_step
        dom_variant = alone["variant"]
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        # HBM-side bytes were counted (stored counter passes) on the pair-fill 256x256 kernel only
        traffic, traffic_src = pmc_traffic() if args.dtype == "bf16" and dom_variant == 8 else (None, None)
        out = {
            "metric": "query images/sec", "value": args.steps * world / elapsed, "unit": "query images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype + ("+fp8-attention" if args.fp8_attention else ""), "data": "synthetic",
            "config": {"workload": "COCO-20i fold0 1-shot, DINOv2-L/14-reg4@518 + CLIP ViT-B/16 GradCAM + AlphaCLIP ViT-L/14@336, "
                                   "64 proposals, exact EMD, rank+merge+IoU (BASELINE configs[1])",
                       "episodes_per_gpu": args.steps, "proposals": NPROP, "parallelism": f"episode-sharded x{world}", "rccl_ranks": ranks_joined,
                       **({"shared_gpu": True, "note": "rehearsal: all ranks on GPU 0 over gloo, not a scaling measurement"} if SHARE_GPU else {})},
            "miou": float(miou), "fb_iou": float(fb_iou),
            # host side of every rank: the exact-EMD solver runs on host threads beside the GPU's proposal batch; a step whose
            # solver was done less than about a millisecond before the step ended (slack) was bound by the rank's cores
            "host": {"emd_threads": getattr(mars.filtering_merging_component, "emd_threads", None),
                     "pinned": os.environ.get("MARS_RANK_PINNED", "0") == "1",
                     "per_rank": [dict(zip(("emd_ms", "emd_ms_max", "emd_slack_ms", "emd_slack_ms_min", "step_ms", "cores"), [round(x, 3) for x in r]))
                                  for r in host_all]},
            "roofline": {"bound": "mfma", "achieved": alone["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": alone["achieved"] / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": RING_KERNELS.get(dom_variant, f"gemm variant {dom_variant}").replace(
                             "<bf16,", "<bf16," if args.dtype == "bf16" else "<f32,"),
                         "launches": alone["launches"], "avg_launch_us": alone["ms"] * 1e3 / max(alone["launches"], 1), "sampled_steps": 1,
                         "sampled": ("one extra step right after the timed region with the stages of the query queued one after the other: "
                                     "the launches of the kernel with nothing beside them" if prof_serial else "the last timed step"),
                         "share_of_step": alone["ms"] / step_ms,
                         "all_gemm": {"achieved": alone["all_achieved"], "launches": alone["all_launches"], "share_of_step": alone["all_ms"] / step_ms},
                         # the same event pairs in the last TIMED step, where kernels of the other stages share the chip with the launch
                         "in_timed_step": {"achieved": in_step["achieved"], "frac": in_step["achieved"] / peak,
                                           "avg_launch_us": in_step["ms"] * 1e3 / max(in_step["launches"], 1), "launches": in_step["launches"],
                                           "all_gemm_achieved": in_step["all_achieved"]}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, sds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
