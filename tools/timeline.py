"""Where a benchmark step's wall time goes, without a profiler: host clock at the phase boundaries of
MARS.predict and GPU events at the same places, for a few steps of the bench workload."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package(require_native=True)
from importlib import import_module  # noqa: E402

MARS = import_module(pkg.__name__ + ".mars.MARS")
FMM = import_module(pkg.__name__ + ".mars.components.FilteringMergingModule").FilteringMergingModule
Evaluator = import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
AverageMeter = import_module(pkg.__name__ + ".mars.utils.logger").AverageMeter
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sds = bench.state_dicts(pkg)
mars = MARS.build_MARS_fss(bench.build_args(dev, torch.bfloat16), state_dicts=sds)
Evaluator.initialize()
meter = AverageMeter(types.SimpleNamespace(benchmark="coco", class_ids=[4 * i for i in range(20)]), device=dev)
eps = []
NSTEPS = int(os.environ.get("STEPS", "4"))
for e in range(2 + NSTEPS):
    ep = pkg.synth.make_episode(e, size=bench.SIZE, nshot=1, nprop=bench.NPROP, fold=0)
    eps.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ep.items()})

marks = []          # (label, host seconds, event or None)


def mark(label, with_event=True):
    ev = None
    if with_event:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
    marks.append((label, time.perf_counter(), ev))


def wrap(obj, name, before=None, after=None, ev_after=True):
    f = getattr(obj, name)

    def g(*a, **k):
        if before:
            mark(before)
        r = f(*a, **k)
        if after:
            mark(after, ev_after)
        return r
    setattr(obj, name, g)


fmm = mars.filtering_merging_component
wrap(mars, "_predict", before="start")
wrap(mars.visual_visual_alignment_component, "compute", before="vva begins", after="vva queued")
wrap(mars.visual_text_alignment_component, "compute", after="vta queued")
wrap(fmm, "_compute_alphaclip_vis_feats", before="alphaclip begins", after="alphaclip queued")
orig_emd = fmm._compute_emd_scores


def emd_timed(*a, **k):           # runs on the solver's own host thread: host clock only
    marks.append(("emd begin (bits on host)", time.perf_counter(), None))
    r = orig_emd(*a, **k)
    marks.append(("emd solved", time.perf_counter(), None))
    return r


fmm._compute_emd_scores = emd_timed
wrap(fmm, "_compute_alphaclip_text_feats", after="alphaclip text queued")
wrap(fmm, "_merge_masks", before="rank queued", after="merge queued")


def step(ep):
    pred = mars.predict(ep["support_imgs"], ep["support_masks"], ep["query_img"], ep["mask_proposals"])
    marks.append(("predict returned", time.perf_counter(), None))
    ai, au = Evaluator.classify_prediction(pred[None], ep)
    meter.update(ai, au, ep["class_id"])
    mars.clear()
    mark("step end")


for k in range(2):
    step(eps[k])
torch.cuda.synchronize()
for k in range(2, 2 + NSTEPS):
    marks.clear()
    t_prev = time.perf_counter()
    step(eps[k])
    torch.cuda.synchronize()
    t0 = marks[0][1]
    e0 = marks[0][2]
    byl = {m[0]: m for m in marks}
    emd_ms = 1e3 * (byl["emd solved"][1] - byl["emd begin (bits on host)"][1])
    print(f"step {k}: wall {1e3 * (marks[-1][1] - t0):.2f} ms, emd {emd_ms:.1f} ms from {1e3 * (byl['emd begin (bits on host)'][1] - t0):.1f}, "
          f"batch done on the gpu at {e0.elapsed_time(byl['alphaclip queued'][2]):.1f}")
    if NSTEPS > 6:
        continue
    for label, t, ev in sorted(marks, key=lambda m: m[1]):
        gpu = f"{e0.elapsed_time(ev):8.2f}" if ev is not None else "       -"
        print(f"   {label:28s} host {1e3 * (t - t0):8.2f} ms   gpu reaches it at {gpu} ms")
