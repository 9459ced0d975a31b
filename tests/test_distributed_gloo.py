"""N > 1 path on CPU: two gloo ranks shard episodes round-robin, accumulate IoU areas locally and
sum the [2, nclass] buffers with one all-reduce; the result must equal the single-process run."""
import importlib
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _episode_areas(e):
    """Deterministic stand-in for one episode's (area_inter, area_union, class_id): the sharding and
    the reduction are what is under test, not the kernels."""
    g = torch.Generator().manual_seed(1000 + e)
    pred = (torch.rand(40, 40, generator=g) > 0.5).float()
    gt = (torch.rand(40, 40, generator=g) > 0.6).float()
    from oracle import pipeline as opipe
    ai, au = opipe.classify_prediction(pred, gt)
    return ai.view(2, 1), au.view(2, 1), torch.tensor([4 * (e % 20)])


def _make_meter():
    from __graft_entry__ import load_package
    pkg = load_package(require_native=False)
    logger = importlib.import_module(pkg.__name__ + ".mars.utils.logger")
    ds = type("DS", (), {"benchmark": "coco", "class_ids": [4 * i for i in range(20)]})()
    return logger.AverageMeter(ds, device="cpu")


def _worker(rank, world, port, episodes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    meter = _make_meter()
    for e in range(rank, episodes, world):          # episode e belongs to rank e mod world
        ai, au, cid = _episode_areas(e)
        meter.update(ai, au, cid)
    meter.all_reduce()
    miou, fb, _ = meter.compute_iou()
    if rank == 0:
        torch.save({"miou": float(miou), "fb": float(fb), "inter": meter.intersection_buf, "union": meter.union_buf}, out)
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    episodes = 13                                   # odd on purpose: ranks get 7 and 6 episodes
    ref = _make_meter()
    for e in range(episodes):
        ref.update(*_episode_areas(e))
    miou_ref, fb_ref, _ = ref.compute_iou()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, 29533, episodes, out), nprocs=2, join=True)
    got = torch.load(out)
    assert torch.equal(got["inter"], ref.intersection_buf) and torch.equal(got["union"], ref.union_buf)
    assert got["miou"] == float(miou_ref) and got["fb"] == float(fb_ref)
