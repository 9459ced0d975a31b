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


# ---------------------------------------------------------------------------- bench.py's own sharding / reduction
def _bench_worker(rank, world, port, per_rank, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    meter = _make_meter()
    mine = bench.rank_episodes(rank, world, per_rank)
    for e in mine:
        meter.update(*_episode_areas(e))
    elapsed = bench.reduce_over_ranks(meter, 1.0 + rank, torch.device("cpu"), world)     # rank 1 is the slow one
    if rank == 0:
        torch.save({"inter": meter.intersection_buf, "union": meter.union_buf, "elapsed": elapsed, "mine": mine,
                    "ranks": dist.get_world_size()}, out)
    dist.destroy_process_group()


def test_bench_sharding_and_reduction_with_two_ranks(tmp_path):
    """bench.py's episode-to-rank map and its two collectives (SUM of the count buffers, MAX of the elapsed
    time) with two gloo ranks against the single-process episode list."""
    import bench
    per_rank, world = 5, 2
    shards = [bench.rank_episodes(r, world, per_rank) for r in range(world)]
    assert sorted(sum(shards, [])) == list(range(per_rank * world))           # a partition of the global list
    assert bench.rank_episodes(0, 1, 7) == list(range(7))
    ref = _make_meter()
    for e in range(per_rank * world):
        ref.update(*_episode_areas(e))
    out = str(tmp_path / "bench_r0.pt")
    mp.spawn(_bench_worker, args=(world, 29541, per_rank, out), nprocs=world, join=True)
    got = torch.load(out)
    assert got["mine"] == shards[0] and got["ranks"] == world
    assert torch.equal(got["inter"], ref.intersection_buf) and torch.equal(got["union"], ref.union_buf)
    assert got["elapsed"] == 2.0                                              # the maximum over ranks


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus N under a launcher that made a different world, or with fewer visible GPUs than N, must fail
    loudly instead of reporting a smaller run (both checks come before anything touches a GPU)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
        assert r.returncode != 0 and "visible" in r.stderr
