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
    env["HIP_VISIBLE_DEVICES"] = "0"                 # one visible GPU, whatever the machine has
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "visible" in r.stderr


def test_visible_gpus_makes_no_gpu_call(monkeypatch):
    """The launching process counts GPUs from the visibility lists (or the KFD topology), never through torch.cuda."""
    import bench
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("torch.cuda called in the launcher")))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    n = bench.visible_gpus()
    assert n is None or n >= 0


def test_rank_cpu_slices_are_disjoint_and_equal():
    import bench
    cpus = list(range(3, 99))                        # 96 cores, not starting at 0
    slices = [bench.rank_cpu_slice(cpus, r, 8) for r in range(8)]
    assert all(len(s_) == 12 for s_ in slices)
    assert sorted(sum(slices, [])) == cpus
    assert bench.rank_cpu_slice(cpus, 0, 1) == cpus
    assert bench.rank_cpu_slice([5, 7, 9], 1, 2) == [7]          # remainder unused: same count for every rank
    assert bench.rank_cpu_slice([4], 3, 8) == [4]                # fewer cores than ranks: left alone


_RANK_SCRIPT = """
import os, sys, time
r = int(os.environ["RANK"])
open(os.path.join(sys.argv[1], "pid%d" % r), "w").write(str(os.getpid()))
mode = sys.argv[2]
if mode == "fail" and r == 1:
    time.sleep(0.3)
    sys.exit(3)
if mode == "ok":
    assert os.environ["WORLD_SIZE"] == "3" and os.environ["MARS_BENCH_STORE"]
    sys.exit(0)
time.sleep(120)
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a child that was reaped is gone; one that is a zombie of ours has been waited for by launch_ranks
    try:
        return open(f"/proc/{pid}/stat").read().split(")")[1].split()[0] != "Z"
    except OSError:
        return False


def test_launch_ranks_stops_everyone_when_one_rank_dies(tmp_path):
    """One rank exits with status 3 while the others would run for two minutes: the launcher returns 3 within
    seconds and leaves no rank behind."""
    import time
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    t0 = time.monotonic()
    rc = bench.launch_ranks(3, [str(tmp_path), "fail"], script=str(script), deadline_s=60)
    assert rc == 3 and time.monotonic() - t0 < 15
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(3)]
    assert not any(_alive(p) for p in pids)


def test_launch_ranks_deadline_and_clean_exit(tmp_path):
    import time
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    assert bench.launch_ranks(3, [str(tmp_path), "ok"], script=str(script), deadline_s=60) == 0
    t0 = time.monotonic()
    rc = bench.launch_ranks(2, [str(tmp_path), "hang"], script=str(script), deadline_s=1.0)
    assert rc == 124 and time.monotonic() - t0 < 15
    assert not any(_alive(int((tmp_path / f"pid{r}").read_text())) for r in range(2))
