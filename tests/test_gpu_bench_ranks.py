"""bench.py's N > 1 path as a fresh process on a one-GPU machine: ``--gpus 2`` starts two ranks through
``launch_ranks``; a test-only switch (MARS_BENCH_SHARE_GPU=1) puts both on GPU 0 and lets them meet over gloo
(RCCL refuses two ranks on one device).  Everything else is what an 8-GPU run executes: the rendezvous through
the launcher's store, a core range per rank, the episode shards, the count buffers as device tensors, their
sum over ranks, the MAX of the elapsed time and rank 0's line.  The counted episodes of the two ranks are the
ones a single process counts with twice the steps, so the IoU figures must be bitwise the same."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, extra_env, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MARS_BENCH_STORE")}
    env.update(extra_env, MARS_BENCH_NO_GEMM_EVENTS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints, nobody else
    return json.loads(lines[0])


def test_two_ranks_on_one_card_equal_one_process():
    two = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"MARS_BENCH_SHARE_GPU": "1", "MARS_EMD_THREADS": "4"})
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and two["config"]["shared_gpu"] is True
    assert two["scaling"] == "weak" and two["steps"] == 2 and two["value"] > 0
    per_rank = two["host"]["per_rank"]
    assert len(per_rank) == 2 and all(r["step_ms"] > 0 and r["emd_ms"] > 0 for r in per_rank)
    if two["host"]["pinned"]:                         # disjoint core ranges of equal size
        assert per_rank[0]["cores"] == per_rank[1]["cores"] >= 1
    one = _bench(["--gpus", "1", "--steps", "4", "--warmup", "2"], {"MARS_EMD_THREADS": "8"})
    assert one["n_gpus"] == 1 and len(one["host"]["per_rank"]) == 1
    assert two["miou"] == one["miou"] and two["fb_iou"] == one["fb_iou"]
