"""Proposal hand-off formats on the GPU and the test loop end to end on synthetic episodes."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 518, 518), (5, 17, 9), (1, 1, 31), (2, 8, 8), (4, 33, 64)])
def test_packed_proposals_round_trip(pkg, cuda, tmp_path, shape):
    pio = importlib.import_module(pkg.__name__ + ".io")
    g = torch.Generator().manual_seed(sum(shape))
    m = (torch.rand(*shape, generator=g) > 0.6).float()
    m[0].zero_()
    n, hw = shape[0], shape[1] * shape[2]
    words = torch.full((n, (hw + 31) // 32), -1, dtype=torch.int32, device=cuda)
    pkg.ops.pack_masks(m.to(cuda), words)
    flat = np.zeros((n, ((hw + 31) // 32) * 32), dtype=np.uint8)
    flat[:, :hw] = m.reshape(n, hw).numpy()
    want = np.packbits(flat.reshape(n, -1, 32), axis=-1, bitorder="little").view(np.uint32).reshape(n, -1)
    assert np.array_equal(words.cpu().numpy().view(np.uint32), want)          # bit-exact, padding bits zero
    back = torch.full(shape, 7.0, device=cuda)
    pkg.ops.unpack_masks(words, back)
    assert torch.equal(back.cpu(), m)
    path = str(tmp_path / "0_0.mprop")
    pio.save_packed(path, m.to(cuda) * 3.0)                                   # any positive value is "set"
    assert os.path.getsize(path) == 20 + 4 * n * ((hw + 31) // 32)
    assert torch.equal(pio.load_proposals(path, cuda).cpu(), m)
    pt = str(tmp_path / "0_0.pt")
    torch.save(m.bool(), pt)
    assert torch.equal(pio.load_proposals(pt, cuda).cpu(), m)
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 4)
    with pytest.raises(ValueError):
        pio.load_proposals(path, cuda)


def test_test_loop_on_synthetic_episodes(pkg, cuda, tmp_path):
    """main_MARS.test_MARS: stored proposals (files), in-batch proposals and generated proposals give a result."""
    import main_MARS
    pio = importlib.import_module(pkg.__name__ + ".io")
    base = ["--synthetic", "3", "--input_size", "126", "--proposals", "6", "--dtype", "fp32", "--log_root_path", str(tmp_path)]
    out = main_MARS.test_MARS(main_MARS.build_parser().parse_args(base + ["--exp_name", "inbatch"]))
    assert out["episodes"] == 3 and 0.0 <= out["miou"] <= 100.0 and 0.0 <= out["fb_iou"] <= 100.0
    rows = open(os.path.join(out["log_path"], "ranking_time.csv")).read().strip().split("\n")
    assert rows[0].startswith("idx,full_mars_prediction_time") and len(rows) == 4 and rows[1].endswith(",6")
    # the same episodes with their proposals read from files of both kinds: the same result
    data = importlib.import_module(pkg.__name__ + ".data")
    ds = data.DatasetSynthetic(3, 126, 1, 6, 0)
    pdir = tmp_path / "props"
    os.makedirs(pdir)
    for i in range(3):
        if i == 1:
            pio.save_packed(str(pdir / f"0_{i}.mprop"), ds[i]["mask_proposals"].to(cuda))
        else:
            torch.save(ds[i]["mask_proposals"], str(pdir / f"0_{i}.pt"))

    class NoProposals(data.DatasetSynthetic):
        def __getitem__(self, idx):
            b = super().__getitem__(idx)
            del b["mask_proposals"]
            return b

    orig = data.DatasetSynthetic
    data.DatasetSynthetic = NoProposals
    try:
        out2 = main_MARS.test_MARS(main_MARS.build_parser().parse_args(base + ["--exp_name", "files", "--mask_proposals_path", str(pdir)]))
    finally:
        data.DatasetSynthetic = orig
    assert out2["miou"] == out["miou"] and out2["fb_iou"] == out["fb_iou"]
    out3 = main_MARS.test_MARS(main_MARS.build_parser().parse_args(
        base + ["--exp_name", "generated", "--generate_proposals", "sam", "--sam_size", "tiny", "--max_episodes", "2"]))
    assert out3["episodes"] == 2 and 0.0 <= out3["miou"] <= 100.0
