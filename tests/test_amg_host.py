"""Host helpers of the mask generator against the oracle (itself equal to the reference's utils.amg, see
tests/golden/make_golden.py) and against vectors made by the reference."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
from oracle import sam as osam  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


@pytest.fixture(scope="module")
def amg():
    pkg = load_package(require_native=False)
    return importlib.import_module(pkg.__name__ + ".segment_anything.amg")


def test_crop_boxes_and_grids(amg):
    assert np.array_equal(np.asarray(amg.generate_crop_boxes((100, 126), 2, 512 / 1500)[0]), GOLD["sam_tiny_crop_boxes"])
    for size, layers, ratio in (((100, 126), 2, 512 / 1500), ((480, 640), 1, 0.25), ((333, 77), 3, 0.4), ((64, 64), 0, 0.3)):
        boxes, lidx, ids = amg.generate_crop_boxes(size, layers, ratio)
        ob, ol = osam.generate_crop_boxes(size, layers, ratio)
        assert boxes == ob and lidx == ol and ids == list(range(len(boxes)))
        assert len(boxes) == sum(4 ** i for i in range(layers + 1))
    grids = amg.build_all_layer_point_grids(8, 2, 2)
    assert [len(g) for g in grids] == [64, 16, 4] and np.array_equal(grids[1], osam.build_point_grid(4))


def test_crop_edge_filter(amg):
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(200, 2, generator=g) * 60
    boxes = torch.cat([xy, xy + torch.rand(200, 2, generator=g) * 40], dim=1).round()
    for crop in ([10, 20, 73, 84], [0, 0, 63, 64], [0, 0, 126, 100]):
        assert torch.equal(amg.is_box_near_crop_edge(boxes, crop, [0, 0, 126, 100]), osam.is_box_near_crop_edge(boxes, crop, [0, 0, 126, 100]))
    assert not bool(amg.is_box_near_crop_edge(boxes, [0, 0, 126, 100], [0, 0, 126, 100]).any())      # the whole image filters nothing


def test_run_length_codes(amg):
    m = torch.as_tensor(GOLD["sam_tiny_masks"]).flatten(0, 1) > 0
    assert amg.mask_to_rle(m[0].numpy())["counts"] == GOLD["sam_tiny_rle0_counts"].tolist()
    g = np.random.default_rng(0)
    cases = [m[i].numpy() for i in range(4)] + [np.zeros((5, 7), bool), np.ones((5, 7), bool), g.random((9, 4)) > 0.5]
    for mask in cases:
        rle = amg.mask_to_rle(mask)
        assert rle == osam.mask_to_rle(torch.as_tensor(mask)[None])[0]
        assert np.array_equal(amg.rle_to_mask(rle), mask) and amg.area_from_rle(rle) == int(mask.sum())
