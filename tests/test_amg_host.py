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


def _remove_small_regions_scipy(mask, area_thresh, mode):
    """utils/amg.py:274-298 with scipy's 8-connected labelling in place of cv2.connectedComponentsWithStats
    (both number the components in raster order of their first pixel)."""
    from scipy import ndimage
    correct_holes = mode == "holes"
    working = (correct_holes ^ mask).astype(np.uint8)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), int))
    sizes = np.bincount(regions.reshape(-1), minlength=n + 1)[1:]
    small = [i + 1 for i, s in enumerate(sizes) if s < area_thresh]
    if not small:
        return mask, False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if not fill:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def test_remove_small_regions(amg):
    """The library's host routine against the reference function restated on scipy labelling: blobs, holes, diagonal
    (8-connected) contacts, all-small masks (the largest stays), empty and full masks, thresholds around the sizes."""
    g = np.random.default_rng(3)
    cases = [g.random((40, 50)) > p for p in (0.3, 0.5, 0.7, 0.9)]
    blob = np.zeros((30, 30), bool); blob[2:12, 2:12] = True; blob[5:7, 5:7] = False; blob[20, 20] = True; blob[21, 21] = True
    cases += [blob, np.zeros((6, 9), bool), np.ones((6, 9), bool), np.eye(7, dtype=bool)]
    tiny = np.zeros((8, 8), bool); tiny[1, 1] = True; tiny[5:7, 5] = True; tiny[3, 6] = True
    cases.append(tiny)
    for mask in cases:
        for mode in ("holes", "islands"):
            for thr in (1, 2, 3, 5, 20, 1000):
                got, ch = amg.remove_small_regions(mask, thr, mode)
                want, wch = _remove_small_regions_scipy(mask, thr, mode)
                assert ch == wch and np.array_equal(got, want), (mode, thr)
    # every island below the threshold: only the largest, the first of equals, stays
    got, ch = amg.remove_small_regions(tiny, 10, "islands")
    assert ch and got.sum() == 2 and got[5, 5] and got[6, 5]


def test_mask_boxes_and_coco_rle(amg):
    g = np.random.default_rng(4)
    masks = np.zeros((3, 12, 17), bool)
    masks[0, 3:7, 2:9] = True
    masks[2, 11, 16] = True
    assert amg.batched_mask_to_box(masks).tolist() == [[2, 3, 8, 6], [0, 0, 0, 0], [16, 11, 16, 11]]
    assert np.array_equal(amg.batched_mask_to_box(masks), osam.mask_to_box(torch.as_tensor(masks)).float().numpy())
    # compressed run lengths: values by hand from the 5-bit group code
    assert amg.coco_encode_rle({"size": [1, 5], "counts": [5]})["counts"] == "5"
    assert amg.coco_encode_rle({"size": [5, 8], "counts": [40]})["counts"] == "X1"
    assert amg.coco_encode_rle({"size": [4, 5], "counts": [3, 10, 3, 4]})["counts"] == "3:3J"
    for mask in [g.random((23, 31)) > p for p in (0.1, 0.5, 0.97)] + [np.zeros((300, 400), bool), np.ones((300, 400), bool)]:
        rle = amg.mask_to_rle(mask)
        enc = amg.coco_encode_rle(rle)
        assert isinstance(enc["counts"], str) and all(48 <= ord(c) < 112 for c in enc["counts"])
        assert amg.coco_decode_rle(enc) == rle
