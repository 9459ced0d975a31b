"""The reference's own COCO-20i split tables (/root/reference/datasets/COCO2014/splits/{trn,val}/fold{0-3}.pkl:
the only data files the reference holds) through the no-code pickle reader, and ``DatasetCOCO`` on a
table of the same layout.

In the build container the real files are read and compared with tests/golden/coco_split_digest.json
(per-class image counts, first and last name of every class list); ``python tests/test_reference_splits.py``
rewrites that digest from the reference files.  The GPU box has no /root/reference: there the
digest itself is turned back into a table of the same shape and read through the same reader."""
import importlib
import json
import os
import pickle
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import GOLDEN, load_package

REF_SPLITS = "/root/reference/datasets/COCO2014/splits"
DIGEST = os.path.join(GOLDEN, "coco_split_digest.json")


def _io():
    pkg = load_package(require_native=False)
    return importlib.import_module(pkg.__name__ + ".io"), pkg


def _digest_of(table):
    return {str(c): [len(table[c]), table[c][0] if table[c] else "", table[c][-1] if table[c] else ""] for c in sorted(table)}


@pytest.mark.skipif(not os.path.isdir(REF_SPLITS), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("split", ["val", "trn"])
@pytest.mark.parametrize("fold", [0, 1, 2, 3])
def test_reference_split_tables_read_as_plain_data(split, fold):
    io, _ = _io()
    table = io.read_plain_pickle(f"{REF_SPLITS}/{split}/fold{fold}.pkl")
    assert isinstance(table, dict) and sorted(table) == list(range(80))
    assert all(isinstance(v, list) and all(isinstance(n, str) for n in v) for v in table.values())
    val_ids = [fold + 4 * i for i in range(20)]                 # matcher/data/coco.py:72-73
    filled = [c for c in range(80) if table[c]]
    if split == "val":
        assert filled == val_ids                                # a validation table lists the fold's classes only
        assert all(n.startswith("val2014/COCO_val2014_") and n.endswith(".jpg") for c in filled for n in table[c])
    else:
        assert not set(filled) & set(val_ids)                   # a training table never lists the fold's validation classes
    want = json.load(open(DIGEST))[f"{split}/fold{fold}"]
    assert _digest_of(table) == want


def test_dataset_coco_on_a_table_of_the_reference_layout(tmp_path):
    """DatasetCOCO's class ids and image list from a split file with the reference's layout (dict class id ->
    list of names, protocol-2 pickle of plain containers), rebuilt from the committed digest."""
    io, pkg = _io()
    digest = json.load(open(DIGEST))["val/fold0"]
    table = {int(c): ([v[1]] if v[0] == 1 else [v[1], v[2]] if v[0] else []) for c, v in digest.items()}
    root = tmp_path / "COCO2014" / "splits" / "val"
    root.mkdir(parents=True)
    with open(root / "fold0.pkl", "wb") as f:
        pickle.dump(table, f, protocol=2)
    data = importlib.import_module(pkg.__name__ + ".data.episodes")
    ds = data.DatasetCOCO(str(tmp_path), fold=0, transform=None, split="test", shot=1)
    assert ds.class_ids == [4 * i for i in range(20)] and ds.benchmark == "coco" and len(ds) == 1000
    assert all(ds.img_metadata_classwise[c] for c in ds.class_ids)
    assert ds.img_metadata == sorted({n for v in table.values() for n in v})


if __name__ == "__main__":
    io, _ = _io()
    out = {}
    for split in ("val", "trn"):
        for fold in range(4):
            out[f"{split}/fold{fold}"] = _digest_of(io.read_plain_pickle(f"{REF_SPLITS}/{split}/fold{fold}.pkl"))
    with open(DIGEST, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", DIGEST, os.path.getsize(DIGEST), "bytes")
