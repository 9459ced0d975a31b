"""File formats and episode readers either side of the ranking path (host side; files are made by the test)."""
import importlib
import os
import pickle
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


@pytest.fixture(scope="module")
def mods():
    pkg = load_package(require_native=False)
    return importlib.import_module(pkg.__name__ + ".io"), importlib.import_module(pkg.__name__ + ".data")


@pytest.mark.parametrize("protocol", [0, 2, 4, 5])
def test_plain_pickle_reader(mods, tmp_path, protocol):
    pio, _ = mods
    obj = {0: ["train2014/a.jpg", "train2014/b.jpg"], 4: [], 8: ["x"] * 300, "meta": (1, 2.5, None, True, b"raw", -2 ** 40),
           "nested": {"s": {1, 2, 3}, "t": ((), [[]])}}
    if protocol < 4:                 # older protocols spell sets and bytes as calls of globals: outside plain data
        obj["meta"] = (1, 2.5, None, True, -2 ** 40)
        obj["nested"] = {"t": ((), [[]])}
    shared = ["same"]
    obj["alias"] = [shared, shared]
    path = tmp_path / "split.pkl"
    path.write_bytes(pickle.dumps(obj, protocol=protocol))
    got = pio.read_plain_pickle(str(path))
    assert got == obj and got["alias"][0] is got["alias"][1]


def test_plain_pickle_reader_refuses_objects(mods, tmp_path):
    pio, _ = mods
    for i, bad in enumerate([np.arange(3), torch.ones(2), os.path.join, complex(1, 2), {"k": np.float32(1)}]):
        path = tmp_path / f"bad{i}.pkl"
        path.write_bytes(pickle.dumps(bad))
        with pytest.raises(ValueError):
            pio.read_plain_pickle(str(path))
    trunc = tmp_path / "trunc.pkl"
    trunc.write_bytes(pickle.dumps({"a": [1, 2, 3]})[:-1])
    with pytest.raises(ValueError):
        pio.read_plain_pickle(str(trunc))


def test_state_dict_reader(mods, tmp_path):
    pio, _ = mods
    sd = {"blocks.0.attn.qkv.weight": torch.randn(6, 2), "cls_token": torch.zeros(1, 1, 2)}
    for name, obj in (("plain", sd), ("wrapped", {"model": sd, "epoch": 3}), ("ddp", {"state_dict": {"module." + k: v for k, v in sd.items()}})):
        p = tmp_path / f"{name}.pth"
        torch.save(obj, p)
        got = pio.load_state_dict(str(p))
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    p = tmp_path / "notsd.pth"
    torch.save({"a": 1}, p)
    with pytest.raises(ValueError):
        pio.load_state_dict(str(p))


class _Leaf(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.lin = torch.nn.Linear(3, 3)
        self.lin.weight.data = torch.randn(3, 3, generator=g)
        self.register_buffer("scale", torch.randn(3, generator=g).half())
        self.register_buffer("steps", torch.arange(3))

    def forward(self, x):
        return self.lin(x) * self.scale.float() + self.steps.float()


class _Tree(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.first = _Leaf(1)
        self.logit_scale = torch.nn.Parameter(torch.tensor(2.5))
        self.proj = torch.nn.Parameter(torch.randn(3, 2, generator=torch.Generator().manual_seed(2)).t().contiguous().t())   # strided
        self.blocks = torch.nn.Sequential(_Leaf(3), _Leaf(4))

    def forward(self, x):
        return self.blocks(self.first(x)) @ self.proj * self.logit_scale


def test_torchscript_archive_tensors_without_running_it(mods, tmp_path):
    """The format OpenAI publishes CLIP in (/root/reference/clip/clip.py:95-150 calls torch.jit.load on it and takes
    state_dict()): scripted and traced archives give the module's state dict, read from the opcode listing."""
    pio, _ = mods
    m = _Tree().eval()
    want = m.state_dict()
    for name, mod in (("scripted", torch.jit.script(m)), ("traced", torch.jit.trace(m, torch.zeros(1, 3)))):
        p = tmp_path / f"{name}.pt"
        torch.jit.save(mod, str(p))
        for got in (pio.read_torchscript_state_dict(str(p)), pio.load_state_dict(str(p))):
            assert set(got) == set(want), (sorted(got), sorted(want))
            for k in want:
                assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k


def test_torchscript_archive_with_a_call_in_it_is_refused(mods, tmp_path):
    """An archive whose data pickle imports or calls anything but the tensor records is refused, and nothing runs."""
    import zipfile
    pio, _ = mods
    marker = tmp_path / "ran"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    good = tmp_path / "good.pt"
    torch.jit.save(torch.jit.script(torch.nn.Linear(2, 2)), str(good))
    cases = {
        "call": pickle.dumps(Evil(), protocol=2),
        "foreign_object": pickle.dumps(pytest.raises, protocol=2),          # a global that is not a scripted class
    }
    for name, payload in cases.items():
        bad = tmp_path / f"{name}.pt"
        with zipfile.ZipFile(good) as zi, zipfile.ZipFile(bad, "w") as zo:
            for item in zi.namelist():
                zo.writestr(item, payload if item.endswith("/data.pkl") else zi.read(item))
        with pytest.raises(ValueError):
            pio.load_state_dict(str(bad))
    assert not marker.exists()
    # a tensor record that reaches beyond its storage: the archive's own records with one storage cut short
    short = tmp_path / "short.pt"
    with zipfile.ZipFile(good) as zi, zipfile.ZipFile(short, "w") as zo:
        for item in zi.namelist():
            zo.writestr(item, zi.read(item)[:4] if item.endswith("/data/0") else zi.read(item))
    with pytest.raises(ValueError):
        pio.read_torchscript_state_dict(str(short))
    assert pio.read_torchscript_state_dict(str(good))["weight"].shape == (2, 2)


def _write_images(root, names, rng, size=(40, 30)):
    from PIL import Image
    for n in names:
        os.makedirs(os.path.dirname(os.path.join(root, n)), exist_ok=True)
        Image.fromarray(rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)).save(os.path.join(root, n))


def test_fss_reader(mods, tmp_path):
    from PIL import Image
    _, data = mods
    rng = np.random.default_rng(0)
    base = tmp_path / "FSS-1000"
    cats = ["zebra_fish", "abacus"]
    os.makedirs(base / "splits")
    (base / "splits" / "test.txt").write_text("\n".join(cats) + "\n")
    for c in cats:
        _write_images(str(base / "data" / c), [f"{i}.jpg" for i in range(1, 11)], rng)
        for i in range(1, 11):
            m = (rng.random((30, 40)) > 0.5).astype(np.uint8) * 255
            Image.fromarray(m).save(base / "data" / c / f"{i}.png")
    data.FSSDataset.initialize(img_size=28, datapath=str(tmp_path), use_original_imgsize=False)
    np.random.seed(0)
    loader = data.FSSDataset.build_dataloader("fss", bsz=1, nworker=0, fold=0, split="test", shot=2)
    assert len(loader.dataset) == 20 and loader.dataset.categories == sorted(cats)
    b = next(iter(loader))
    assert b["query_img"].shape == (1, 3, 28, 28) and b["support_imgs"].shape == (1, 2, 3, 28, 28)
    assert b["query_mask"].shape == (1, 28, 28) and b["support_masks"].shape == (1, 2, 28, 28)
    assert 0.0 <= float(b["query_img"].min()) and float(b["query_img"].max()) <= 1.0
    assert set(b["query_mask"].unique().tolist()) <= {0.0, 1.0} and int(b["class_id"]) == 760      # 'abacus' sorts first
    assert all(s[0] != b["query_name"][0] for s in b["support_names"])
    # the image transform is Pillow's antialiased bilinear resize followed by /255
    im = Image.open(b["query_name"][0]).convert("RGB").resize((28, 28), Image.BILINEAR)
    assert torch.equal(b["query_img"][0], torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 255)
    with pytest.raises(NotImplementedError):
        data.FSSDataset.build_dataloader("pascal_part", 1, 0, 0, "test")


def test_coco_reader(mods, tmp_path):
    from PIL import Image
    _, data = mods
    rng = np.random.default_rng(1)
    base = tmp_path / "COCO2014"
    fold, names_by_class = 1, {}
    for c in [fold + 4 * v for v in range(20)]:
        names = [f"val2014/img_{c}_{i}.jpg" for i in range(3)]
        names_by_class[c] = names
        _write_images(str(base), names, rng)
        for n in names:
            ann = rng.integers(0, 3, size=(30, 40)).astype(np.uint8) * (c + 1)      # 0, c+1, or 2(c+1) (another class)
            os.makedirs(base / "annotations" / "val2014", exist_ok=True)
            Image.fromarray(ann).save(str(base / "annotations" / n).replace(".jpg", ".png"))
    os.makedirs(base / "splits" / "val")
    (base / "splits" / "val" / f"fold{fold}.pkl").write_bytes(pickle.dumps(names_by_class))
    data.FSSDataset.initialize(img_size=28, datapath=str(tmp_path), use_original_imgsize=False)
    np.random.seed(3)
    ds = data.FSSDataset.build_dataloader("coco", 1, 0, fold, "test", shot=1).dataset
    assert len(ds) == 1000 and ds.class_ids == [fold + 4 * v for v in range(20)] and len(ds.img_metadata) == 60
    b = ds[0]
    c = int(b["class_id"])
    assert c in ds.class_ids and b["query_name"] in names_by_class[c] and b["support_names"][0] != b["query_name"]
    ann = np.asarray(Image.open(str(base / "annotations" / b["query_name"]).replace(".jpg", ".png")))
    want = torch.nn.functional.interpolate(torch.from_numpy((ann == c + 1).astype(np.float32))[None, None], (28, 28), mode="nearest")[0, 0]
    assert torch.equal(b["query_mask"], want) and b["org_query_imsize"] == (40, 30)


def test_pascal5i_reader(mods, tmp_path):
    """PASCAL-5i: fold membership by label, VOC validation list for testing, SBD (MATLAB label maps) + VOC for
    training with the overlap rules, the 255 band as query_ignore_idx, 1-based class_ids shifted by the meter."""
    from PIL import Image
    from scipy.io import savemat
    _, data = mods
    rng = np.random.default_rng(5)
    sbd, voc = tmp_path / "PASCAL5i" / "sbd", tmp_path / "PASCAL5i" / "VOCdevkit" / "VOC2012"
    for d in (sbd / "img", sbd / "cls", voc / "JPEGImages", voc / "SegmentationClass", voc / "ImageSets" / "Segmentation"):
        os.makedirs(d)

    def label_map(classes):
        m = np.zeros((30, 40), dtype=np.uint8)
        for k, c in enumerate(classes):
            m[2 + 9 * k: 9 + 9 * k, 5:30] = c
            m[1 + 9 * k, 5:30] = 255                       # boundary band
        return m

    voc_val = {"v0": [6, 7], "v1": [6], "v2": [1], "v3": [7, 12], "v4": [6, 10]}      # fold 1 = labels 6..10
    voc_trn = {"t0": [1, 6], "t1": [2]}
    sbd_trn = {"s0": [3], "s1": [8], "t1": [4], "v1": [5]}                              # t1 / v1 also in VOC
    sbd_val = {"s2": [11, 20]}
    for table, imgdir in ((voc_val, voc / "JPEGImages"), (voc_trn, voc / "JPEGImages"), (sbd_trn, sbd / "img"), (sbd_val, sbd / "img")):
        _write_images(str(imgdir), [n + ".jpg" for n in table], rng)
    for table in (voc_val, voc_trn):
        for n, cs in table.items():
            Image.fromarray(label_map(cs)).save(voc / "SegmentationClass" / (n + ".png"))
    for table in (sbd_trn, sbd_val):
        for n, cs in table.items():
            savemat(str(sbd / "cls" / (n + ".mat")), {"GTcls": {"Segmentation": label_map(cs)}})
    (voc / "ImageSets" / "Segmentation" / "val.txt").write_text("\n".join(voc_val) + "\n")
    (voc / "ImageSets" / "Segmentation" / "train.txt").write_text("\n".join(voc_trn) + "\n")
    (sbd / "train.txt").write_text("\n".join(sbd_trn) + "\n")
    (sbd / "val.txt").write_text("\n".join(sbd_val) + "\n")

    data.FSSDataset.initialize(img_size=28, datapath=str(tmp_path), use_original_imgsize=False)
    np.random.seed(2)
    ds = data.FSSDataset.build_dataloader("pascal5i", 1, 0, 1, "test", shot=1).dataset
    assert len(ds) == 1000 and ds.class_ids == [6, 7, 8, 9, 10]
    assert [os.path.basename(p) for p in ds.images] == ["v0.jpg", "v1.jpg", "v3.jpg", "v4.jpg"]     # v2 holds no fold label
    assert ds.img_class_map == {0: [6, 7], 1: [6], 2: [7], 3: [6, 10]} and ds.class_img_map[6] == [0, 1, 3] and ds.class_img_map[8] == []
    b = ds[4 + 1]                                          # wraps to image 1; the class offset stays 0 after the modulo
    assert os.path.basename(b["query_name"]) == "v1.jpg" and int(b["class_id"]) == 5
    assert os.path.basename(b["support_names"][0]) in ("v0.jpg", "v4.jpg")
    lab = torch.nn.functional.interpolate(torch.from_numpy(label_map([6]).astype(np.float32))[None, None], (28, 28), mode="nearest")[0, 0]
    assert torch.equal(b["query_mask"], (lab == 6).float()) and torch.equal(b["query_ignore_idx"], (lab == 255).float())
    assert b["support_masks"].shape == (1, 28, 28) and b["support_ignore_idxs"].shape == (1, 28, 28) and b["org_query_imgsize"] == (40, 30)
    # the class tables were cached beside the data and are read back as plain containers
    assert os.path.exists(voc / "dataset_1_False.pt")
    ds2 = data.DatasetPASCAL5i(str(tmp_path), 1, data.EpisodeTransform(28), "test", 1)
    assert ds2.images == ds.images and ds2.img_class_map == ds.img_class_map and ds2.class_img_map == ds.class_img_map
    # training: SBD train + val without anything VOC lists, VOC annotation for the rest; labels outside the fold
    trn = data.DatasetPASCAL5i(str(tmp_path), 1, data.EpisodeTransform(28), "train", 1)
    assert [os.path.basename(p) for p in trn.images] == ["s0.jpg", "s2.jpg", "t0.jpg", "t1.jpg"]     # s1 holds label 8 only
    assert trn.targets[0].endswith("s0.mat") and trn.targets[3].endswith("t1.png") and trn.img_class_map[1] == [11, 20]
    # the meter shifts the 1-based labels to the 0-based class_id of the batches
    logger = importlib.import_module(mods[0].__name__.rsplit(".", 1)[0] + ".mars.utils.logger")
    meter = logger.AverageMeter(ds, device="cpu")
    assert meter.class_ids_interest.tolist() == [5, 6, 7, 8, 9] and meter.nclass == 20


def test_polygon_rasteriser(mods):
    """The COCO polygon rule restated: an integer rectangle covers exactly its w x h pixels, areas follow the shoelace
    formula to within the outline, and the result differs from Pillow's polygon fill only along the outline."""
    from PIL import Image, ImageDraw
    ep = importlib.import_module(mods[1].__name__ + ".episodes")
    m = ep.polygon_to_mask([2, 3, 7, 3, 7, 9, 2, 9], 12, 10)
    want = np.zeros((12, 10), bool)
    want[3:9, 2:7] = True
    assert np.array_equal(m, want)
    assert np.array_equal(ep.polygon_to_mask([7, 9, 2, 9, 2, 3, 7, 3], 12, 10), want)            # orientation and start do not matter
    assert ep.polygon_to_mask([-5, -5, 30, -5, 30, 30, -5, 30], 12, 10).all()                   # clipped to the image
    g = np.random.default_rng(0)
    for _ in range(8):
        pts = (g.random((7, 2)) * [60, 40]).round(2)
        c = pts.mean(0)
        pts = pts[np.argsort(np.arctan2(pts[:, 1] - c[1], pts[:, 0] - c[0]))]
        m = ep.polygon_to_mask(pts.reshape(-1), 40, 60)
        x, y = pts[:, 0], pts[:, 1]
        area = 0.5 * abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))
        per = np.hypot(x - np.roll(x, 1), y - np.roll(y, 1)).sum()
        im = Image.new("L", (60, 40), 0)
        ImageDraw.Draw(im).polygon([tuple(p) for p in pts], fill=1)
        assert abs(m.sum() - area) <= 0.5 * per and int((m ^ np.asarray(im).astype(bool)).sum()) <= per


def test_lvis_reader(mods, tmp_path):
    """LVIS-92i: validation categories with more than `shot` images dealt to ten folds, episodes walking the fold's
    classes, masks as unions of polygon / run-length / array annotations."""
    _, data = mods
    amg = importlib.import_module(mods[1].__name__.rsplit(".", 1)[0] + ".segment_anything.amg")
    rng = np.random.default_rng(7)
    base = tmp_path / "LVIS"
    names = [f"val2017/{i:03d}.jpg" for i in range(6)]
    _write_images(str(base / "coco"), names, rng)                       # 40 x 30 images
    rect = lambda x0, y0, x1, y1: [float(v) for v in (x0, y0, x1, y0, x1, y1, x0, y1)]   # noqa: E731
    arr = np.zeros((30, 40), np.uint8)
    arr[20:25, 30:38] = 1
    rle_mask = np.zeros((30, 40), bool)
    rle_mask[1:4, 1:9] = True
    rle = amg.coco_encode_rle(amg.mask_to_rle(rle_mask))
    ann = lambda segs: {"annotations": [{"segmentation": s} for s in segs]}   # noqa: E731
    val = {}
    for k in range(25):                                                 # 25 categories; category 3 has a single image
        cat = 100 + k
        imgs = names[:1] if k == 3 else names[k % 3: k % 3 + 3]
        val[cat] = {n: ann([[rect(5, 6, 15, 16)], rle]) for n in imgs}
    val[100][names[0]] = ann([[rect(5, 6, 15, 16), rect(20, 2, 30, 8)], {"size": [30, 40], "counts": amg.mask_to_rle(rle_mask)["counts"]}])
    train = {100: val[100], 101: val[101], 999: val[102]}
    os.makedirs(base, exist_ok=True)
    (base / "lvis_val.pkl").write_bytes(pickle.dumps(val))
    (base / "lvis_train.pkl").write_bytes(pickle.dumps(train))
    data.FSSDataset.initialize(img_size=28, datapath=str(tmp_path), use_original_imgsize=False)
    np.random.seed(1)
    ds = data.FSSDataset.build_dataloader("lvis", 1, 0, 0, "test", shot=1).dataset
    kept = [c for c in val if c != 103]                                 # more than one image
    assert len(ds) == 2300 and ds.nclass == 24 and ds.class_ids_ori == [kept[0], kept[10]] and ds.class_ids == [0, 1]
    assert ds.img_metadata == sorted(names[:5])
    found = False
    for i in range(6):
        b = ds[2 * i]                                                   # class index 0 = category 100
        assert int(b["class_id"]) == 0 and b["support_names"][0] != b["query_name"] and b["query_img"].shape == (3, 28, 28)
        if b["query_name"] == names[0]:
            full = np.zeros((30, 40), np.float32)
            full[6:16, 5:15] = 1
            full[2:8, 20:30] = 1
            full[1:4, 1:9] = 1
            want = torch.nn.functional.interpolate(torch.from_numpy(full)[None, None], (28, 28), mode="nearest")[0, 0]
            assert torch.equal(b["query_mask"], want) and b["org_query_imsize"] == (40, 30)
            found = True
    assert found
    ep = importlib.import_module(mods[1].__name__ + ".episodes")
    assert torch.equal(ep.segmentation_to_mask(arr, (40, 30)), torch.from_numpy(arr))
    trn = data.DatasetLVIS(str(tmp_path), 0, data.EpisodeTransform(28), "trn", 1)
    assert trn.class_ids_ori == [101, 999] and trn.nclass == 3          # category 100 is a validation class of fold 0


def test_synthetic_dataset(mods):
    _, data = mods
    ds = data.DatasetSynthetic(3, 126, 2, 5, fold=2)
    b = ds[1]
    assert b["query_img"].shape == (3, 126, 126) and b["support_imgs"].shape == (2, 3, 126, 126)
    assert b["mask_proposals"].shape == (5, 126, 126) and int(b["class_id"]) in ds.class_ids
    assert torch.equal(ds[1]["query_img"], b["query_img"])
