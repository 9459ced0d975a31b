"""Few-shot episode datasets: the reference's batch dictionary (query_img [3, S, S] in [0, 1], query_mask [S, S],
support_imgs [ns, 3, S, S], support_masks [ns, S, S], class_id, names) for FSS-1000
(/root/reference/matcher/data/fss.py), COCO-20i (coco.py), PASCAL-5i (pascal5i.py), LVIS-92i (lvis.py) and seeded synthetic episodes.

The image transform is the reference's ``Resize((S, S))`` + ``ToTensor()`` (dataset.py:32-35) on Pillow images:
bilinear with antialiasing, then /255.  Masks are resized with nearest-neighbour sampling (fss.py:38, coco.py:47).
"""
from __future__ import annotations

import os
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

from .. import synth


class EpisodeTransform:
    def __init__(self, img_size: int):
        self.size = int(img_size)

    def __call__(self, img) -> torch.Tensor:
        from PIL import Image
        im = img.convert("RGB").resize((self.size, self.size), Image.BILINEAR)
        return torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)

    def mask(self, m: torch.Tensor) -> torch.Tensor:
        return F.interpolate(m[None, None].float(), (self.size, self.size), mode="nearest")[0, 0]


def _batch(tr: EpisodeTransform, q_img, q_mask, s_imgs, s_masks, q_name, s_names, class_id: int) -> dict:
    return {"query_img": tr(q_img), "query_mask": tr.mask(q_mask), "query_name": q_name,
            "support_imgs": torch.stack([tr(im) for im in s_imgs]), "support_masks": torch.stack([tr.mask(m) for m in s_masks]),
            "support_names": s_names, "class_id": torch.tensor(class_id)}


class DatasetFSS(torch.utils.data.Dataset):
    """FSS-1000: ``<datapath>/FSS-1000/data/<category>/{1..10}.jpg|.png`` and ``splits/<split>.txt``."""

    def __init__(self, datapath: str, fold: int, transform: EpisodeTransform, split: str, shot: int, use_original_imgsize: bool = False):
        self.split, self.benchmark, self.shot, self.transform = split, "fss", shot, transform
        self.base_path = os.path.join(datapath, "FSS-1000/data")
        with open(os.path.join(datapath, "FSS-1000/splits/%s.txt" % split)) as f:
            self.categories = sorted(f.read().split("\n")[:-1])
        self.class_ids = list({"trn": range(0, 520), "val": range(520, 760), "test": range(760, 1000)}[split])
        self.img_metadata: List[str] = []
        for cat in self.categories:
            names = sorted(n for n in os.listdir(os.path.join(self.base_path, cat)) if n.split(".")[-1] == "jpg")
            self.img_metadata.extend(os.path.join(self.base_path, cat, n) for n in names)

    def __len__(self):
        return len(self.img_metadata)

    @staticmethod
    def read_mask(path: str) -> torch.Tensor:
        from PIL import Image
        return torch.from_numpy((np.asarray(Image.open(path).convert("L")) >= 128).astype(np.uint8))

    def sample_episode(self, idx: int):
        query = self.img_metadata[idx]
        class_id = self.categories.index(query.split("/")[-2]) + {"trn": 0, "val": 520, "test": 760}[self.split]
        support = []
        while len(support) < self.shot:                    # draws with replacement until `shot` non-query names
            name = os.path.join(os.path.dirname(query), str(np.random.choice(range(1, 11), 1, replace=False)[0])) + ".jpg"
            if name != query:
                support.append(name)
        return query, support, class_id

    def __getitem__(self, idx: int) -> dict:
        from PIL import Image
        query, support, class_id = self.sample_episode(idx)
        png = lambda p: os.path.splitext(p)[0] + ".png"    # noqa: E731
        return _batch(self.transform, Image.open(query), self.read_mask(png(query)), [Image.open(s) for s in support],
                      [self.read_mask(png(s)) for s in support], query, support, class_id)


class DatasetCOCO(torch.utils.data.Dataset):
    """COCO-20i: 1000 test episodes drawn uniformly over the fold's 20 classes (coco.py:36-38, 96-130).  The split
    table (class id -> image names) is ``<datapath>/COCO2014/splits/<trn|val>/fold<k>.pkl``, read as plain data."""

    def __init__(self, datapath: str, fold: int, transform: EpisodeTransform, split: str, shot: int, use_original_imgsize: bool = False):
        from ..io import read_plain_pickle
        self.split = "val" if split in ("val", "test") else "trn"
        self.fold, self.nfolds, self.nclass, self.benchmark, self.shot, self.transform = fold, 4, 80, "coco", shot, transform
        self.base_path = os.path.join(datapath, "COCO2014")
        val_ids = [fold + self.nfolds * v for v in range(self.nclass // self.nfolds)]
        self.class_ids = val_ids if self.split == "val" else [c for c in range(self.nclass) if c not in val_ids]
        self.img_metadata_classwise = read_plain_pickle(f"{self.base_path}/splits/{self.split}/fold{fold}.pkl")
        self.img_metadata = sorted({n for names in self.img_metadata_classwise.values() for n in names})

    def __len__(self):
        return 3000 if self.split == "trn" else 1000

    def read_mask(self, name: str) -> torch.Tensor:
        from PIL import Image
        path = os.path.join(self.base_path, "annotations", name)
        return torch.from_numpy(np.asarray(Image.open(path[:path.index(".jpg")] + ".png")).copy())

    def __getitem__(self, idx: int) -> dict:
        from PIL import Image
        c = np.random.choice(self.class_ids, 1, replace=False)[0]
        names = self.img_metadata_classwise[c]
        query = np.random.choice(names, 1, replace=False)[0]
        support = []
        while len(support) < self.shot:
            s = np.random.choice(names, 1, replace=False)[0]
            if s != query:
                support.append(s)
        cut = lambda m: (m == c + 1).to(torch.uint8)       # noqa: E731
        op = lambda n: Image.open(os.path.join(self.base_path, n))   # noqa: E731
        b = _batch(self.transform, op(query), cut(self.read_mask(query)), [op(s) for s in support],
                   [cut(self.read_mask(s)) for s in support], query, support, int(c))
        b["org_query_imsize"] = op(query).size
        return b


class DatasetPASCAL5i(torch.utils.data.Dataset):
    """PASCAL-5i from SBD + VOC2012 (/root/reference/matcher/data/pascal5i.py:18-384).

    ``<datapath>/PASCAL5i/sbd/{train,val}.txt, img/, cls/*.mat`` and ``<datapath>/PASCAL5i/VOCdevkit/VOC2012/
    {ImageSets/Segmentation/{train,val}.txt, JPEGImages, SegmentationClass}``.  Training uses SBD train + val minus
    everything in the VOC validation list, with the VOC annotation where an image is in both (pascal5i.py:62-88);
    every other split is the VOC validation list.  The fold's labels are 1-based ``5 fold + 1 .. 5 fold + 5``
    (``class_ids``, as the reference keeps them); an image belongs to the fold if its label map holds one of
    them.  Test episodes: 1000, query ``idx mod len``, its classes taken in turn (pascal5i.py:242-273, 321-326),
    supports drawn with replacement from the class's images until ``shot`` differ from the query.  The batch
    carries the 255 boundary band as ``query_ignore_idx`` (pascal5i.py:299-316)."""

    def __init__(self, datapath: str, fold: int, transform: EpisodeTransform, split: str, shot: int, use_original_imgsize: bool = False):
        assert 0 <= fold <= 3
        self.nfold, self.nclass, self.benchmark, self.shot, self.transform = 4, 20, "pascal5i", shot, transform
        self.train = split == "train"
        self.split = "trn" if self.train else "val"
        self.use_original_imgsize = use_original_imgsize
        sbd = os.path.join(datapath, "PASCAL5i", "sbd")
        voc = os.path.join(datapath, "PASCAL5i", "VOCdevkit", "VOC2012")
        names = lambda p: [str(n) for n in np.atleast_1d(np.loadtxt(p, dtype="str"))]   # noqa: E731
        voc_trn = names(os.path.join(voc, "ImageSets", "Segmentation", "train.txt"))
        voc_val = names(os.path.join(voc, "ImageSets", "Segmentation", "val.txt"))
        if self.train:
            held = set(voc_val) | set(voc_trn)
            sbd_list = [n for n in names(os.path.join(sbd, "train.txt")) + names(os.path.join(sbd, "val.txt")) if n not in held]
            images = [os.path.join(sbd, "img", n + ".jpg") for n in sbd_list] + [os.path.join(voc, "JPEGImages", n + ".jpg") for n in voc_trn]
            targets = [os.path.join(sbd, "cls", n + ".mat") for n in sbd_list] + [os.path.join(voc, "SegmentationClass", n + ".png") for n in voc_trn]
        else:
            images = [os.path.join(voc, "JPEGImages", n + ".jpg") for n in voc_val]
            targets = [os.path.join(voc, "SegmentationClass", n + ".png") for n in voc_val]
        self.val_label_set = list(range(fold * 5 + 1, fold * 5 + 6))
        self.train_label_set = [i for i in range(1, 21) if i not in self.val_label_set]
        self.label_set = self.class_ids = self.train_label_set if self.train else self.val_label_set
        # image <-> class tables of the fold; cached beside the data under the reference's file name and keys
        cache = os.path.join(voc, f"dataset_{fold}_{self.train}.pt")
        if os.path.exists(cache):
            d = torch.load(cache, weights_only=True)         # plain containers only; nothing in the file is executed
            self.img_class_map, self.class_img_map = {int(k): list(v) for k, v in d["icm"].items()}, {int(k): list(v) for k, v in d["cim"].items()}
            self.images, self.targets = list(d["fi"]), list(d["ft"])
        else:
            self.images, self.targets = [], []
            self.class_img_map = {x: [] for x in self.label_set}
            self.img_class_map = {}
            for img, tgt in zip(images, targets):
                present = set(np.unique(self.load_seg_mask(tgt)).tolist())
                here = [x for x in self.label_set if x in present]
                if here:
                    self.images.append(img)
                    self.targets.append(tgt)
                    k = len(self.images) - 1
                    self.img_class_map[k] = here
                    for x in here:
                        self.class_img_map[x].append(k)
            try:
                torch.save({"icm": self.img_class_map, "cim": self.class_img_map, "fi": self.images, "ft": self.targets}, cache)
            except OSError:
                pass                                         # read-only data directory: rebuild next time

    @staticmethod
    def load_seg_mask(path: str) -> np.ndarray:
        """Label map in 0..20 with the 255 boundary band (pascal5i.py:167-188); SBD keeps it in a MATLAB file."""
        from PIL import Image
        if path.endswith(".mat"):
            from scipy.io import loadmat
            return np.array(loadmat(path)["GTcls"][0]["Segmentation"][0], dtype=np.int_)
        return np.array(Image.open(path), dtype=np.int_)

    def __len__(self):
        return len(self.images) if self.train else 1000

    def sample_episode(self, idx: int, offset: int = 0):
        query = self.images[idx]
        classes = self.img_class_map[idx]
        c = classes[offset % len(classes)]
        support = []
        while len(support) < self.shot:
            s = self.images[np.random.choice(self.class_img_map[c], 1, replace=False)[0]]
            if s != query:
                support.append(s)
        return query, support, c

    @staticmethod
    def extract_binary_mask(mask: torch.Tensor, class_id: int):
        boundary = (mask / 255).floor()
        return (mask == class_id + 1).to(mask.dtype), boundary

    def __getitem__(self, idx: int) -> dict:
        from PIL import Image
        idx %= len(self.images)
        offset = idx // len(self.images)                     # 0 after the modulo, as in the reference (pascal5i.py:325-326)
        query, support, c = self.sample_episode(idx, offset)
        c -= 1
        tr = self.transform
        lab = lambda n: torch.from_numpy(self.load_seg_mask(self.targets[self.images.index(n)]))   # noqa: E731
        q_pil = Image.open(query).convert("RGB")
        q_lab = lab(query).float()
        if not self.use_original_imgsize:
            q_lab = tr.mask(q_lab)
        q_mask, q_ign = self.extract_binary_mask(q_lab, c)
        s_pairs = [self.extract_binary_mask(tr.mask(lab(s)), c) for s in support]
        return {"query_img": tr(q_pil), "query_mask": q_mask, "query_name": query, "query_ignore_idx": q_ign,
                "org_query_imgsize": q_pil.size, "support_imgs": torch.stack([tr(Image.open(s)) for s in support]),
                "support_masks": torch.stack([m for m, _ in s_pairs]), "support_names": support,
                "support_ignore_idxs": torch.stack([g for _, g in s_pairs]), "class_id": torch.tensor(c)}


def polygon_to_mask(xy, h: int, w: int) -> np.ndarray:
    """One polygon (x0, y0, x1, y1, ...) -> [h, w] bool, the COCO API's ``rleFrPoly`` restated (the reference reaches
    it through detectron2's ``polygons_to_bitmask`` -> pycocotools ``frPyObjects``, /root/reference/matcher/data/
    lvis.py:99-105): the outline is drawn at five times the resolution with one point per step along the longer axis
    of every edge, the points where the outline changes column give the crossings of the pixel columns, and the
    column-major run lengths between sorted crossings alternate outside / inside."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    k = xy.shape[0]
    scale = 5.0
    x = np.floor(scale * xy[:, 0] + 0.5).astype(np.int64)
    y = np.floor(scale * xy[:, 1] + 0.5).astype(np.int64)
    x, y = np.append(x, x[0]), np.append(y, y[0])
    us, vs = [], []
    for j in range(k):
        xs, xe, ys, ye = int(x[j]), int(x[j + 1]), int(y[j]), int(y[j + 1])
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            sl = (ye - ys) / dx if dx else 0.0
            d = np.arange(dx + 1)
            t = dx - d if flip else d
            us.append(t + xs)
            vs.append(np.floor(ys + sl * t + 0.5).astype(np.int64))
        else:
            sl = (xe - xs) / dy
            d = np.arange(dy + 1)
            t = dy - d if flip else d
            vs.append(t + ys)
            us.append(np.floor(xs + sl * t + 0.5).astype(np.int64))
    u, v = np.concatenate(us), np.concatenate(vs)
    # crossings of the pixel columns, downsampled
    ch = np.flatnonzero(u[1:] != u[:-1]) + 1
    uj, up, vj, vp = u[ch], u[ch - 1], v[ch], v[ch - 1]
    xd = (np.where(uj < up, uj, uj - 1).astype(np.float64) + 0.5) / scale - 0.5
    ok = (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
    yd = (np.minimum(vj, vp).astype(np.float64) + 0.5) / scale - 0.5
    yd = np.ceil(np.clip(yd, 0, h))
    a = np.sort((xd[ok].astype(np.int64) * h + yd[ok].astype(np.int64)))
    a = np.append(a, h * w)
    runs = np.diff(np.concatenate([[0], a]))
    # zero-length runs join their neighbours (two crossings at one place cancel)
    b = []
    j = 0
    b.append(int(runs[j])); j += 1
    while j < len(runs):
        if runs[j] > 0:
            b.append(int(runs[j])); j += 1
        else:
            j += 1
            if j < len(runs):
                b[-1] += int(runs[j]); j += 1
    flat = np.zeros(h * w, dtype=bool)
    pos, val = 0, False
    for c in b:
        if val:
            flat[pos:pos + c] = True
        pos += c
        val = not val
    return flat.reshape(w, h).T


def segmentation_to_mask(segm, image_size) -> torch.Tensor:
    """lvis.py:94-115: polygons (union), a COCO run-length dict (compressed string or counts list), or a mask array;
    ``image_size`` is Pillow's (width, height)."""
    from ..segment_anything.amg import coco_decode_rle, rle_to_mask
    w, h = image_size
    if isinstance(segm, list):
        m = np.zeros((h, w), dtype=bool)
        for poly in segm:
            m |= polygon_to_mask(poly, h, w)
        return torch.from_numpy(m.astype(np.uint8))
    if isinstance(segm, dict):
        counts = segm["counts"]
        if isinstance(counts, bytes):
            counts = counts.decode("ascii")
        rle = coco_decode_rle({"size": segm["size"], "counts": counts}) if isinstance(counts, str) else segm
        return torch.from_numpy(rle_to_mask(rle).astype(np.uint8))
    arr = np.asarray(segm)
    assert arr.ndim == 2, f"Expect segmentation of 2 dimensions, got {arr.ndim}."
    return torch.from_numpy(arr.astype(np.uint8))


class DatasetLVIS(torch.utils.data.Dataset):
    """LVIS-92i (/root/reference/matcher/data/lvis.py:15-189): ``<datapath>/LVIS/lvis_{train,val}.pkl`` hold
    {category id: {image name: {"annotations": [{"segmentation": ...}]}}} and the images live under
    ``<datapath>/LVIS/coco``.  Validation categories are those with more than ``shot`` images, dealt to ten folds
    (``val_cat_ids[fold + 10 v]``); 2300 test episodes walk the fold's classes in turn, query and supports drawn from
    the class's images; a mask is the union of the image's annotations of the class.  The tables are read as plain
    data (nothing from the pickles is executed)."""

    def __init__(self, datapath: str, fold: int, transform: EpisodeTransform, split: str, shot: int, use_original_imgsize: bool = False):
        from ..io import read_plain_pickle
        self.split = "val" if split in ("val", "test") else "trn"
        self.fold, self.nfolds, self.benchmark, self.shot, self.transform = fold, 10, "lvis", shot, transform
        self.use_original_imgsize = use_original_imgsize
        self.anno_path = os.path.join(datapath, "LVIS")
        self.base_path = os.path.join(datapath, "LVIS", "coco")
        train = read_plain_pickle(os.path.join(self.anno_path, "lvis_train.pkl"), max_bytes=1 << 34)
        val = read_plain_pickle(os.path.join(self.anno_path, "lvis_val.pkl"), max_bytes=1 << 34)
        train_ids = list(train.keys())
        val_ids = [c for c in val.keys() if len(val[c]) > shot]
        ids_val = [val_ids[fold + self.nfolds * v] for v in range(len(val_ids) // self.nfolds)]
        ids_trn = [c for c in train_ids if c not in ids_val]
        self.class_ids_ori = ids_trn if self.split == "trn" else ids_val
        self.nclass = len(train_ids) if self.split == "trn" else len(val_ids)
        self.img_metadata_classwise = train if self.split == "trn" else val
        self.class_ids_c = {cid: i for i, cid in enumerate(self.class_ids_ori)}
        self.class_ids = sorted(self.class_ids_c.values())
        self.img_metadata = sorted({n for names in self.img_metadata_classwise.values() for n in names})

    def __len__(self):
        return len(self.img_metadata) if self.split == "trn" else 2300

    def _mask(self, info: dict, size) -> torch.Tensor:
        parts = [segmentation_to_mask(a["segmentation"], size)[None].float() for a in info["annotations"]]
        return (torch.cat(parts, dim=0).sum(0) > 0).to(torch.uint8)

    def __getitem__(self, idx: int) -> dict:
        from PIL import Image
        idx %= len(self.class_ids)
        c = self.class_ids_ori[idx]
        table = self.img_metadata_classwise[c]
        names = list(table.keys())
        query = np.random.choice(names, 1, replace=False)[0]
        q_img = Image.open(os.path.join(self.base_path, query)).convert("RGB")
        q_mask = self._mask(table[query], q_img.size)
        support = []
        while len(support) < self.shot:
            s = np.random.choice(names, 1, replace=False)[0]
            if s != query:
                support.append(s)
        s_imgs = [Image.open(os.path.join(self.base_path, s)).convert("RGB") for s in support]
        s_masks = [self._mask(table[s], im.size) for s, im in zip(support, s_imgs)]
        b = _batch(self.transform, q_img, q_mask, s_imgs, s_masks, str(query), [str(s) for s in support], self.class_ids_c[c])
        b["org_query_imsize"] = q_img.size
        return b


class DatasetSynthetic(torch.utils.data.Dataset):
    """Seeded synthetic episodes of the benchmark's shapes (no files), with stored-proposal tensors."""

    def __init__(self, n_episodes: int, img_size: int, shot: int, n_proposals: int, fold: int = 0):
        self.n, self.size, self.shot, self.nprop, self.fold = n_episodes, img_size, shot, n_proposals, fold
        self.benchmark, self.class_ids = "coco", [fold + 4 * i for i in range(20)]

    def __len__(self):
        return self.n

    def __getitem__(self, idx: int) -> dict:
        ep = synth.make_episode(1000 * self.fold + idx, size=self.size, nshot=self.shot, nprop=self.nprop, fold=self.fold)
        return {"query_img": ep["query_img"][0], "query_mask": ep["query_mask"][0], "query_name": f"synthetic_{idx}",
                "support_imgs": ep["support_imgs"][0], "support_masks": ep["support_masks"][0], "support_names": [],
                "class_id": torch.as_tensor(ep["class_id"]).reshape(()), "mask_proposals": ep["mask_proposals"]}
