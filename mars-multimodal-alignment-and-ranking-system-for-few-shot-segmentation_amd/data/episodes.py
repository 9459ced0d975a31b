"""Few-shot episode datasets: the reference's batch dictionary (query_img [3, S, S] in [0, 1], query_mask [S, S],
support_imgs [ns, 3, S, S], support_masks [ns, S, S], class_id, names) for FSS-1000
(/root/reference/matcher/data/fss.py), COCO-20i (coco.py) and seeded synthetic episodes.

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
