"""Dataloader builder (/root/reference/matcher/data/dataset.py:14-49)."""
from __future__ import annotations

from torch.utils.data import DataLoader

from .episodes import DatasetCOCO, DatasetFSS, DatasetLVIS, DatasetPASCAL5i, EpisodeTransform


class FSSDataset:
    datasets = {"coco": DatasetCOCO, "fss": DatasetFSS, "pascal5i": DatasetPASCAL5i, "lvis": DatasetLVIS}

    @classmethod
    def initialize(cls, img_size, datapath, use_original_imgsize):
        cls.datapath, cls.use_original_imgsize = datapath, use_original_imgsize
        cls.transform = EpisodeTransform(img_size)

    @classmethod
    def build_dataloader(cls, benchmark, bsz, nworker, fold, split, shot=1):
        if benchmark not in cls.datasets:
            raise NotImplementedError(f"benchmark {benchmark!r}: readers exist for {sorted(cls.datasets)}")
        dataset = cls.datasets[benchmark](cls.datapath, fold=fold, transform=cls.transform, split=split, shot=shot,
                                          use_original_imgsize=cls.use_original_imgsize)
        return DataLoader(dataset, batch_size=bsz, shuffle=split == "trn", num_workers=nworker if split == "trn" else 0)
