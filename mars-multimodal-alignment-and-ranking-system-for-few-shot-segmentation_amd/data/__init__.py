"""Episode readers with the reference's batch layout (/root/reference/matcher/data)."""
from .dataset import FSSDataset  # noqa: F401
from .episodes import DatasetCOCO, DatasetFSS, DatasetLVIS, DatasetPASCAL5i, DatasetSynthetic, EpisodeTransform  # noqa: F401
