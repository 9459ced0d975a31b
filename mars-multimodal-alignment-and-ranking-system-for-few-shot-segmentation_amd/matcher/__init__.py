"""Matcher pieces that share the ranking path's kernels (features, similarity, per-mask scores)."""
from .Matcher import Matcher, RobustPromptSampler, build_matcher_oss  # noqa: F401
