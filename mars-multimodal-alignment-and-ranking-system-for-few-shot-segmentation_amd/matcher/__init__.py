"""The Matcher (proposal producer of the reference) on the ranking path's kernels."""
from .Matcher import Matcher, MatcherProposals, RobustPromptSampler, build_matcher_oss, kmeans_pp  # noqa: F401
