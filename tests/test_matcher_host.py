"""Host pieces of the Matcher prompt-sampling stage: the assignment solver against scipy (the solver the
reference calls), prompt-set sampling against the oracle's line-by-line restatement under equal seeds."""
import importlib
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
from oracle import matcher as omatch  # noqa: E402


@pytest.fixture(scope="module")
def mods():
    pkg = load_package(require_native=False)
    return importlib.import_module(pkg.__name__ + ".ops"), importlib.import_module(pkg.__name__ + ".matcher.Matcher")


@pytest.mark.parametrize("n,m", [(1, 1), (1, 9), (9, 1), (6, 6), (17, 40), (40, 17), (120, 333), (333, 120), (256, 256)])
def test_assignment_equals_scipy(mods, n, m):
    from scipy.optimize import linear_sum_assignment
    ops, _ = mods
    g = torch.Generator().manual_seed(n * 1000 + m)
    for maximize in (True, False):
        w = torch.rand(n, m, generator=g) * 2 - 1
        rows, cols = ops.lap_host(w, maximize)
        r, c = linear_sum_assignment(w.numpy(), maximize=maximize)
        assert np.array_equal(rows.numpy(), r) and np.array_equal(cols.numpy(), c)
    # a strided view (rows of a wider matrix) and cosine-like values
    wide = torch.rand(n, m + 5, generator=g)
    rows, cols = ops.lap_host(wide[:, 2:2 + m], True)
    r, c = linear_sum_assignment(wide[:, 2:2 + m].numpy(), maximize=True)
    assert np.array_equal(rows.numpy(), r) and np.array_equal(cols.numpy(), c)


def test_assignment_rejects_non_finite(mods):
    ops, _ = mods
    w = torch.rand(4, 6)
    w[2, :] = float("nan")
    with pytest.raises(RuntimeError):
        ops.lap_host(w, True)


@pytest.mark.parametrize("npts,nneg,rng_,iters", [(5, 0, (4, 6), 30), (8, 3, (1, 3), 30), (20, 0, (4, 6), 7), (20, 30, (2, 3), 5),
                                                   (20, 4, (4, 6), 6), (3, 0, (4, 6), 30)])
def test_sample_points_follow_the_reference_draws(mods, npts, nneg, rng_, iters):
    _, M = mods
    g = np.random.default_rng(npts * 10 + nneg)
    points = g.integers(0, 500, size=(npts, 2))
    negs = g.integers(0, 500, size=(nneg, 2)) if nneg else None
    rps = M.RobustPromptSampler(9, rng_, iters)
    random.seed(123)
    got_s, got_l = rps.sample_points(points, negs)
    random.seed(123)
    want_s, want_l = omatch.sample_points(points, negs, rng_, iters)
    assert len(got_s) == len(want_s) >= 1
    for a, b, la, lb in zip(got_s, want_s, got_l, want_l):
        assert np.array_equal(a, b) and np.array_equal(la, lb)
        assert a.shape[:2] == la.shape and a.shape[2] == 2
    assert M.RobustPromptSampler.combinations(5, 3) == omatch._combinations(5, 3)
    assert M.RobustPromptSampler.combinations(3, 4) == [] and M.RobustPromptSampler.combinations(4, 0) == [[]]


def test_kmeans_gives_k_populated_clusters(mods):
    _, M = mods
    torch.manual_seed(0)
    pts = np.concatenate([np.random.default_rng(1).normal(c, 3.0, size=(15, 2)) for c in ((0, 0), (80, 10), (40, 90))])
    centers, assign = M.kmeans_pp(pts, 3)
    assert centers.shape == (3, 2) and set(assign.tolist()) == {0, 1, 2}
    # each blob is one cluster
    for b in range(3):
        assert len(set(assign[b * 15:(b + 1) * 15].tolist())) == 1
