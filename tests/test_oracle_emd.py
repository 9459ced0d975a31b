"""Exact EMD: the oracle's C solver against scipy HiGHS, and the shipped host solver (network
simplex in libmars_hip.so) against the oracle.  POT itself is absent (parity unpinned against the
wheel); the LP optimum is unique, so two independent exact solvers pin the value."""
import ctypes

import numpy as np
import pytest

from oracle.emd import emd2_highs, emd2_uniform


@pytest.mark.parametrize("n,m", [(1, 1), (1, 9), (6, 1), (4, 4), (17, 5), (30, 45), (64, 64)])
def test_oracle_ssp_matches_highs(n, m):
    rng = np.random.default_rng(n * 100 + m)
    M = rng.random((n, m))
    assert abs(emd2_uniform(M) - emd2_highs(M)) < 1e-10


def test_oracle_known_answers():
    assert emd2_uniform(np.zeros((3, 4))) == 0.0
    assert abs(emd2_uniform(np.full((5, 2), 0.25)) - 0.25) < 1e-15
    # identity assignment: zero diagonal, unit off-diagonal
    assert abs(emd2_uniform(1.0 - np.eye(6))) < 1e-15
    # 2 sources, 1 sink: mean of the column
    assert abs(emd2_uniform(np.array([[0.2], [0.6]])) - 0.4) < 1e-15


@pytest.mark.parametrize("n,m", [(1, 1), (1, 9), (6, 1), (2, 2), (16, 16), (17, 5), (30, 45), (64, 64), (200, 120), (333, 333)])
def test_product_simplex_matches_oracle(pkg, n, m):
    lib = pkg.hip.lib()
    rng = np.random.default_rng(7 * n + m)
    M = np.ascontiguousarray(rng.random((n, m)))
    got = lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), n, m)
    assert abs(got - emd2_uniform(M)) < 1e-10


def test_product_simplex_degenerate_costs(pkg):
    """Ties everywhere (quantised costs, equal marginals): the perturbed simplex must still stop at
    the optimum."""
    lib = pkg.hip.lib()
    rng = np.random.default_rng(5)
    for n, m in ((40, 40), (60, 30), (25, 100)):
        M = np.ascontiguousarray(np.round(rng.random((n, m)), 1))
        got = lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), n, m)
        assert abs(got - emd2_uniform(M)) < 1e-10
    M = np.zeros((50, 50))
    assert lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), 50, 50) == 0.0


def test_product_emd_scores_batch(pkg):
    """marshost_emd_scores: gather of C = (1-S)/2 (float32) by support rows and mask columns, threads."""
    import torch
    ops = pkg.ops
    rng = np.random.default_rng(1)
    R, P, N = 50, 40, 7
    S = torch.from_numpy(rng.uniform(-1, 1, (R, 64)).astype(np.float32))
    fg = torch.from_numpy((rng.random(R) < 0.4).astype(np.uint8))
    bits = torch.from_numpy((rng.random((N, P)) < 0.3).astype(np.uint8))
    bits[:, 0] = 1
    got = ops.emd_scores_host(S, fg, bits, threads=3)
    C = ((1 - S[:, :P]) / 2).numpy()
    for k in range(N):
        M = C[fg.numpy().astype(bool)][:, bits[k].numpy().astype(bool)].astype(np.float64)
        assert abs(float(got[k]) - (1 - emd2_uniform(M))) < 1e-10
    got2 = ops.emd_scores_host(torch.from_numpy(C).contiguous(), fg, bits, threads=1, is_cost=True)
    assert float((got - got2).abs().max()) < 1e-12


def test_product_emd_scores_float_costs_and_vector_tails(pkg):
    """The batch entry prices float32 costs with a vector minimum over whole rows: every tail length of that
    loop (numbers of sinks from 1 to 11, and around 128) and costs with the structure of the path's own (cosines
    of correlated features) against the double-precision single-problem entry and the oracle's solver."""
    import torch
    ops, lib = pkg.ops, pkg.hip.lib()
    rng = np.random.default_rng(11)
    R, P = 90, 140
    f_s = rng.normal(size=(R, 24)) + 2.0 * rng.normal(size=(1, 24))
    f_q = rng.normal(size=(P, 24)) + 2.0 * rng.normal(size=(1, 24))
    f_s /= np.linalg.norm(f_s, axis=1, keepdims=True)
    f_q /= np.linalg.norm(f_q, axis=1, keepdims=True)
    S = torch.from_numpy(np.pad((f_s @ f_q.T).astype(np.float32), ((0, 0), (0, 192 - P))))
    fg = torch.from_numpy((rng.random(R) < 0.6).astype(np.uint8))
    sizes = list(range(1, 12)) + [127, 128, 129, 131, P]
    bits = torch.zeros(len(sizes), P, dtype=torch.uint8)
    for k, m in enumerate(sizes):
        bits[k, torch.from_numpy(rng.permutation(P)[:m].copy())] = 1
    got = ops.emd_scores_host(S, fg, bits, threads=4)
    C = ((1 - S[:, :P]) / 2).numpy()
    for k, m in enumerate(sizes):
        M = np.ascontiguousarray(C[fg.numpy().astype(bool)][:, bits[k].numpy().astype(bool)].astype(np.float64))
        one = lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), M.shape[0], M.shape[1])
        assert abs(float(got[k]) - (1 - one)) < 1e-12, m
        assert abs(one - emd2_uniform(M)) < 1e-10, m


def test_product_simplex_sse2_pricing_loop():
    """The pricing loop exists twice (AVX2 where the CPU has it, SSE2 otherwise): the SSE2 form, forced through
    the environment in a process of its own, gives the same optima."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from __graft_entry__ import load_package\n"
        "from oracle.emd import emd2_uniform\n"
        "lib = load_package(require_native=True).hip.lib()\n"
        "rng = np.random.default_rng(2)\n"
        "for n, m in ((5, 1), (7, 3), (40, 37), (64, 130)):\n"
        "    M = np.ascontiguousarray(rng.random((n, m)))\n"
        "    got = lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), n, m)\n"
        "    assert abs(got - emd2_uniform(M)) < 1e-10, (n, m)\n"
        "print('ok')\n" % root)
    env = dict(os.environ, MARS_EMD_NO_AVX2="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_product_simplex_rejects_non_finite_costs(pkg):
    """A NaN or infinite cost has no transport optimum: the answer is NaN, promptly, instead of a
    pivoting sequence on comparisons that no longer order anything."""
    lib = pkg.hip.lib()
    rng = np.random.default_rng(3)
    for bad in (np.nan, np.inf, -np.inf):
        M = np.ascontiguousarray(rng.random((30, 40)))
        M[7, 11] = bad
        assert np.isnan(lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), 30, 40))
    M = np.full((12, 9), np.nan)
    assert np.isnan(lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), 12, 9))


def test_product_simplex_many_small_shapes(pkg):
    """The basis tree is kept as a pre-order thread that is cut and re-strung at every pivot, with childless rows
    kept out of it: small problems of every shape (either side the larger one, single rows and columns, ties)
    walk through all of its corner cases.  Against the oracle's successive-shortest-path solver."""
    lib = pkg.hip.lib()
    rng = np.random.default_rng(99)
    for trial in range(150):
        n, m = int(rng.integers(1, 28)), int(rng.integers(1, 28))
        M = rng.random((n, m))
        if trial % 3 == 0:
            M = np.round(M, 1)                        # many equal costs
        if trial % 5 == 0:
            M = np.abs(rng.normal(size=(n, 1)) - rng.normal(size=(1, m)))      # costs of points on a line
        M = np.ascontiguousarray(M)
        got = lib.marshost_emd2_uniform(ctypes.c_void_p(M.ctypes.data), n, m)
        assert abs(got - emd2_uniform(M)) < 1e-10, (trial, n, m)


def test_product_emd_scores_exact_pricing_switch():
    """Entering arcs of the batch entry are selected by a single-precision scan; MARS_EMD_EXACT_PRICING=1 selects
    them in double.  Both reach the same optimum (the scan only chooses among improving arcs)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from __graft_entry__ import load_package\n"
        "ops = load_package(require_native=True).ops\n"
        "rng = np.random.default_rng(4)\n"
        "S = torch.from_numpy(rng.uniform(-1, 1, (120, 128)).astype(np.float32))\n"
        "fg = torch.from_numpy((rng.random(120) < 0.7).astype(np.uint8))\n"
        "bits = torch.from_numpy((rng.random((9, 100)) < 0.4).astype(np.uint8)); bits[:, 0] = 1\n"
        "print(' '.join(repr(float(x)) for x in ops.emd_scores_host(S, fg, bits, threads=2)))\n" % root)
    outs = []
    for extra in ({}, {"MARS_EMD_EXACT_PRICING": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([float(x) for x in r.stdout.split()])
    assert len(outs[0]) == 9 and max(abs(a - b) for a, b in zip(*outs)) < 1e-12
