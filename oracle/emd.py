"""Exact EMD with uniform marginals for the oracle (test infrastructure only).

``emd2_uniform`` is the value POT's ``ot.emd2(1/n, 1/m, M)`` returns at the call sites
/root/reference/mars/components/FilteringMergingModule.py:160-166 and
/root/reference/matcher/Matcher.py:1188-1193.  POT is absent (parity unpinned against
the wheel); the LP optimum is unique, and two independent exact solvers are kept here:
the C successive-shortest-path solver in ``emd_ssp.c`` and scipy's HiGHS.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = os.path.join(_BUILD, "libemd_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "emd_ssp.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _LIB, src, "-lm"])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
        _lib.emd_uniform_ssp.restype = ctypes.c_double
        _lib.emd_uniform_ssp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return _lib


def emd2_uniform(M: np.ndarray) -> float:
    M = np.ascontiguousarray(M, dtype=np.float64)
    n, m = M.shape
    return float(_load().emd_uniform_ssp(M.ctypes.data, n, m))


def emd2_highs(M: np.ndarray) -> float:
    """Same LP through scipy.optimize.linprog(method='highs'); small problems only."""
    from scipy.optimize import linprog
    from scipy.sparse import lil_matrix
    M = np.asarray(M, dtype=np.float64)
    n, m = M.shape
    A = lil_matrix((n + m, n * m))
    for i in range(n):
        A[i, i * m:(i + 1) * m] = 1
    for j in range(m):
        A[n + j, j::m] = 1
    b = np.concatenate([np.full(n, 1.0 / n), np.full(m, 1.0 / m)])
    # drop one redundant constraint for numerical cleanliness
    res = linprog(M.reshape(-1), A_eq=A.tocsr()[:-1], b_eq=b[:-1], bounds=(0, None), method="highs")
    assert res.status == 0, res.message
    return float(res.fun)
