"""Times the exact transport solver (marshost_emd_scores) on dumped benchmark cases (tools/dump_emd_cases.py):
core-milliseconds on one thread, wall time on T threads, and the largest deviation from the stored scores."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("MARS_LIB", os.path.join(ROOT, "mars-multimodal-alignment-and-ranking-system-for-few-shot-segmentation_amd", "csrc", "libmars_hip.so")))
f = lib.marshost_emd_scores
f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
              ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
f.restype = ctypes.c_int


def run(case, threads, reps=1):
    S, fg, bits = (np.ascontiguousarray(case[k]) for k in ("S", "fg", "bits"))
    out = np.zeros(bits.shape[0], dtype=np.float64)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = f(S.ctypes.data, S.shape[1], S.shape[0], S.shape[1], fg.ctypes.data, bits.ctypes.data, bits.shape[0], 0, out.ctypes.data, threads)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    return best * 1e3, out


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(os.path.join(ROOT, "gpurun_out", "emd_cases", x) for x in os.listdir(os.path.join(ROOT, "gpurun_out", "emd_cases")))
    for path in files:
        case = np.load(path)
        ms1, out = run(case, 1)
        ms8, _ = run(case, 8, reps=3)
        dev = float(np.nanmax(np.abs(out - case["emd"])))
        print(f"{os.path.basename(path)}: n={int(case['fg'].sum())} 1 thread {ms1:.0f} core-ms, 8 threads {ms8:.1f} ms, max |d| vs stored {dev:.2e}")
