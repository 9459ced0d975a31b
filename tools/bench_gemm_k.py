"""K sweep of one GEMM variant: separates the per-tile fixed cost (prologue + epilogue) from the
per-step cost of the main loop."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ops, lib = pkg.ops, pkg.hip.lib()
dev = torch.device("cuda:0")
V = int(os.environ.get("VARIANT", "3"))
lib.marship_set_gemm_variant(V)
for (M, N, odt, res) in [(36928, 1024, torch.float32, True), (36928, 3072, torch.bfloat16, False),
                         (36928, 1024, torch.float32, False), (65536, 1024, torch.float32, True),
                         (1374, 3072, torch.bfloat16, False), (2748, 3072, torch.bfloat16, False)]:
    line = f"M={M} N={N} {str(odt)[6:]} res={int(res)}:"
    for K in (64, 256, 1024, 2048, 4096):
        a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        b = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M, N, device=dev, dtype=odt)
        kw = dict(bias=bias)
        if res:
            kw["residual"] = out
        ts = []
        for r in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.linear(a, b, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
        ts.sort()
        line += f"  K={K}: {ts[2] * 1e3:7.1f}us"
    print(line, flush=True)
