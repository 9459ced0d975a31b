"""Times the products of a SAM ViT-H encoder layer under every GEMM tile variant (marship_set_gemm_variant)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(require_native=True)
ops, lib = pkg.ops, pkg.hip.lib()
dev, bf = torch.device("cuda"), torch.bfloat16
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("qkv window", 4900, 3840, 1280, bf, 0, False), ("qkv global", 4096, 3840, 1280, bf, 0, False), ("proj window", 4096, 1280, 1280, torch.float32, 0, True),
          ("fc1", 4096, 5120, 1280, bf, 1, False), ("fc2", 4096, 1280, 5120, torch.float32, 0, True)]
for name, M, N, K, odt, act, res in shapes:
    a = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf) * 0.02
    bias = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=odt)
    line = []
    for v in (4, 5, 6, 7, 8, 9, 10, 11):
        lib.marship_set_gemm_variant(v)
        try:
            t = timed(lambda: ops.linear(a, w, out, bias=bias, act=act, residual=out if res else None))
            line.append(f"v{v} {t:6.1f}")
        except Exception as e:
            line.append(f"v{v}    err")
    lib.marship_set_gemm_variant(4)
    print(f"{name:12s} M={M} N={N} K={K}: " + "  ".join(line) + f"   ({2.0*M*N*K/1e9:.0f} GFLOP)", flush=True)
