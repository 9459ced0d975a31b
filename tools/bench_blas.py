"""Library GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the AlphaCLIP product shapes, as a yardstick for what
this chip sustains on a dense bf16 contraction of the same size (no epilogue: plain C = A . B^T)."""
import torch

dev = torch.device("cuda:0")
SHAPES = [("aclip qkv", 36928, 3072, 1024), ("aclip proj", 36928, 1024, 1024), ("aclip fc1", 36928, 4096, 1024),
          ("aclip fc2", 36928, 1024, 4096), ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()
    best, tot, n = 1e9, 0.0, 0
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.matmul(a, b.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e3
        best, tot, n = min(best, t), tot + t, n + 1
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M} N={N} K={K}: {tot / n:8.1f} us {fl / (tot / n * 1e-6) / 1e12:7.1f} TF (min {best:.1f} us)", flush=True)
