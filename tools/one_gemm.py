"""A handful of launches of one GEMM shape (for rocprofv3 --pmc passes):
python tools/one_gemm.py M N K variant out(bf16|f32) res(0|1) [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ops, lib = pkg.ops, pkg.hip.lib()
M, N, K, V = (int(x) for x in sys.argv[1:5])
odt = torch.bfloat16 if sys.argv[5] == "bf16" else torch.float32
res = sys.argv[6] == "1"
n = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0")
lib.marship_set_gemm_variant(V)
a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
b = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=odt)
for _ in range(n):
    ops.linear(a, b, out, bias=bias, residual=out if res else None)
torch.cuda.synchronize()
print("done")
