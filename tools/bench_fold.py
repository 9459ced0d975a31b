"""Times the four products of an AlphaCLIP block with and without the folded-LayerNorm epilogues, and the
stand-alone passes they replace, interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ops = pkg.ops
dev = torch.device("cuda:0")
M, D, H = 36928, 1024, 4096
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x = rn(M, D)
xb = x.to(torch.bfloat16)
xn = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
part, stat = torch.zeros(M, D // 64, 2, device=dev), torch.zeros(M, 2, device=dev)
lnw, lnb = rn(D), rn(D)
wqkv, bqkv, cqkv = (rn(3 * D, D) * 0.05).to(torch.bfloat16), rn(3 * D), rn(3 * D)
wfc1, bfc1, cfc1 = (rn(H, D) * 0.05).to(torch.bfloat16), rn(H), rn(H)
wproj, bproj = (rn(D, D) * 0.05).to(torch.bfloat16), rn(D)
wfc2, bfc2 = (rn(D, H) * 0.05).to(torch.bfloat16), rn(D)
qkv = torch.zeros(M, 3 * D, device=dev, dtype=torch.bfloat16)
hid = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
ao = rn(M, D).to(torch.bfloat16)
lo = torch.zeros_like(xb)
ops.row_stats(x, 1e-5, xb, lo, stat)

CASES = {
    "layernorm": lambda: ops.layernorm(x, lnw, lnb, xn, 1e-5),
    "row_stats": lambda: ops.row_stats(x, 1e-5, xb, lo, stat),
    "ln_finish": lambda: ops.ln_finish(part, M, D, 1e-5, stat),
    "qkv plain": lambda: ops.linear(xn, wqkv, qkv, bias=bqkv, gamma=cqkv),
    "qkv ln_in": lambda: ops.linear(xb, wqkv, qkv, bias=bqkv, gamma=cqkv, ln_in=(stat, cqkv)),
    "fc1 plain": lambda: ops.linear(xn, wfc1, hid, bias=bfc1, act=2),
    "fc1 ln_in": lambda: ops.linear(xb, wfc1, hid, bias=bfc1, act=2, ln_in=(stat, cfc1)),
    "proj plain": lambda: ops.linear(ao, wproj, x, bias=bproj, residual=x),
    "proj ln_out": lambda: ops.linear(ao, wproj, x, bias=bproj, ln_out=(xb, lo, part)),
    "fc2 plain": lambda: ops.linear(hid, wfc2, x, bias=bfc2, residual=x),
    "fc2 ln_out": lambda: ops.linear(hid, wfc2, x, bias=bfc2, ln_out=(xb, lo, part)),
}
times = {k: [] for k in CASES}
for r in range(int(os.environ.get("ROUNDS", "8"))):
    for k, fn in CASES.items():
        x.normal_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            times[k].append(e0.elapsed_time(e1) / 4 * 1e3)
for k, v in times.items():
    v.sort()
    print(f"{k:12s} median {v[len(v) // 2]:8.1f} us   min {v[0]:8.1f}")
