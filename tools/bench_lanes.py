"""One AlphaCLIP block's chain of kernels (folded-LayerNorm products, fused attention) over 64 sequences: as one
batch on one stream against LANES batches of 64 / LANES sequences, each on a stream of its own."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ops = pkg.ops
dev = torch.device("cuda:0")
NSEQ, T, D, HID, NH = 64, 577, 1024, 4096, 16
LAYERS = int(os.environ.get("LAYERS", "8"))
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
wqkv, bqkv, cqkv = (rn(3 * D, D) * 0.05).to(torch.bfloat16), rn(3 * D), rn(3 * D)
wfc1, bfc1, cfc1 = (rn(HID, D) * 0.05).to(torch.bfloat16), rn(HID), rn(HID)
wproj, bproj = (rn(D, D) * 0.02).to(torch.bfloat16), rn(D) * 0.1
wfc2, bfc2 = (rn(D, HID) * 0.01).to(torch.bfloat16), rn(D) * 0.1
qs = torch.ones(3 * D, device=dev)
qs[:D] = ops.RAW_QUERY_SCALE_LOG2


class Lane:
    def __init__(self, n):
        M = n * T
        self.n, self.M = n, M
        self.x = rn(M, D)
        self.hi = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
        self.lo = torch.zeros_like(self.hi)
        self.part, self.stat = torch.zeros(M, D // 64, 2, device=dev), torch.zeros(M, 2, device=dev)
        self.qkv = torch.zeros(M, 3 * D, device=dev, dtype=torch.bfloat16)
        self.ao = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
        self.hid = torch.zeros(M, HID, device=dev, dtype=torch.bfloat16)
        self.lse = torch.zeros(n * NH, T, device=dev)
        self.stream = torch.cuda.Stream()

    def run(self):
        ops.row_stats(self.x, 1e-5, self.hi, self.lo, self.stat)
        for _ in range(LAYERS):
            ops.linear(self.hi, wqkv, self.qkv, bias=bqkv, gamma=qs, ln_in=(self.stat, cqkv))
            ops.flash_attn(self.qkv, self.ao, self.lse, self.n, T, NH, False, score_scale_log2=1.0)
            ops.linear(self.ao, wproj, self.x, bias=bproj, ln_out=(self.hi, self.lo, self.part))
            ops.ln_finish(self.part, self.M, D, 1e-5, self.stat)
            ops.linear(self.hi, wfc1, self.hid, bias=bfc1, act=2, ln_in=(self.stat, cfc1))
            ops.linear(self.hid, wfc2, self.x, bias=bfc2, ln_out=(self.hi, self.lo, self.part))
            ops.ln_finish(self.part, self.M, D, 1e-5, self.stat)


def timed(lanes):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record()
    for ln in lanes:
        ln.stream.wait_stream(main)
        with torch.cuda.stream(ln.stream):
            ln.run()
    for ln in lanes:
        main.wait_stream(ln.stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / LAYERS * 1e3


configs = {1: [Lane(NSEQ)], 2: [Lane(NSEQ // 2) for _ in range(2)], 4: [Lane(NSEQ // 4) for _ in range(4)]}
res = {k: [] for k in configs}
for r in range(int(os.environ.get("ROUNDS", "7"))):
    for k, lanes in configs.items():
        for ln in lanes:
            ln.x.normal_()
        t = timed(lanes)
        if r >= 2:
            res[k].append(t)
for k, v in res.items():
    v.sort()
    print(f"{k} lane(s): median {v[len(v) // 2]:8.1f} us a layer   min {v[0]:8.1f}")
