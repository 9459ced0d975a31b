"""Fused-attention micro-benchmark at the tower shapes (random data), with a correctness check."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ops = pkg.ops
dev = torch.device("cuda:0")
for name, B, T, H in [("aclip", 64, 577, 16), ("dino ", 2, 1374, 16), ("clip ", 1, 1090, 12), ("text ", 3, 77, 12)]:
    D = H * 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * T, 3 * D, generator=g).to(dev).to(torch.bfloat16)
    O = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B * H, T, device=dev)
    causal = name.startswith("text")
    ops.flash_attn(qkv, O, lse, B, T, H, causal)
    q = qkv[:T].float().view(T, 3, H, 64)
    s = torch.einsum("thd,shd->hts", q[:, 0], q[:, 1]) * 0.125
    if causal:
        s = s + torch.full((T, T), float("-inf"), device=dev).triu_(1)
    ref = torch.einsum("hts,shd->thd", s.softmax(-1), q[:, 2]).reshape(T, D)
    err = float((O[:T].float() - ref).abs().max())
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(3):
            ops.flash_attn(qkv, O, lse, B, T, H, causal)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ts.sort()
    flops = 4.0 * B * H * T * T * 64 * (0.5 if causal else 1.0)
    print(f"{name} B={B} T={T} H={H}: {ts[2] * 1e3:8.1f} us  {flops / ts[2] / 1e9:7.1f} TF  max err {err:.4f}", flush=True)
