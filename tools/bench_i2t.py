"""Times ops.sam_i2t_block and ops.sam_t2i_block alone at the proposal stage's size (1024 prompts x 4096 image tokens)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package(require_native=True)
ops = pkg.ops
dev, bf = torch.device("cuda"), torch.bfloat16
B, Ni, Tk = int(os.environ.get("B", "1024")), 4096, 7
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
wq, wo = (r(128, 256) * 0.08).to(bf), (r(256, 128) * 0.1).to(bf)
slot = torch.tensor([16 * h + 8 * (e // 4) + 4 * gg + e % 4 for h in range(8) for gg in range(2) for e in range(8)], device=dev)
wq_runs = wq.view(128, 32, 8).permute(1, 0, 2).contiguous()
wo_runs = wo[:, slot].view(256, 16, 8).permute(1, 0, 2).contiguous()
qbias, prm = r(Ni, 128), torch.cat([r(256) * 0.2, r(256) * 0.3 + 1, r(256) * 0.2]).contiguous()
k, v = r(B * Tk, 128).to(bf), r(B * Tk, 128).to(bf)
kvo = torch.empty(B, 8, 64, 2, 8, device=dev, dtype=bf)
ops.sam_i2t_kv_operands(k, v, B, Tk, kvo)
x0, res0 = r(Ni, 256).to(bf), r(Ni, 256)
X = torch.randn(B * Ni, 256, device=dev, dtype=bf)
R = torch.randn(B * Ni, 256, device=dev)
Y = torch.empty(B * Ni, 256, device=dev)
C1 = torch.empty(B * Ni, 256, device=dev, dtype=bf)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = {
    "shared in, y + c1 out (first layer)": lambda: ops.sam_i2t_block(x0, wq_runs, wo_runs, qbias, prm, kvo, Tk, res0, B, Ni, 1e-5, Y, C1),
    "shared in, c1 out only": lambda: ops.sam_i2t_block(x0, wq_runs, wo_runs, qbias, prm, kvo, Tk, res0, B, Ni, 1e-5, None, C1),
    "per-prompt in, c1 out (last layer)": lambda: ops.sam_i2t_block(X, wq_runs, wo_runs, qbias, prm, kvo, Tk, R, B, Ni, 1e-5, None, C1),
    "per-prompt in, y + c1 out (middle layer)": lambda: ops.sam_i2t_block(X, wq_runs, wo_runs, qbias, prm, kvo, Tk, R, B, Ni, 1e-5, Y, C1),
}
for name, fn in cases.items():
    print(f"i2t {name}: {timed(fn):.3f} ms", flush=True)
