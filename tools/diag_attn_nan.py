"""Where a fused-attention result differs from the f32 softmax: rows / columns of non-finite values and of large errors."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(require_native=True)
B, T, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 1374, 2)
D = H * 64
g = torch.Generator().manual_seed(5)
x = torch.randn(B * T, 3 * D, generator=g).cuda() * 1.5
c2 = pkg.ops.RAW_QUERY_SCALE_LOG2
x[:, :D] *= c2
qkv = x.to(torch.bfloat16)
O = torch.full((B * T, D), float("nan"), device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B * H, T, device="cuda")
pkg.ops.flash_attn(qkv, O, lse, B, T, H, False, score_scale_log2=1.0)
q = qkv.float().view(B, T, 3, H, 64)
s = torch.einsum("bthd,bshd->bhts", q[:, :, 0], q[:, :, 1]) * 0.6931471805599453
ref = torch.einsum("bhts,bshd->bthd", s.softmax(-1), q[:, :, 2]).reshape(B * T, D)
bad = ~torch.isfinite(O.float())
err = (O.float() - ref).abs()
err[bad] = 99.0
big = err > 3e-2
print("non-finite:", int(bad.sum()), "large errors:", int(big.sum()))
rows = big.any(1).nonzero().flatten().tolist()
print("rows:", rows[:40], "..." if len(rows) > 40 else "", len(rows))
for h in range(H):
    blk = big[:, 64 * h:64 * h + 64]
    r = blk.any(1).nonzero().flatten().tolist()
    c = blk.any(0).nonzero().flatten().tolist()
    print(f"head {h}: rows {r[:8]}..{r[-3:] if r else ''} ({len(r)}), cols {c}")
