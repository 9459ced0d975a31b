"""Which torch (aten) operators still run inside one benchmark step, and from which line of this repository.
    python tools/trace_torch_ops.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

pkg = load_package()
from importlib import import_module  # noqa: E402

MARS = import_module(pkg.__name__ + ".mars.MARS")
Evaluator = import_module(pkg.__name__ + ".mars.utils.evaluation").Evaluator
dev = torch.device("cuda:0")
mars = MARS.build_MARS_fss(bench.build_args(dev, torch.bfloat16), state_dicts=bench.state_dicts(pkg))
Evaluator.initialize()
ep = pkg.synth.make_episode(0, size=518, nshot=1, nprop=64, fold=0)
ep = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ep.items()}


def step():
    pred = mars.predict(ep["support_imgs"], ep["support_masks"], ep["query_img"], ep["mask_proposals"])
    Evaluator.classify_prediction(pred[None], ep)
    mars.clear()


for _ in range(2):
    step()
torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(torch.is_tensor(a) and a.is_cuda for a in args) or "empty" in name or "zeros" in name:
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT) and "trace_torch_ops" not in fr.filename:
                    where = f"{os.path.relpath(fr.filename, ROOT)[-60:]}:{fr.lineno}"
                    break
            counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step()
torch.cuda.synchronize()
VIEWS = ("view", "slice", "select.int", "aten.t.", "alias", "unsqueeze", "squeeze", "expand", "permute", "transpose", "reshape", "_unsafe_view",
         "empty", "as_strided", "detach", "unbind", "split", "narrow")
launching = [(k, n) for k, n in counts.most_common() if not any(v in k[0] for v in VIEWS)]
print("operators that launch kernels or copies:", sum(n for _, n in launching), "calls;  views and allocations:",
      sum(n for k, n in counts.items()) - sum(n for _, n in launching))
for (name, where), n in launching:
    print(f"{n:5d}  {name:40s} {where}")
