"""Writes the transport solver's inputs for a few benchmark episodes (similarity matrix, support foreground
rows, per-proposal patch bits) to gpurun_out/emd_cases/, so that the solver can be timed and tuned on the host
alone (tools/bench_emd.py): the heaviest episodes by foreground rows and a typical one."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package(require_native=True)
from importlib import import_module  # noqa: E402

MARS = import_module(pkg.__name__ + ".mars.MARS")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
mars = MARS.build_MARS_fss(bench.build_args(dev, torch.bfloat16), state_dicts=bench.state_dicts(pkg))
fmm = mars.filtering_merging_component
NEP = int(os.environ.get("EPISODES", "24"))
cases = []
for e in range(NEP):
    ep = pkg.synth.make_episode(e, size=bench.SIZE, nshot=1, nprop=bench.NPROP, fold=0)
    ep = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ep.items()}
    mars.predict(ep["support_imgs"], ep["support_masks"], ep["query_img"], ep["mask_proposals"])
    w = fmm.last["w"]
    h = mars.visual_visual_alignment_component.similarity_handle()
    S = h.wait()[:, :h.P].clone().numpy()
    cases.append(dict(e=e, S=S, fg=w["fg_host"].clone().numpy(), bits=w["bits_host"].clone().numpy(),
                      emd=w["emd_host"].clone().numpy(), ms=fmm.timing["emd_ms"]))
    print(e, "fg rows", int(cases[-1]["fg"].sum()), "mean sinks", float(cases[-1]["bits"].sum(1).mean()), "emd ms", round(fmm.timing["emd_ms"], 1), flush=True)
    mars.clear()
out = os.path.join("gpurun_out", "emd_cases")
os.makedirs(out, exist_ok=True)
by = sorted(cases, key=lambda c: -c["ms"])
keep = [by[0], by[1], by[len(by) // 2]]
for c in keep:
    np.savez_compressed(os.path.join(out, f"ep{c['e']}.npz"), S=c["S"], fg=c["fg"], bits=c["bits"], emd=c["emd"])
    print("kept", c["e"], c["ms"])
