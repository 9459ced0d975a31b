"""Copies the evidence of one `tools/profile_round.sh <tag>` run from gpurun_out/ into profiles/:
the kernel-trace statistics, the GEMM rows of the two counter passes, the per-launch HBM traffic of the
dominant kernel and the bench line.   python tools/collect_profiles.py <tag>"""
import csv
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
src, dst = f"gpurun_out/{tag}", "profiles"
shutil.copy(f"{src}/stats/{tag}_kernel_stats.csv", f"{dst}/{tag}_bench_kernel_stats.csv")
if os.path.exists(f"gpurun_out/{tag}_bench.json"):
    shutil.copy(f"gpurun_out/{tag}_bench.json", f"{dst}/{tag}_bench.json")
for kind in ("fetch", "write"):
    rows = [r for r in csv.DictReader(open(f"{src}/{kind}/{tag}_counter_collection.csv")) if "gemm_tn_ring_kernel" in r["Kernel_Name"]]
    with open(f"{dst}/{tag}_pmc_{kind}_size.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in w.fieldnames})
# the dominant kernel: the 256 x 256 pair-fill ring instantiations (bf16 inputs, either output type: the AlphaCLIP
# QKV and FC2 products); rocprofv3 prints some names demangled and some not
dominant = "2, 4, 8, 4, 2, 4, true, false>|Li2ELi4ELi8ELi4ELi2ELi4ELb1ELb0E"
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", f"{src}/fetch/{tag}_counter_collection.csv",
                       f"{src}/write/{tag}_counter_collection.csv", dominant, f"{dst}/{tag}_pmc_traffic.json"])
