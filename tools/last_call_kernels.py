"""Per-kernel totals of the last ProposalGenerator.generate() call in a rocprofv3 kernel trace of tools/prof_sam_generate.py
(calls are split at the 8-bit image resample that opens set_image).  usage: last_call_kernels.py <kernel_trace.csv> [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "resample_axis_u8" in r["Kernel_Name"]]
last = rows[idx[::2][-1]:]
t = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r["Kernel_Name"][:100]
    t[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t[k][1] += 1
print("kernels", len(last), "busy ms %.3f" % (sum(v[0] for v in t.values()) / 1e6),
      "span ms %.3f" % ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6))
for k, v in sorted(t.items(), key=lambda kv: -kv[1][0])[: int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{v[0] / 1e6:8.3f} {v[1]:5d}  {k}")
