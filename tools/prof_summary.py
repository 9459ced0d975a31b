"""Per-step summary of a rocprofv3 kernel_stats.csv: python tools/prof_summary.py <csv> <steps incl. warmup>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU busy per step: {tot / steps / 1e6:.2f} ms")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:72]
    print(f"{n:72s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / steps / 1e6:7.2f} ms {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%")
