"""Mean counter values per kernel from a rocprofv3 counter_collection.csv: python tools/pmc_summary.py <csv> [substr]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sub in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in v.items():
        print(f"  {c:34s} {sum(vals) / len(vals):18.0f}  (n={len(vals)})")
