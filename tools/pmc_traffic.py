"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in a run of
its own over the same command):

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substr> <out.json>

<kernel substr> may hold alternatives separated by '|' (a kernel name has to contain one of them).

Both counters are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide
(16 B per lane) streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte
per lane stores.  Infinity-Cache hits are included in both (they are fabric-side counters)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter, sub):
    tot, n = 0.0, 0
    byname = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(a in r["Kernel_Name"] for a in sub.split("|")):
            v = float(r["Counter_Value"])
            tot += v
            n += 1
            e = byname[r["Kernel_Name"]]
            e[0] += v
            e[1] += 1
    return tot, n, byname


fetch_csv, write_csv, sub, out = sys.argv[1:5]
f_tot, f_n, f_by = per_kernel(fetch_csv, "FETCH_SIZE", sub)
w_tot, w_n, w_by = per_kernel(write_csv, "WRITE_SIZE", sub)
assert f_n == w_n and f_n > 0, (f_n, w_n)
res = {
    "kernel": sub, "launches": f_n,
    "fetch_kib_raw_per_launch": f_tot / f_n, "fetch_correction": 2.0, "write_kib_per_launch": w_tot / w_n,
    "bytes_per_launch": (2.0 * f_tot / f_n + w_tot / w_n) * 1024.0,
    "by_instantiation": {k: {"launches": v[1], "fetch_kib_raw": v[0] / v[1], "write_kib": w_by[k][0] / max(1, w_by[k][1])}
                         for k, v in f_by.items()},
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `python bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline`; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); Infinity-Cache hits included",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel", "launches", "fetch_kib_raw_per_launch", "write_kib_per_launch", "bytes_per_launch")}))
