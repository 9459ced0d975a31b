#!/bin/bash
# SQ counters of the fused attention kernel (diagnostic build, AlphaCLIP shape).  usage: tools/attn_pmc.sh <tag>
set -u
TAG=${1:?tag}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 60 tools/diag_bin/diag_attn 64 577 16 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
    -d gpurun_out/$TAG -o a --output-format csv -- tools/diag_bin/diag_attn 64 577 16 > gpurun_out/$TAG.log 2>&1 || { tail -5 gpurun_out/$TAG.log; exit 1; }
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
last = max(int(r["Dispatch_Id"]) for r in rows)
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        print(r["Counter_Name"], r["Counter_Value"])
PY
