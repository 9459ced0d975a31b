#!/bin/bash
# One round of evidence on the GPU box: kernel-trace statistics of the bench command, then the two
# HBM-traffic counter passes (each in a run of its own).  usage: tools/profile_round.sh <tag>
set -u
TAG=${1:?tag}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o "$TAG" --output-format csv -- python3 $CMD > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o "$TAG" --output-format csv -- python3 $CMD > "$OUT/fetch.log" 2>&1 || { tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o "$TAG" --output-format csv -- python3 $CMD > "$OUT/write.log" 2>&1 || { tail -5 "$OUT/write.log"; exit 1; }
find "$OUT" -name "*.csv" | head -20
