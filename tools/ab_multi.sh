#!/bin/bash
# bench.py over several values of one environment variable: tools/ab_multi.sh VAR rounds val...
set -e
VAR=$1; R=$2; shift 2
for i in $(seq $R); do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],2), 'ms', round(d['roofline']['achieved']), 'TF dom', round(d['roofline']['all_gemm']['achieved']), 'TF all')"
  done
done
