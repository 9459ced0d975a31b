#!/bin/bash
# bench.py A/B over an environment variable: tools/ab_bench.sh VAR val_a val_b [rounds]
set -e
VAR=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],2), 'ms', round(d['roofline']['achieved']), 'TF dom', round(d['roofline']['all_gemm']['achieved']), 'TF all')"
  done
done
