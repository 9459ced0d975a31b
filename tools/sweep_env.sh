#!/bin/bash
# Step time of bench.py under the values of one environment variable.  usage: tools/sweep_env.sh NAME v1 v2 ...
set -u
NAME=$1; shift
for v in "$@"; do
  echo "$NAME=$v"
  env "$NAME=$v" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', round(d['ms_per_step'],2), 'dom', round(d['roofline']['achieved'],1), 'all', round(d['roofline']['all_gemm']['achieved'],1), 'miou', d['miou'])" || exit 1
done
