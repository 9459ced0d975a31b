#!/bin/bash
# A/B of the persistent 256x256 shape (variant 12 / ring shape 8) against one workgroup per tile (8 / 4).
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" --timeout 120 --timeout-method thread
VARIANTS=8,12 ONLY="aclip qkv,aclip fc1,aclip fc2,tails" ROUNDS=7 timeout -k 10 200 python tools/bench_gemm.py
for s in 4 8; do
  timeout -k 10 60 tools/diag_bin/diag_gemm 36928 3072 1024 $s 0 0
  timeout -k 10 60 tools/diag_bin/diag_gemm 32768 1024 4096 $s 1 1
done
