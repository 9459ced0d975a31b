#!/bin/bash
# GEMM micro-benchmark and whole step for two builds of the library: tools/ab_lib.sh old.so new.so
set -e
A=$1; B=$2
for lib in $A $B $A $B; do
  echo "== $lib"
  MARS_HIP_LIB=$lib VARIANTS=40 ONLY="aclip" ROUNDS=7 timeout -k 10 200 python tools/bench_gemm.py | cut -c1-110
done
tools/ab_bench.sh MARS_HIP_LIB $A $B 3
