#!/bin/bash
# tools/ab_multi.sh <rounds> <lib>...   interleaved kernel-time comparison of several builds of libalacgpu.so
R=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    ALACGPU_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$L', d['roofline'].get('kernel_ms'))"
  done
done
