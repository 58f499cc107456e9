#!/bin/bash
# Interleaved A/B timing of two builds of libalacgpu.so on the same box:
#   tools/ab_bench.sh <libA.so> <libB.so> [rounds] [bench.py args...]
# Prints kernel ms per batch for each round; box-to-box noise (about 3 %) cancels inside one call.
A=$1; B=$2; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    ALACGPU_LIB=$L python bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$L', d['config'].get('workload'), 'ms_per_step', round(d['ms_per_step'], 4), 'kernel_ms', d['roofline'].get('kernel_ms'))"
  done
done
