#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of bench.py for the non-default configs.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/final_cfgs
rm -rf "$OUT" && mkdir -p "$OUT"
for c in 3 4 5; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_cfg$c.log 2>&1 || exit $c
done
echo collected
