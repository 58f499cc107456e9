#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): for every BASELINE config that fits one GPU (cfg2..cfg5 at their per-GPU sizes) a
# rocprofv3 --kernel-trace --stats run and three PMC passes of the same bench command (FETCH_SIZE / WRITE_SIZE+GRBM / SQ: one
# counter group per pass, no tracing domain other than --kernel-trace, as the pool requires); then the bench lines themselves,
# big-batch lines, the two-rank rehearsal and the file-level (cfg1) run.  Outputs under gpurun_out/final/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/final
rm -rf "$OUT" && mkdir -p "$OUT"
for c in ${CONFIGS:-2 3 4 5}; do
  B="python3 bench.py --config $c --steps 12 --warmup 3 --no-cpu-baseline --no-host-path --no-big-batch --no-in-flight"
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- $B > $OUT/stats_cfg$c.log 2>&1 || exit 2
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cfg$c -- $B > $OUT/pmc_fetch_cfg$c.log 2>&1 || exit 3
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write_cfg$c -- $B > $OUT/pmc_write_cfg$c.log 2>&1 || exit 4
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq_cfg$c -- $B > $OUT/pmc_sq_cfg$c.log 2>&1 || exit 5
  echo "profiled cfg$c"
done
# the issue-bound regime: cfg2 in a 32768-packet batch (the dense arrangement)
B="python3 bench.py --config 2 --packets 32768 --steps 8 --warmup 2 --no-cpu-baseline --no-host-path --no-big-batch --no-in-flight"
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2_32768 -- $B > $OUT/stats_cfg2_32768.log 2>&1 || exit 12
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_cfg2_32768 -- $B > $OUT/pmc_sq_cfg2_32768.log 2>&1 || exit 13
if [ -z "$PROFILES_ONLY" ]; then
  for c in 2 3 4 5; do
    timeout -k 10 250 python3 bench.py --config $c --steps 20 --warmup 3 --cpu-seconds 3 --no-host-path --no-big-batch 2>/dev/null | grep metric > $OUT/cfg$c.json || exit 7
  done
  for n in 8192 16384 32768; do
    timeout -k 10 250 python3 bench.py --config 2 --packets $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | grep metric > $OUT/cfg2_$n.json || exit 11
  done
  timeout -k 10 300 python3 bench.py 2>/dev/null | grep metric > $OUT/bench_default.json || exit 8
  timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 2>/dev/null | grep metric > $OUT/two_ranks_one_gpu_gloo.json || exit 14
  timeout -k 10 250 python3 tools/bench_m4a.py 2>/dev/null | tail -1 > $OUT/cfg1_m4a.json || exit 9
fi
echo collected
