#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of the default bench, the bench lines of
# all BASELINE configs, big-batch lines, a two-rank rehearsal and the file-level (cfg1) run.  Outputs under gpurun_out/final/.
# PMC passes are separate from any tracing domain other than --kernel-trace, as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/final
rm -rf "$OUT" && mkdir -p "$OUT"
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-big-batch --no-in-flight"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || exit 2
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 4
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 5
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || exit 6
for c in 2 3 4 5; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-big-batch --no-in-flight > $OUT/stats_cfg$c.log 2>&1 || exit 10
  timeout -k 10 250 python3 bench.py --config $c --steps 20 --warmup 3 --cpu-seconds 3 --no-host-path --no-big-batch 2>/dev/null | grep metric > $OUT/cfg$c.json || exit 7
done
for n in 8192 16384 32768; do
  timeout -k 10 250 python3 bench.py --config 2 --packets $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | grep metric > $OUT/cfg2_$n.json || exit 11
done
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2_32768 -- python3 bench.py --config 2 --packets 32768 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-in-flight > $OUT/stats_cfg2_32768.log 2>&1 || exit 12
timeout -k 10 250 python3 bench.py 2>/dev/null | grep metric > $OUT/bench_default.json || exit 8
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 2>/dev/null | grep metric > $OUT/two_ranks_one_gpu_gloo.json || exit 13
timeout -k 10 250 python3 tools/bench_m4a.py 2>/dev/null | tail -1 > $OUT/cfg1_m4a.json || exit 9
echo collected
