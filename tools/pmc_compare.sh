#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the default bench for two builds of libalacgpu.so (A/B diagnosis).
#   tools/pmc_compare.sh <libA.so> <libB.so>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for L in "$@"; do
  n=$(basename $L .so)
  export ALACGPU_LIB=$GRAFT_REPO_ROOT/$L
  rm -rf gpurun_out/pmc_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_$n -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$n.log 2>&1 || echo "failed $n"
  python3 - <<PY
import csv, glob, collections
f = max(glob.glob("gpurun_out/pmc_$n/*/*_counter_collection.csv"))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "alac_decode_ab" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$n", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in agg.items()}, "(millions per launch)")
PY
done
