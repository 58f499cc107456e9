#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
for L in libalacgpu libalacgpu_exp1 libalacgpu_exp2; do
  export ALACGPU_LIB=$GRAFT_REPO_ROOT/alac.net_amd/csrc/$L.so
  rm -rf gpurun_out/r2f/pmc_$L
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/r2f/pmc_$L -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path > gpurun_out/r2f/pmc_$L.log 2>&1 || echo "failed $L"
  python3 - <<PY
import csv, glob, collections
f = max(glob.glob("gpurun_out/r2f/pmc_$L/*/*_counter_collection.csv"))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("alac_decode_ab_kernel"):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$L", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in agg.items()}, "(millions per launch)")
PY
done
