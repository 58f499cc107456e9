#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters of the cfg2 launch for one or more builds of the library.
#   tools/pmc_icache.sh <lib.so> [<lib.so> ...]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache; mkdir -p $OUT
for L in "$@"; do
  export ALACGPU_LIB=$GRAFT_REPO_ROOT/$L
  name=$(basename $L .so)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-big-batch --no-in-flight --no-extra > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 3; }
  python3 - "$OUT/$name" "$name" <<'P'
import sys, glob, csv, collections
d, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in acc:
    if "alac_decode_ab_kernel" in k:
        print(name, k, {c: round(v / max(n[k], 1)) for c, v in acc[k].items()})
P
done
