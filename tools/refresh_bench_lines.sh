#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), AFTER tools/collect_profiles.sh + tools/summarize_profiles.py have put the PMC figures of
# the current kernel sources into profiles/traffic_cfgN.json: the bench lines once more, so that the committed lines carry
# roofline.traffic and roofline.issue_ceiling (bench.py reports them only when the file's kernel_source_sha matches).
# Outputs under gpurun_out/final2/; copy them over profiles/<tag>_bench_*.json.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/final2
rm -rf "$OUT" && mkdir -p "$OUT"
for c in 3 4 5; do
  timeout -k 10 250 python3 bench.py --config $c --steps 20 --warmup 3 --cpu-seconds 3 --no-host-path --no-big-batch 2>/dev/null | grep metric > $OUT/cfg$c.json || exit 7
done
for n in 8192 16384 32768; do
  timeout -k 10 250 python3 bench.py --config 2 --packets $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | grep metric > $OUT/cfg2_$n.json || exit 11
done
timeout -k 10 300 python3 bench.py 2>/dev/null | grep metric > $OUT/bench_default.json || exit 8
python3 - <<'EOF'
import json
for f in ("bench_default", "cfg3", "cfg4", "cfg5"):
    j = json.load(open("gpurun_out/final2/%s.json" % f))
    r = j["roofline"]
    print(f, j["ms_per_step"], j["value"], r["frac"], r["traffic"], r.get("issue_ceiling", {}).get("frac_of_issue_ceiling"), j["parity_vs_oracle"])
EOF
