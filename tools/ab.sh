#!/bin/bash
# Runs ON THE GPU BOX: A/B of several builds of libalacgpu.so on the bench configs (kernel time from events, ms).
#   tools/ab.sh <rounds> <variant> [<variant> ...]   interleaved over <rounds>; a variant is <lib.so> (path relative to the
#   repo root) or VAR=VALUE@<lib.so> (one environment variable set for that variant, e.g. ALACGPU_DENSE=1@alac.net_amd/csrc/libalacgpu.so)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$1; shift
SPECS=${AB_SPECS:-"2:4096 2:8192 2:32768 3:8192 4:8192 5:4096"}
for r in $(seq 1 $R); do
  for L in "$@"; do
    envset=""
    lib=$L
    if [[ "$L" == *@* ]]; then envset=${L%%@*}; lib=${L##*@}; fi
    export ALACGPU_LIB=$GRAFT_REPO_ROOT/$lib
    line="$(basename $lib .so)${envset:+[$envset]}"
    for spec in $SPECS; do
      c=${spec%%:*}; n=${spec##*:}
      ms=$(env $envset timeout -k 5 120 python3 bench.py --config $c --packets $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-in-flight 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f%s' % (j['roofline']['kernel_ms'], '' if j['status_ok'] else '!BAD'))")
      line="$line  cfg$c@$n=$ms"
    done
    echo "$line"
  done
done
