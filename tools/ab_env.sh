#!/bin/bash
# A/B of environment settings on the headline bench within one box:  gpurun -- 'bash tools/ab_env.sh VAR v1 v2 ... -- [bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2; do
  for v in "${vals[@]}"; do
    if [ "$v" == "unset" ]; then unset $var; else export $var=$v; fi
    python $R/bench.py --steps 50 --no-cpu-baseline --no-second-row --no-rows "$@" 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step)": [0-9.]*' | head -2 | tr '\n' ' '
    echo " $var=$v"
  done
done
