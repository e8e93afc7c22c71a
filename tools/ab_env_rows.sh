#!/bin/bash
# KQ_DEMOD_OVERLAP on / off over the bench configurations, one box:   gpurun -- 'bash tools/ab_env_rows.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "--config cfg4" "--config cfg3" "--config cfg2" "--config cfg5 --blocks 16" "--config cfg4 --n0 0"; do
  for rep in 1 2; do
    for ov in 1 0; do
      export KQ_DEMOD_OVERLAP=$ov
      python3 $R/bench.py $cfg --steps 50 --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step)": [0-9.]*' | head -2 | tr '\n' ' '
      echo " overlap=$ov $cfg"
    done
  done
done
