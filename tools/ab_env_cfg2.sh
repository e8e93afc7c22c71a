#!/bin/bash
# A/B of an environment switch at cfg 2: tools/ab_env_cfg2.sh VAR [extra bench args]  -> ms_per_step / kernel_ms / demod_ms with VAR=0 and VAR=1, twice each
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
for rep in 1 2; do
  for val in 0 1; do
    env $V=$val python $R/bench.py --config cfg2 --steps 100 --spinup 4000 --no-cpu-baseline --no-second-row --no-rows --no-realtime --no-host-io "$@" 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step|demod_ms|step_frac)": [0-9.]*' | tr '\n' ' '
    echo " $V=$val"
  done
done
