#!/bin/bash
# A/B of an environment switch on one box: tools/ab_env.sh VAR cfgA[,cfgB...] [extra bench args] -> ms_per_step / kernel_ms / demod_ms
# with VAR=0 and VAR=1, twice each per config
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; CFGS=$2; shift 2
for cfg in ${CFGS//,/ }; do
  for rep in 1 2; do
    for val in 0 1; do
      env $V=$val python $R/bench.py --config $cfg --steps 100 --spinup-seconds 6 --no-cpu-baseline --no-second-row --no-rows --no-realtime --no-host-io "$@" 2>/dev/null < /dev/null | grep -o -E '"(kernel_ms|ms_per_step|demod_ms|step_frac)": [0-9.]*' | head -4 | tr '\n' ' '
      echo " $cfg $V=$val"
    done
  done
done
