#!/bin/bash
# cfg 2's step with the demodulators' stream chosen by the library (unset) / never (0) / always (1), twice each
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for val in unset 0 1; do
    if [ $val = unset ]; then unset KQ_DEMOD_OVERLAP; else export KQ_DEMOD_OVERLAP=$val; fi
    python $R/bench.py --config cfg2 --steps 100 --spinup-seconds 6 --no-cpu-baseline --no-second-row --no-rows --no-realtime --no-host-io 2>/dev/null < /dev/null | grep -o -E '"(kernel_ms|ms_per_step|demod_ms|step_frac)": [0-9.]*' | head -4 | tr '\n' ' '
    echo " KQ_DEMOD_OVERLAP=$val"
  done
done
