#!/bin/bash
# every ab/*.so dropped in place of the product library and timed alternately, REPS times each:  tools/ab_libs.sh REPS bench-args...
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/ka9q_sdr_amd/lib/libka9q_hip.so
REPS=$1; shift
cp $LIB /tmp/libka9q_hip.keep
for rep in $(seq $REPS); do
  for f in $R/ab/*.so; do
    cp $f $LIB
    x=$(python $R/bench.py --steps 100 --spinup-seconds 5 --no-cpu-baseline --no-second-row --no-rows --no-realtime --no-host-io "$@" 2>/dev/null < /dev/null | grep -o -E '"(ms_per_step|kernel_ms)": [0-9.]*' | head -2 | tr '\n' ' ')
    echo "$(basename $f .so) $x"
  done
done
cp /tmp/libka9q_hip.keep $LIB
