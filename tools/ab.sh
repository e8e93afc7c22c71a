#!/bin/bash
# A/B timing of library variants on the GPU box: every ab/*.so (built here with different EXTRA= flags, see
# tools/ab_build.sh) is dropped in place of the product library and timed with the headline bench.
#   gpurun -- 'bash tools/ab.sh [bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/ka9q_sdr_amd/lib/libka9q_hip.so
cp $LIB /tmp/libka9q_hip.keep
for f in $R/ab/*.so; do
  cp $f $LIB
  for rep in 1 2; do
    python $R/bench.py --steps 50 --no-cpu-baseline --no-second-row --no-rows "$@" 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step)": [0-9.]*' | head -2 | tr '\n' ' '
  done
  echo " $(basename $f .so)"
done
cp /tmp/libka9q_hip.keep $LIB
