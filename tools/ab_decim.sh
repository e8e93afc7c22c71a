#!/bin/bash
# A/B timing of library variants (ab/*.so) on the half-band cascade.   gpurun -- 'bash tools/ab_decim.sh [bench_decim args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/ka9q_sdr_amd/lib/libka9q_hip.so
cp $LIB /tmp/libka9q_hip.keep
for f in $R/ab/*.so; do
  cp $f $LIB
  for rep in 1 2; do
    python $R/tools/bench_decim.py "$@" 2>/dev/null | grep -o -E '"ms_per_call": [0-9.]*' | tr '\n' ' '
  done
  echo " $(basename $f .so)"
done
cp /tmp/libka9q_hip.keep $LIB
