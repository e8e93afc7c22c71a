#!/bin/bash
# A/B of library variants at the channels-at-real-time operating point: every ab/*.so in place of the product library,
# tools/realtime_probe.py at C channels x B blocks (twice each).   gpurun -- 'bash tools/ab_rt.sh 34560 2'
C=${1:-32768}; B=${2:-2}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/ka9q_sdr_amd/lib/libka9q_hip.so
cp $LIB /tmp/libka9q_hip.keep
for f in $R/ab/*.so; do
  cp $f $LIB
  for rep in 1 2; do
    timeout 120 python $R/tools/realtime_probe.py --channels $C --blocks $B --seconds 2 "$@" 2>/dev/null | grep -o -E '"(ms_per_call|filter_kernel_ms|realtime_factor)": [0-9.]*' | tr '\n' ' '
  done
  echo " $(basename $f .so)"
done
cp /tmp/libka9q_hip.keep $LIB
