#!/bin/bash
# as ab_decim.sh, over several cascades
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/ka9q_sdr_amd/lib/libka9q_hip.so
cp $LIB /tmp/libka9q_hip.keep
for f in $R/ab/*.so; do
  cp $f $LIB
  for args in "--log 6" "--log 3 --out 4194304" "--log 4 --out 4194304" "--log 8 --out 262144"; do
    python $R/tools/bench_decim.py $args 2>/dev/null | grep -o -E '"ms_per_call": [0-9.]*' | tr '\n' ' '
  done
  echo " $(basename $f .so)"
done
cp /tmp/libka9q_hip.keep $LIB
