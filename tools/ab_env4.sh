#!/bin/bash
# tools/ab_env4.sh VAR cfg: ms_per_step with VAR=0 / VAR=1 alternating, four times each, and the two means
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; CFG=$2; shift 2
for rep in 1 2 3 4; do
  for val in 0 1; do
    x=$(env $V=$val python $R/bench.py --config $CFG --steps 200 --spinup-seconds 5 --no-cpu-baseline --no-second-row --no-rows --no-realtime --no-host-io "$@" 2>/dev/null < /dev/null | grep -o -E '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "$CFG $V=$val $x"
  done
done | tee /tmp/ab4.txt
awk '{s[$2]+=$3; n[$2]++} END{for(k in s) printf "mean %s %.4f\n", k, s[k]/n[k]}' /tmp/ab4.txt
