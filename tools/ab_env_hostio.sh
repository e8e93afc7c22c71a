#!/bin/bash
# with_host_io row over values of one environment variable:  gpurun -- 'bash tools/ab_env_hostio.sh VAR v1 v2 ...'
R=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" == "unset" ]; then unset $var; else export $var=$v; fi
    python3 $R/bench.py --steps 50 --spinup 2000 --no-cpu-baseline --no-second-row --no-rows $BENCH_ARGS 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline()); h=d.get("with_host_io",{})
print("step %.4f  kernel %.4f  with_host_io %.4f  d2h %.1f GB/s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], h["ms_per_step"], h["d2h_GBps"]), end="")'
    echo "   $var=$v"
  done
done
