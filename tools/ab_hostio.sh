#!/bin/bash
# with_host_io row of the bench with the demodulators overlapped (1) or on the main stream (0):  gpurun -- 'bash tools/ab_hostio.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for ov in 1 0; do
    export KQ_DEMOD_OVERLAP=$ov
    python3 $R/bench.py --steps 50 --no-cpu-baseline --no-second-row --no-rows 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline()); h=d.get("with_host_io",{})
print("step %.4f  kernel %.4f  with_host_io %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], {k:h[k] for k in h if k in ("ms_per_step","fraction_of_value","vs_value")}))'
    echo "   overlap=$ov"
  done
done
