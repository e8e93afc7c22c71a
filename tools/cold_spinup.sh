#!/bin/bash
# The headline step after an idle pause of 30 s, against the length of the untimed spin-up:  gpurun -- 'bash tools/cold_spinup.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
for sp in 300 3000 10000 300 3000 10000; do
  sleep 30
  python3 $R/bench.py --steps 100 --spinup $sp --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline()); g=d.get("gpu_state") or {}
print("step %.4f kernel %.4f  sclk %s MHz  %s W" % (d["ms_per_step"], d["roofline"]["kernel_ms"], g.get("sclk_mhz"), g.get("power_w")), end="")'
  echo "  spinup=$sp after 30 s idle"
done
