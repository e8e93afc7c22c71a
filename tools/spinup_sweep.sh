#!/bin/bash
# The headline step against the length of the untimed spin-up, with the clocks sampled over the timed steps:
#   gpurun -- 'bash tools/spinup_sweep.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for sp in 0 300 1000 3000; do
  python3 $R/bench.py --steps 100 --spinup $sp --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline()); g=d.get("gpu_state") or {}
print("step %.4f kernel %.4f  %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], {k:(v["mean"] if isinstance(v,dict) else v) for k,v in g.items() if k.endswith("mhz") or k=="power_w"}), end="")'
  echo "  spinup=$sp"
done
done
