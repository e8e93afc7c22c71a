#!/bin/bash
# One line per bench configuration: step, kernel, roofline fraction.   gpurun -- 'bash tools/bench_rows.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
row() {
  python $R/bench.py "$@" --steps 50 --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.readline())
r = d["roofline"]
g = d.get("gpu_state") or {}
print("%-28s step %.4f ms  kernel %.4f ms  frac %.4f  valu_frac %.3f  demod %.4f ms  value %.0f  (%s MHz, %s W)" % (" ".join(sys.argv[1:]), d["ms_per_step"], r["kernel_ms"], r["frac"], r["valu_frac"], r.get("demod_ms", 0), d["value"], g.get("sclk_mhz"), g.get("power_w")))' "$@"
}
row --config cfg4
row --config cfg3
row --config cfg2
row --config cfg2 --blocks 256
row --config cfg5 --blocks 16
row --config cfg5 --blocks 16 --n0 0
row --config cfg4 --n0 0
