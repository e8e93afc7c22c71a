#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
row() {
  python $R/bench.py "$@" --steps 50 --no-cpu-baseline --no-second-row --no-host-io 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.readline())
r = d["roofline"]
print("%-40s step %.4f ms  kernel %.4f ms  frac %.4f  demod %.4f ms" % (" ".join(sys.argv[1:]), d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("demod_ms", 0)))' "$@"
}
row --config cfg2 --channels 1024
row --config cfg2 --channels 1024 --blocks 16
row --config cfg4 --channels 256
row --config cfg4 --channels 256 --blocks 16
