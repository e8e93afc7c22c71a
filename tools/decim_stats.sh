#!/bin/bash
# per-kernel durations of the half-band cascade: gpurun -- 'bash tools/decim_stats.sh [bench_decim args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dstat
rocprofv3 --kernel-trace --stats -d /tmp/dstat -o d --output-format csv -- python3 $R/tools/bench_decim.py --steps 300 "$@" > /tmp/dstat.log 2>&1
tail -1 /tmp/dstat.log
f=$(find /tmp/dstat -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "hb_" in n or "rocclr" in n:
        print("%-70s calls %5s avg %9.1f ns min %8s" % (n[:70], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
