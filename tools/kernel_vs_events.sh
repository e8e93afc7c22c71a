#!/bin/bash
# The filter kernel's duration as rocprofv3 sees it against the bank's event interval, with the demodulators on their own
# stream and on the main one:   gpurun -- 'bash tools/kernel_vs_events.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ov in 1 0; do
  export KQ_DEMOD_OVERLAP=$ov
  python3 $R/bench.py --steps 50 --spinup 300 --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step)": [0-9.]*' | head -2 | tr '\n' ' '
  echo " unprofiled, overlap=$ov"
  rm -rf /tmp/kve
  rocprofv3 --kernel-trace --stats -d /tmp/kve -o k --output-format csv -- python3 $R/bench.py --steps 50 --spinup 300 --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step)": [0-9.]*' | head -2 | tr '\n' ' '
  echo " under rocprofv3, overlap=$ov"
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kve/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "full16k" in r["Name"] or "demod64" in r["Name"]:
        print("   %-60s calls %5s avg %9.1f ns" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
done
