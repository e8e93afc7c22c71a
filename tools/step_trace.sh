#!/bin/bash
# Kernel timeline of the headline step: rocprofv3 --kernel-trace of a short bench run, the last kernels in start order with
# the idle time in front of each.   gpurun -- 'bash tools/step_trace.sh [bench args]'   (KQ_DEMOD_OVERLAP etc. pass through)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/steptrace
rm -rf $OUT
rocprofv3 --kernel-trace -d $OUT -o st --output-format csv -- python3 $R/bench.py --steps 20 --spinup 40 --no-cpu-baseline --no-second-row --no-rows --no-host-io "$@" > /dev/null 2>&1
python3 - <<PY
import csv
rows=[]
for r in csv.DictReader(open("$OUT/st_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-36:], r.get("Queue_Id","")))
rows.sort()
rows=rows[-40:-8]
t0=rows[0][0]; last_end=rows[0][0]
for s,e,n,q in rows:
    print("%9.1f us  gap %6.1f  run %8.1f us  q%-3s %s" % ((s-t0)/1e3,(s-last_end)/1e3,(e-s)/1e3,q,n))
    last_end=max(last_end,e)
PY
