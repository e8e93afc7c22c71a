#!/bin/bash
# Kernel / copy timeline of bench.py's with_host_io row: rocprofv3 --kernel-trace --memory-copy-trace, then the last steps in
# start order.   gpurun -- 'bash tools/hostio_trace.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/hio
rm -rf $OUT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT -o hio --output-format csv -- python3 $R/bench.py --steps 10 --spinup 20 --no-cpu-baseline --no-second-row --no-rows "$@" > /dev/null 2>&1
python3 - <<PY
import csv
rows=[]
for r in csv.DictReader(open("$OUT/hio_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id","")))
for r in csv.DictReader(open("$OUT/hio_memory_copy_trace.csv")):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r["Direction"], ""))
rows.sort()
t0=rows[-60][0]
for s,e,n,q in rows[-60:]:
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s-t0)/1e3,(e-s)/1e3,q,n))
PY
