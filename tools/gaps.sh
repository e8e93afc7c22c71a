#!/bin/bash
# Idle time between consecutive kernels of the steady-state bench steps (rocprofv3 --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/raw -o t --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --spinup 300 --no-host-io --no-cpu-baseline --no-second-row --no-rows > $OUT/bench.json 2> $OUT/err.log
F=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$F")) if "kq::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-50:]
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    print("%-40s dur %8.1f us  gap-before %7.1f us"%(r["Kernel_Name"].split("(")[0][-40:],(e-s)/1e3,gap))
    prev=e
PY
