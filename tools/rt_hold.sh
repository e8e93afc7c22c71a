#!/bin/bash
# the 60-second paced holds, with the longest interval's breakdown:   gpurun -- 'bash tools/rt_hold.sh [tag] [channels ...]'
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-hold}; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
gcc -std=gnu11 -O2 -Iinclude examples/radio_realtime.c -Lka9q_sdr_amd/lib -lka9q_hip -Wl,-rpath,$R/ka9q_sdr_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -lpthread -o /tmp/radio_realtime
for c in ${@:-34560 32768 30720}; do
  python tools/realtime_probe.py --channels $c --blocks 2 --seconds 60 --paced > $O/py_$c.json 2>> $O/err.txt
  python - <<PY
import json
d=json.loads(open("$O/py_$c.json").read().strip().splitlines()[-1])
print("python $c: late %d of %d (worst %.3f ms), backlog max %d, intervals %s, filter pass mean %.3f max %.3f, longest: %s" % (d["deadline"]["late_deliveries"], d["deadline"]["deliveries"], d["deadline"]["worst_lateness_ms"], d["deadline"]["backlog_calls"]["max"], d["delivery_interval_ms"], d["filter_kernel_ms"], d["filter_kernel_max_ms"], {k: v for k, v in (d["longest_interval"] or {}).items() if k != "note"}))
PY
  /tmp/radio_realtime $c 2 60 0 0 1 > $O/c_$c.txt 2>&1
  grep -E "intervals|deadline|longest|stalls" $O/c_$c.txt | sed "s/^/C $c: /"
done
uptime
