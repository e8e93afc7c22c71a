#!/bin/bash
# paced (deadline) legs of the real-time point, GC on/off A/B, compact status A/B:  gpurun -- 'bash tools/rt_deadline.sh [tag] [channels]'
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-rtd}
C=${2:-33792}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
P="python tools/realtime_probe.py --blocks 2"
$P --channels $C --seconds 3 > $O/thr_float.json 2> $O/err.txt
for c in $C $((C*98/100/256*256)) $((C*96/100/256*256)) $((C*94/100/256*256)); do
  $P --channels $c --seconds 10 --paced > $O/paced_float_$c.json 2>> $O/err.txt
done
KQ_RT_GC=1 $P --channels $((C*96/100/256*256)) --seconds 30 --paced > $O/paced_float_gc_on.json 2>> $O/err.txt
$P --channels $((C*96/100/256*256)) --seconds 8 --pcm --compact --paced > $O/paced_pcm_compact.json 2>> $O/err.txt
$P --channels $((C*96/100/256*256)) --seconds 8 --pcm --paced --control-plane > $O/paced_pcm_ctl.json 2>> $O/err.txt
gcc -std=gnu11 -O2 -Iinclude examples/radio_realtime.c -Lka9q_sdr_amd/lib -lka9q_hip -Wl,-rpath,$R/ka9q_sdr_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -lpthread -o /tmp/radio_realtime
/tmp/radio_realtime $((C*96/100/256*256)) 2 20 1 1 1 > $O/c_paced_operator.txt 2>&1
/tmp/radio_realtime $((C*96/100/256*256)) 2 20 0 0 1 > $O/c_paced_float.txt 2>&1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    dl=d.get("deadline") or {}
    print("%-28s C %d factor %.4f ms/call %.4f iv %s late %s/%s worst %s backlog %s lag %s busy %s lockmax %s hold %s (%s) gc %s" % (os.path.basename(f), d["channels"], d["realtime_factor"], d["ms_per_call"],
          d["delivery_interval_ms"], dl.get("late_deliveries"), dl.get("deliveries"), dl.get("worst_lateness_ms"), dl.get("backlog_calls"), dl.get("start_lag_ms"), dl.get("host_busy_fraction"), d.get("lock_wait_max_ms"), d.get("ctl_hold_max_ms"), d.get("worst_lock_holder"), d.get("gc")))
PY
cat $O/c_paced_operator.txt $O/c_paced_float.txt
tail -3 $O/err.txt
