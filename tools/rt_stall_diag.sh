#!/bin/bash
# where do the rare long delivery intervals of a paced receiver come from?  gpurun -- 'bash tools/rt_stall_diag.sh [tag]'
# (1) the C receiver, paced, with the times of its stalls; (2) the same bank paced WITHOUT host I/O (resident input, nothing over
# the link); (3) a kernel trace of a paced run: per kernel the longest instances and what ran beside them
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-stall}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
gcc -std=gnu11 -O2 -Iinclude examples/radio_realtime.c -Lka9q_sdr_amd/lib -lka9q_hip -Wl,-rpath,$R/ka9q_sdr_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -lpthread -o /tmp/radio_realtime
/tmp/radio_realtime 32256 2 30 1 0 1 > $O/c_paced_32256.txt 2>&1
grep -E "x real time|intervals|deadline|longest|stalls" $O/c_paced_32256.txt
python tools/realtime_probe.py --channels 32256 --blocks 2 --seconds 30 --paced --no-io > $O/paced_noio.json 2> $O/err.txt
python tools/realtime_probe.py --channels 32256 --blocks 2 --seconds 30 --paced --pcm > $O/paced_pcm.json 2>> $O/err.txt
python - <<PY
import json
for f in ("paced_noio","paced_pcm"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "filter kernel mean %.3f max %.3f ms; intervals %s; deadline %s" % (d["filter_kernel_ms"], d["filter_kernel_max_ms"], d["delivery_interval_ms"], {k:d["deadline"][k] for k in ("late_deliveries","deliveries","worst_lateness_ms")}))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o rt -- /tmp/radio_realtime 32256 2 10 1 0 1 > $O/prof_run.txt 2>&1
cd $R
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"].split("(")[0][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
by = collections.defaultdict(list)
for n, s, e in rows: by[n].append(e - s)
import numpy as np
print("kernel trace: %d launches" % len(rows))
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    d = np.array(d) / 1e3
    print("  %-60s n %6d mean %9.1f p99.9 %9.1f max %9.1f us" % (n, len(d), d.mean(), np.percentile(d, 99.9), d.max()))
# the five longest filter kernels: what else ran inside their interval
filt = [r for r in rows if "k_filter_full16k" in r[0]]
filt.sort(key=lambda r: r[1] - r[2])
for n, s, e in filt[:5]:
    beside = [(m, (max(s, s2) - s) / 1e3, (min(e, e2) - s) / 1e3, (e2 - s2) / 1e3) for m, s2, e2 in rows if s2 < e and e2 > s and (s2, e2) != (s, e)]
    print("  filter pass of %.1f us at t = %.3f s; beside it: %s" % ((e - s) / 1e3, (s - rows[0][1]) / 1e9, [(m[:24], round(a), round(b), round(d)) for m, a, b, d in beside]))
PY
rm -rf $O/prof
tail -2 $O/err.txt
