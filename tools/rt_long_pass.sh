#!/bin/bash
# the rare long filter pass of a paced receiver (one or two a minute, 3-20 ms where the mean is 1.4): a kernel trace of a 60 s paced
# run of the C receiver, every filter pass beyond 2.5 ms with what ran on the device around it.  gpurun -- 'bash tools/rt_long_pass.sh [tag]'
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-longpass}
S=${2:-60}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
gcc -std=gnu11 -O2 -Iinclude examples/radio_realtime.c -Lka9q_sdr_amd/lib -lka9q_hip -Wl,-rpath,$R/ka9q_sdr_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -lpthread -o /tmp/radio_realtime < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lp_prof -o rt -- /tmp/radio_realtime 32256 2 $S 1 0 1 > $O/run.txt 2>&1 < /dev/null
cd $R
grep -E "x real time|intervals|deadline|longest|stalls" $O/run.txt
python3 - > $O/long_passes.txt 2>&1 < /dev/null <<PY
import csv, glob
import numpy as np
fs = glob.glob("/tmp/lp_prof/**/*kernel_trace.csv", recursive=True)
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"].split("(")[0].replace("void kq::", "").replace("kq::", "")[:28], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort(key=lambda r: r[1])
t0 = rows[0][1]
print("kernel trace: %d launches over %.1f s" % (len(rows), (rows[-1][2] - t0) / 1e9))
filt = [(i, r) for i, r in enumerate(rows) if "k_filter_full16k" in r[0]]
d = np.array([r[2] - r[1] for _, r in filt]) / 1e6
print("filter passes: n %d mean %.3f p99 %.3f p99.9 %.3f max %.3f ms; beyond 2.5 ms: %d" % (len(d), d.mean(), np.percentile(d, 99), np.percentile(d, 99.9), d.max(), int((d > 2.5).sum())))
# gaps between consecutive filter passes' starts (the receiver's cadence as the device saw it)
st = np.array([r[1] for _, r in filt]) / 1e6
gap = np.diff(st)
print("start-to-start of filter passes: p50 %.3f p99.9 %.3f max %.3f ms" % (np.percentile(gap, 50), np.percentile(gap, 99.9), gap.max()))
import collections, re
by = collections.defaultdict(list)
for m, s2, e2, q in rows: by[(m, q)].append((e2 - s2) / 1e6)
for (m, q), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print("  %-28s q%-3s n %6d mean %8.3f p99.9 %8.3f max %8.3f ms" % (m, q, len(v), v.mean(), np.percentile(v, 99.9), v.max()))
# the pass the library itself names as its longest interval (marker to marker in the queue): 50 warm-up calls in front of the timed part
mm = re.search(r"filter interval in the device's queue ([0-9.]+) ms \(mean [0-9.]+; pass (\d+) of the timed part, ([0-9.]+) ms", open("$O/run.txt").read())
named = set()
if mm:
    named.add(50 + int(mm.group(2)))
    print("the library's longest interval: %s ms at pass %s of the timed part (host between its markers %s ms)" % (mm.group(1), mm.group(2), mm.group(3)))
for n, (i, r) in enumerate(filt):
    if (r[2] - r[1]) / 1e6 <= 2.5 and n not in named:
        continue
    s, e = r[1], r[2]
    print("\nfilter pass of %.3f ms at t = %.3f s (queue %s); the device from 4 ms before it to 2 ms after it [start, end relative to its start, ms]:" % ((e - s) / 1e6, (s - t0) / 1e9, r[3]))
    for m, s2, e2, q in rows[max(0, i - 40):i + 40]:
        if e2 > s - 4e6 and s2 < e + 2e6:
            print("   %-28s q%-3s %9.3f %9.3f  (%.3f ms)" % (m, q, (s2 - s) / 1e6, (e2 - s) / 1e6, (e2 - s2) / 1e6))
PY
head -c 20000 $O/long_passes.txt
rm -rf /tmp/lp_prof
