#!/bin/bash
# cfg 5: the memory side of a shared first stage against today's four-fold window reads (tools/handover_probe.hip), with the L2's
# hit / miss and the memory-side byte counters of each mode:   gpurun -- 'bash tools/handover_probe.sh [tag]'
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-handover}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/handover_probe.hip -o /tmp/handover_probe 2> /dev/null
for body in 200 280 360; do timeout 60 /tmp/handover_probe 512 $body; done 2>&1 | tee $O/times.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o p --output-format csv -- /tmp/handover_probe 512 280 > /dev/null 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $O/pmc_mem -o p --output-format csv -- /tmp/handover_probe 512 280 > /dev/null 2>&1 || true
cd $R
python - <<PY | tee $O/counters.txt
import csv, glob, collections
for tag in ("pmc_l2", "pmc_mem"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][-20:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        print(tag, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "(mean per launch, n = %d)" % len(next(iter(d.values()))))
PY
rm -rf $O/pmc_l2 $O/pmc_mem
