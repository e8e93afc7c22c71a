#!/bin/bash
# HBM-side bytes per launch of the half-band kernels (FETCH_SIZE / WRITE_SIZE, one pass each; read side x2 per the gfx950
# note in MI355X_MICROARCH.md) next to their algorithmic bytes, then the SQ counters of tools/pmc_decim.sh.
#   gpurun -- 'bash tools/pmc_decim_traffic.sh > gpurun_out/pmc_decim.txt'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmc_decim_t
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/$c -o $c --output-format csv -- python3 $R/tools/bench_decim.py --steps 2 --warmup 1 > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
def per_kernel(c):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_hb_group" in r["Kernel_Name"]:
                agg[re.search(r"k_hb_group<[^>]*>", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f, w = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
n_in, n_out = 64 << 20, 1 << 20
alg = {4: (8.0 * n_in, 8.0 * n_in / 16), 2: (8.0 * n_in / 16, 12.0 * n_out)}
print("64 Mi input samples, log_decimate 6 (4 + 2 stages); MB per launch, read side = FETCH_SIZE x 1024 x 2")
for k in sorted(f):
    g = 4 if "<4" in k else 2
    rd, wr = f[k] * 1024 * 2 / 1e6, w.get(k, 0) * 1024 / 1e6
    print("  %-30s read %8.1f (algorithmic %8.1f)   written %7.1f (algorithmic %7.1f)" % (k, rd, alg[g][0] / 1e6, wr, alg[g][1] / 1e6))
PY
bash $R/tools/pmc_decim.sh 2>/dev/null
