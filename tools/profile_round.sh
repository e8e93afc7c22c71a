#!/bin/bash
# Round profile: rocprofv3 --kernel-trace --stats of the default bench (compute_n0 headline + the pruned row), of the
# other configurations, of the half-band cascade, and the FETCH_SIZE / WRITE_SIZE PMC passes of the headline kernel.
# Usage on the GPU box: bash tools/profile_round.sh r01     (writes gpurun_out/profiles_<tag>/; copy into profiles/<tag>/)
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
run_stats() {  # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o $name --output-format csv -- python3 "$@" > $OUT/${name}_bench.json 2> $OUT/${name}.err
  local f=$(find $OUT/raw_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
Q="--no-cpu-baseline --no-second-row --no-rows --no-host-io --spinup 300"
run_stats full_n0_cfg4 $R/bench.py --steps 10 --spinup 300 --cpu-seconds 4
run_stats pruned_cfg4 $R/bench.py --steps 10 --n0 0 $Q
run_stats full_n0_cfg2 $R/bench.py --steps 10 --config cfg2 $Q
run_stats full_n0_cfg3 $R/bench.py --steps 10 --config cfg3 $Q
run_stats pruned_cfg3 $R/bench.py --steps 10 --config cfg3 --n0 0 $Q
run_stats full_n0_cfg5 $R/bench.py --steps 10 --config cfg5 --blocks 16 $Q
run_stats stream_cfg5 $R/bench.py --steps 10 --config cfg5 --blocks 16 --n0 0 $Q
run_stats decim_log6 $R/tools/bench_decim.py
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/raw_pmc_$c -o pmc_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --n0 0 $Q --spinup 0 > $OUT/pmc_$c.log 2>&1
done
F=$(find $OUT/raw_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/raw_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_pruned_cfg4.json k_pruned cfg4 1024 64 pruned > /dev/null
# the same two passes for the full-spectrum kernel with compute_n0 (the bench's second row)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/raw_pmcn0_$c -o pmcn0_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $Q --spinup 0 > $OUT/pmcn0_$c.log 2>&1
done
F=$(find $OUT/raw_pmcn0_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/raw_pmcn0_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_full16k_n0_cfg4.json k_filter_full16k cfg4 1024 64 full > /dev/null
# and for the N = 65536 full-spectrum kernel with compute_n0 (cfg 5)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/raw_pmc5_$c -o pmc5_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --config cfg5 --blocks 16 $Q --spinup 0 > $OUT/pmc5_$c.log 2>&1
done
F=$(find $OUT/raw_pmc5_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/raw_pmc5_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_full64k_n0_cfg5.json k_filter_full16k cfg5 512 16 full > /dev/null
rm -rf $OUT/raw_* $OUT/*.log
ls -la $OUT
for f in $OUT/*_bench.json; do echo "== $f"; tail -1 $f | cut -c1-600; done
