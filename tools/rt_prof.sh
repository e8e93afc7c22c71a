#!/bin/bash
# rocprofv3 kernel trace of the channels-at-real-time probe (tools/realtime_probe.py): which kernels fill a call at C channels x B blocks
# Usage on the GPU box: bash tools/rt_prof.sh [channels] [blocks] [extra probe flags]   -> gpurun_out/r05/rt_<C>x<B>_kernel_stats.csv
C=${1:-32768}; B=${2:-2}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/raw_rt -o rt --output-format csv -- python3 $R/tools/realtime_probe.py --channels $C --blocks $B --seconds 0.3 "$@" > $OUT/rt_${C}x${B}.json 2> $OUT/rt_${C}x${B}.err
f=$(find $OUT/raw_rt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $OUT/rt_${C}x${B}_kernel_stats.csv; head -14 $OUT/rt_${C}x${B}_kernel_stats.csv | cut -c1-200; fi
rm -rf $OUT/raw_rt
