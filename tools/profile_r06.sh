#!/bin/bash
# Round-6 profile: rocprofv3 --kernel-trace --stats of every bench row (+ the channels-at-real-time operating point), the
# FETCH_SIZE / WRITE_SIZE passes of cfg 4 / cfg 2 / cfg 3 / cfg 5 (so that every rows.* entry of the bench line has a
# traffic_source), and the SQ counter passes of the headline kernel and of cfg 2's kernels (k_demod_fm256 included).
# Usage on the GPU box: bash tools/profile_r06.sh     (writes gpurun_out/profiles_r06/; copy into profiles/r06/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_r06
mkdir -p $OUT
Q="--no-cpu-baseline --no-second-row --no-rows --no-host-io --no-realtime --spinup 300"
run_stats() {  # name, program args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o $name --output-format csv -- python3 "$@" > $OUT/${name}_bench.json 2> $OUT/${name}.err < /dev/null
  local f=$(find $OUT/raw_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/raw_$name
}
run_stats full_n0_cfg4 $R/bench.py --steps 10 $Q
run_stats pruned_cfg4 $R/bench.py --steps 10 --n0 0 $Q
run_stats full_n0_cfg2 $R/bench.py --steps 10 --config cfg2 $Q
run_stats full_n0_cfg3 $R/bench.py --steps 10 --config cfg3 $Q
run_stats full_n0_cfg5 $R/bench.py --steps 10 --config cfg5 --blocks 16 $Q
run_stats realtime_32768x2 $R/tools/realtime_probe.py --channels 32768 --blocks 2 --seconds 0.3
run_stats realtime_32768x2_pcm $R/tools/realtime_probe.py --channels 32768 --blocks 2 --seconds 0.3 --pcm
# the same with an operator at work (k_ctl_apply and k_design in front of the calls) and with 4096 swept channels (the PLAIN == 2 variant)
run_stats realtime_32768x2_pcm_control_plane $R/tools/realtime_probe.py --channels 32768 --blocks 2 --seconds 0.3 --pcm --control-plane
run_stats realtime_32768x2_pcm_swept4096 $R/tools/realtime_probe.py --channels 32768 --blocks 2 --seconds 0.3 --pcm --swept 4096
run_stats realtime_32768x2_pcm_compact $R/tools/realtime_probe.py --channels 32768 --blocks 2 --seconds 0.3 --pcm --compact
pmc_pair() {  # tag, kernel substring, config, channels, blocks, fwd, bench args...
  local tag=$1 kern=$2 cfg=$3 ch=$4 bl=$5 fwd=$6; shift 6
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c -d $OUT/raw_${tag}_$c -o ${tag}_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $Q --spinup 0 "$@" > $OUT/${tag}_$c.log 2>&1 < /dev/null
  done
  local F=$(find $OUT/raw_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  local W=$(find $OUT/raw_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_${tag}.json $kern $cfg $ch $bl $fwd > /dev/null
  rm -rf $OUT/raw_${tag}_* $OUT/${tag}_*.log
}
pmc_pair full16k_n0_cfg4 k_filter_full16k cfg4 1024 64 full
pmc_pair full16k_n0_cfg2 k_filter_full16k cfg2 256 64 full --config cfg2
pmc_pair full16k_n0_cfg3 k_filter_full16k cfg3 1024 64 full --config cfg3
pmc_pair full64k_n0_cfg5 k_filter_full16k cfg5 512 16 full --config cfg5 --blocks 16
pmc_pair pruned_cfg4 k_pruned_resident cfg4 1024 64 pruned --n0 0
SQ="--steps 2 --warmup 1 --spinup 0 --no-host-io --no-cpu-baseline --no-second-row --no-rows --no-realtime"
timeout 900 bash $R/tools/pmc_sq.sh k_filter_full16k bench.py $SQ > $OUT/pmc_sq_full16k_n0_cfg4.txt 2>&1 < /dev/null
timeout 900 bash $R/tools/pmc_sq.sh k_ bench.py $SQ --config cfg2 > $OUT/pmc_sq_cfg2_all_kernels.txt 2>&1 < /dev/null
timeout 900 bash $R/tools/pmc_sq.sh k_demod64 bench.py $SQ > $OUT/pmc_sq_demod64_cfg4.txt 2>&1 < /dev/null
timeout 900 bash $R/tools/pmc_sq.sh k_filter_full16k bench.py $SQ --config cfg5 --blocks 16 > $OUT/pmc_sq_full64k_n0_cfg5.txt 2>&1 < /dev/null
rm -rf $R/gpurun_out/pmc_sq_*
ls -la $OUT
