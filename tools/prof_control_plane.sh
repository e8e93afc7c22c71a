cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_cp
mkdir -p $OUT
name=realtime_32768x2_pcm_control_plane
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -o $name --output-format csv -- python3 $R/tools/realtime_probe.py --channels 32768 --blocks 2 --seconds 0.3 --pcm --control-plane > $OUT/${name}_bench.json 2> $OUT/${name}.err < /dev/null
f=$(find $OUT/raw_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
rm -rf $OUT/raw_$name
head -8 $OUT/${name}_kernel_stats.csv | cut -c1-150
