#!/bin/bash
# The headline step without and with the 50 Hz clock / power sampler (--gpu-state):  gpurun -- 'bash tools/gs_probe.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for v in 0 1; do
  python3 $R/bench.py $( [ $v = 1 ] && echo --gpu-state ) --steps 100 --no-cpu-baseline --no-second-row --no-rows --no-host-io 2>/dev/null | grep -o -E '"(kernel_ms|ms_per_step)": [0-9.]*' | head -2 | tr '\n' ' '
  echo " gpustate=$v"
done
done
