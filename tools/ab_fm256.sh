#!/bin/bash
# cfg 2: k_demod_fm256 with 8 against 16 waves per channel (KQ_FM256_WAVES), alternating, one box:  gpurun -- 'bash tools/ab_fm256.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
Q="--config cfg2 --steps 200 --no-cpu-baseline --no-second-row --no-rows --no-host-io --no-realtime --spinup-seconds 6"
for round in 1 2 3; do
  for w in 8 16; do
    KQ_FM256_WAVES=$w python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('waves $w: step %.4f ms  kernel %.4f  demod %.4f  step_frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['demod_ms'], d['step_frac']))"
  done
done
