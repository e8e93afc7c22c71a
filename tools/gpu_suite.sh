#!/bin/bash
# the GPU suite on the box, its summary lines on stdout and the whole log under gpurun_out/:  gpurun -- 'bash tools/gpu_suite.sh [tag]'
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-suite}
mkdir -p $R/gpurun_out/$T
cd $R && timeout 1500 python -m pytest tests -m gpu -x -q --timeout 420 < /dev/null > gpurun_out/$T/gpu.txt 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/$T/gpu.txt
grep -E "^E  " gpurun_out/$T/gpu.txt | head -20
