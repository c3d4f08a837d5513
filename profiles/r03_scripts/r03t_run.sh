#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03t
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
python tools/path_sweep.py 1,4,6,8,11,12,15,16,21,31 > $O/path_sweep.txt 2>&1; cat $O/path_sweep.txt
timeout 200 python tools/gpu_fuzz.py --seconds 90 --seed 79 > $O/gpu_fuzz.log 2>&1; tail -1 $O/gpu_fuzz.log
