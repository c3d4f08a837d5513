#!/bin/bash
# wide builds with the DPP-fused strand pick: parity (whole GPU suite), k sweep, fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python tools/path_sweep.py 4,8,12,15,16,21,23,24,26,28,31,32 > $O/path_sweep.txt 2>&1; cat $O/path_sweep.txt
timeout 200 python tools/gpu_fuzz.py --seconds 100 --seed 78 > $O/gpu_fuzz.log 2>&1; tail -2 $O/gpu_fuzz.log
