#!/bin/bash
# fused-minimizer builds under the 6-waves-per-SIMD register budget: parity, grid timings, fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "minim" > $O/pytest_min.log 2>&1; tail -2 $O/pytest_min.log
python tools/min_grid.py > $O/min_grid.txt 2>&1; cat $O/min_grid.txt
timeout 200 python tools/gpu_fuzz.py --seconds 120 --seed 77 > $O/gpu_fuzz.log 2>&1; tail -3 $O/gpu_fuzz.log
