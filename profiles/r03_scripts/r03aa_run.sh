#!/bin/bash
# late tile load as the default loop: parity (GPU suite), every family's time, minimizer grid, fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03aa
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -1
python tools/path_sweep.py 1,4,8,11,15,16,17,21,23,24,27,31,32 > $O/path_sweep.txt 2>&1; cat $O/path_sweep.txt
python tools/min_grid.py 2>&1 | grep -v amdgpu.ids > $O/min_grid.txt; grep "w=11\|w=10 q" $O/min_grid.txt
python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/bench_nocpu.json; python -c "
import json; d=json.loads(open('$O/bench_nocpu.json').read()); print(d['roofline']['kernel_ms'], d['roofline']['frac']); [print(k, v.get('kernel_ms')) for k,v in d['secondary'].items()]"
timeout 200 python tools/gpu_fuzz.py --seconds 100 --seed 82 2>/dev/null | tail -1 | tee $O/gpu_fuzz.log
