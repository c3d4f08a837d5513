#!/bin/bash
# refresh of the files in profiles/r03c that the last kernel commit (word builds, odd k) touches
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
python tools/path_sweep.py 1,3,4,6,8,11,13,15,16,17,19,21,22,23,24,27,31,32 > $O/path_sweep.txt 2>&1
bash tools/path_pmc.sh r03c > /dev/null 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -1
tail -c 600 $O/bench.json
