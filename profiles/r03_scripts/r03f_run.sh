#!/bin/bash
# Round 3: the refactored regions in every scan2 build - full GPU suite, path sweep, minimizer grid, ablations, default bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
NTK_FUZZ_SECONDS=40 timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 600 python tools/path_sweep.py 4,6,8,11,16,17,21,22,23,24,27,31,32 > $O/path_sweep.txt 2>&1
cat $O/path_sweep.txt
timeout 600 python tools/min_grid.py > $O/min_grid.txt 2>&1
cat $O/min_grid.txt
( cd tools
  for v in s2_hb14 s2_default a_floor a_nolds a_nodigest a_noemit a_nomaskalg a_nosdwa a_loads; do [ -x ./kb_$v ] && timeout 120 ./kb_$v 10000000 21 512 768 20 $v 24 256; done
  timeout 120 ./kb_cur 10000000 21 768 512 20 r01_kernel 16 8
  timeout 120 ./kb_s2_hb14 10000000 31 512 768 20 s2_hb14_k31 24 256
  timeout 120 ./kb_s2_hb14 10000000 23 512 768 20 s2_hb14_k23 24 256 ) > $O/ablation.txt 2>&1
cat $O/ablation.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
