#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
for seed in 301 302 303; do timeout 400 python tools/gpu_fuzz.py --seconds 240 --seed $seed; done > $O/gpu_fuzz.log 2>&1
cat $O/gpu_fuzz.log
