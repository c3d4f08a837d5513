#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03ab
mkdir -p $O
cd $R/tools
{
for k in 21 31 23 17; do
for rep in 1 2 3; do
  timeout 120 ./kb_old 10000000 $k 512 768 20 rotate_k$k 24 256
  timeout 120 ./kb_s2_hb14 10000000 $k 512 768 20 lateload_k$k 24 256
done
done
} > $O/ab.txt 2>&1
cut -c1-100 $O/ab.txt
