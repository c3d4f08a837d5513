#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R/tools
{
for rep in 1 2 3; do
  timeout 120 ./kb_s2_hb14 10000000 21 512 768 20 ship_k21 24 256
  timeout 120 ./kb_r3_lateload 10000000 21 512 768 20 lateload_k21 24 256
done
timeout 120 ./kb_s2_hb14 10000000 31 512 768 20 ship_k31 24 256
timeout 120 ./kb_r3_lateload 10000000 31 512 768 20 lateload_k31 24 256
} > $O/ab.txt 2>&1
cut -c1-110 $O/ab.txt
