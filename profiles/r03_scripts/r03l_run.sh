#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R/tools
{
for rep in 1 2 3; do
for k in 21; do
timeout 120 ./kb_s2_hb14 10000000 $k 512 768 20 new_k$k 24 256
timeout 120 ./kb_r3_pp2 10000000 $k 512 768 20 pp2_k$k 24 256
done
done
timeout 120 ./kb_r3_pp2 10000000 21 512 768 20 pp2_chunk32 32 256
timeout 120 ./kb_r3_pp2 10000000 31 512 768 20 pp2_k31 24 256
timeout 120 ./kb_s2_hb14 10000000 31 512 768 20 new_k31 24 256
} > $O/ab.txt 2>&1
cat $O/ab.txt
