#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R/tools
{
for rep in 1 2; do
for k in 31 23 21; do
timeout 120 ./kb_old 10000000 $k 512 768 20 old_k$k 24 256
timeout 120 ./kb_s2_hb14 10000000 $k 512 768 20 new_k$k 24 256; timeout 120 ./kb_r3_widecell 10000000 $k 512 768 20 widecell_k$k 24 256
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
