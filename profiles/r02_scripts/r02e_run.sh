#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R/tools
{
for t in 768 1024 896 640 512; do for c in 16 24 32; do timeout 120 ./kb_s2_hb14 10000000 21 512 $t 20 s2_t${t}_c$c $c 256; done; done
timeout 120 ./kb_s2_hb14 10000000 21 768 512 20 s2_768x512 24 256
timeout 120 ./kb_s2_hb14 10000000 21 256 1024 20 s2_256x1024 24 256
timeout 120 ./kb_s2_hb14 10000000 31 512 1024 20 s2_k31_t1024 24 256
timeout 120 ./kb_s2_hb14 10000000 31 512 768 20 s2_k31_t768 24 256
} > $O/ab.txt 2>&1
cat $O/ab.txt
