#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R/tools
{
for n in 1250000 2500000 5000000 10000000 20000000 40000000; do timeout 120 ./kb_r3_sl $n 21 512 768 20 sl_n$n 32 256; done
for c in 4 8 16 24 32 48; do timeout 120 ./kb_r3_sl 10000000 21 512 768 20 sl_chunk$c $c 256; done
for n in 2500000 10000000 40000000; do timeout 120 ./kb_r3_base $n 21 512 768 20 base_n$n 32 256; done
} > $O/ab.txt 2>&1
cat $O/ab.txt
