#!/bin/bash
# GPU call r02a: instruction rates + first sv2 A/B (old kernel vs sv2 variants), k = 21 and 31
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R/tools
./ubench > $O/ubench.txt 2>&1
{
for g in "768 512" "1024 512" "512 1024"; do
  timeout 120 ./kb_cur 10000000 21 $g 20 cur 16
  for v in s2 s2_mis s2_salu s2_nolds s2_loads; do timeout 120 ./kb_$v 10000000 21 $g 20 $v 16; done
done
for g in "512 512" "256 1024"; do
  for v in s2_hb14 s2_mis14; do timeout 120 ./kb_$v 10000000 21 $g 20 $v 16; done
done
for v in cur s2 s2_mis; do timeout 120 ./kb_$v 10000000 31 768 512 20 $v 16; done
timeout 120 ./kb_s2 10000000 23 768 512 20 s2 16
for c in 4 8 32 64; do timeout 120 ./kb_s2 10000000 21 768 512 20 s2_chunk$c $c; done
} > $O/ab.txt 2>&1
cat $O/ab.txt
