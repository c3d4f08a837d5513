#!/bin/bash
# A/B: the masked region with exec never narrowed (NTK_ABL_NOEXEC) against the shipped one - what a no-break fast path could gain
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R/tools
{
for rep in 1 2; do
for k in 21 23 31 12; do
  timeout 120 ./kb_s2_hb14 10000000 $k 512 768 20 ship_k$k 16 256
  timeout 120 ./kb_a_noexec 10000000 $k 512 768 20 noexec_k$k 16 256
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
