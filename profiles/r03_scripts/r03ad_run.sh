#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03ad
mkdir -p $O
cd $R/tools
{
for k in 21 31; do
for rep in 1 2 3 4; do
for v in s2_hb14 r3_nopost; do
  timeout 120 ./kb_$v 10000000 $k 512 768 20 ${v}_k$k 24 256
done
done
done
} > $O/ab.txt 2>&1
cut -c1-100 $O/ab.txt
