#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03ac
mkdir -p $O
cd $R/tools
{
for rep in 1 2; do
for v in s2_hb14 r3_nopost r3_relax r3_o2 r3_inl; do
  timeout 120 ./kb_$v 10000000 21 512 768 20 ${v}_k21 24 256
done
done
} > $O/ab.txt 2>&1
cut -c1-100 $O/ab.txt
