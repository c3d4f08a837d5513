#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R/tools
{
for rep in 1 2; do
for v in sl sl_acc2 sl_ldslast sl_acc2_ldslast sl_maxilp sl_minreg sl_maxocc sl_defsched; do
  [ -x ./kb_r3_$v ] && timeout 120 ./kb_r3_$v 10000000 21 512 768 20 $v 16 256
done
done
timeout 120 ./kb_r3_sl 10000000 31 512 768 20 sl_k31 16 256
timeout 120 ./kb_r3_sl 10000000 23 512 768 20 sl_k23 16 256
} > $O/ab.txt 2>&1
cat $O/ab.txt
