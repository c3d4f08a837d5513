#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R/tools
{
for rep in 1 2; do
for v in base sl sl_cxor sl_bperm sl_cxor_bperm sl_cxor_g8 sl_cxor_bperm_g8 sl_cxor_defsched; do
  [ -x ./kb_r3_$v ] && timeout 120 ./kb_r3_$v 10000000 21 512 768 20 $v 32 256
done
done
for t in 640 1024; do timeout 120 ./kb_r3_sl_cxor 10000000 21 512 $t 20 sl_cxor_t$t 32 256; done
} > $O/ab.txt 2>&1
cat $O/ab.txt
