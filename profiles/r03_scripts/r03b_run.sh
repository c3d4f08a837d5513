#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R/tools
timeout 120 ./ubench3 > $O/ubench3.txt 2>&1
cat $O/ubench3.txt
{
for rep in 1 2; do
for v in base scnt vaddc scnt_lazy scnt_lazy_g8 vaddc_lazy vaddc_lazy_g8 scnt_g8 cell_lazy floor floor_scnt_lazy; do
  [ -x ./kb_r3_$v ] && timeout 120 ./kb_r3_$v 10000000 21 512 768 20 $v 32 256
done
done
for t in 512 640 896 1024; do timeout 120 ./kb_r3_scnt_lazy 10000000 21 512 $t 20 scnt_lazy_t$t 32 256; done
timeout 120 ./kb_r3_scnt_lazy 10000000 21 768 512 20 scnt_lazy_768x512 32 256
} > $O/ab.txt 2>&1
cat $O/ab.txt
