#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R/tools
./ubench > $O/ubench.txt 2>&1
{
for v in cur s2 s2_nolds s2_noexec s2_nomaskalg s2_nosdwa s2_nowindows s2_nodigest s2_noemit s2_loads; do timeout 120 ./kb_$v 10000000 21 768 512 20 $v 32 256; done
for sh in 256 384 768; do for c in 16 32 64; do timeout 120 ./kb_s2 10000000 21 768 512 20 s2_c${c}_s$sh $c $sh; done; done
for g in "1024 512" "1280 512" "512 1024" "1536 256" "1792 256"; do timeout 120 ./kb_s2 10000000 21 $g 20 s2_grid 32 256; done
timeout 120 ./kb_s2 10000000 31 768 512 20 s2_k31 32 256
timeout 120 ./kb_cur 10000000 31 768 512 20 cur_k31 32 256
} > $O/ab.txt 2>&1
cat $O/ab.txt
grep -n "cndmask\|min_u32\|and_b32\|ds_add\|dot4\|mad_u64\|lshl_add_u64\|v_add_u32 \|xor\|alignbit\|sdwa" $O/ubench.txt
