#!/bin/bash
# GPU call r02b: instruction rates, sv2 ablations, skeleton (shards / chunk) sweep, PMC counters of kb_s2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R/tools
./ubench > $O/ubench.txt 2>&1
{
for v in cur s2 s2_nolds s2_noexec s2_nomaskalg s2_nosdwa s2_nowindows s2_nodigest s2_noemit s2_loads; do timeout 120 ./kb_$v 10000000 21 768 512 20 $v 16; done
echo "--- shards / chunk sweep (args: chunk shards)"
for sh in 8 32 64 256; do for c in 8 16 32; do timeout 120 ./kb_s2 10000000 21 768 512 20 s2_c${c}_s$sh $c $sh; done; done
for sh in 8 64 256; do for c in 8 16 32; do timeout 120 ./kb_s2_loads 10000000 21 768 512 20 loads_c${c}_s$sh $c $sh; done; done
for g in "512 512" "1024 256" "1536 256" "2048 256"; do timeout 120 ./kb_s2 10000000 21 $g 20 s2_grid 16 64; timeout 120 ./kb_s2_loads 10000000 21 $g 20 loads_grid 16 64; done
} > $O/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $R/tools/kb_s2 10000000 21 768 512 5 s2 16 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- $R/tools/kb_s2 10000000 21 768 512 5 s2 16 > /dev/null 2> $O/pmc_sq2.err
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_EXP_GDS SQ_INST_LEVEL_LDS --output-format csv -d $O/pmc_sq3 -o p -- $R/tools/kb_s2 10000000 21 768 512 5 s2 16 > /dev/null 2> $O/pmc_sq3.err
ls $O/pmc_sq* | head -30
cat $O/ab.txt
