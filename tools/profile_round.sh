#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes (separate runs, as the guide requires)
# + kbench ablations.  Outputs under gpurun_out/$1/ ; tools/collect_profiles.py turns them into profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- $BENCH --steps 20 --warmup 3 > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $BENCH --steps 5 --warmup 1 --no-verify > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $BENCH --steps 5 --warmup 1 --no-verify > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $BENCH --steps 5 --warmup 1 --no-verify > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- $BENCH --steps 5 --warmup 1 --no-verify > /dev/null 2> $O/pmc_sq2.err
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
if [ -x tools/kb_cur ]; then   # tools/build_kbench.sh; grid = resident blocks (3 x 512 threads per CU)
  ( cd tools; for v in cur nohist nodigest; do [ -x ./kb_$v ] && ./kb_$v 10000000 21 768 512 10 $v 16; done ) > $O/ablation.txt 2>&1
  ( cd tools; ./ubench ) > $O/ubench.txt 2>&1
fi
python tools/quality_bench.py --cutoff 34 > $O/quality_bench.json 2> $O/quality.err
READS=10000000 python tools/pipeline_bench.py > $O/pipeline_and_materialize.json 2> $O/pipeline.err
ls $O
