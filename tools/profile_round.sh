#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes (separate runs, as the guide requires)
# + kbench ablations + floor kernel + instruction micro-benchmarks + per-path sweeps.  Outputs under gpurun_out/$1/ ;
# tools/collect_profiles.py turns them into profiles/$1/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-secondary --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- $BENCH --steps 20 --warmup 3 > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $BENCH --steps 5 --warmup 1 --no-verify --preheat-ms 5 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $BENCH --steps 5 --warmup 1 --no-verify --preheat-ms 5 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $BENCH --steps 5 --warmup 1 --no-verify --preheat-ms 5 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- $BENCH --steps 5 --warmup 1 --no-verify --preheat-ms 5 > /dev/null 2> $O/pmc_sq2.err
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -o p -- $BENCH --steps 5 --warmup 1 --no-verify --preheat-ms 5 > /dev/null 2> $O/pmc_sq3.err
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
# one rank through the launcher: the RCCL all-reduce path of the N > 1 runs (communicator of size 1) on configs[3]'s 8-GPU shard size
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload c4 --reads 12500000 --no-cpu-baseline --no-secondary > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
if [ -x tools/kb_s2_hb14 ]; then   # tools/build_kbench.sh; 2 x 768 threads per CU, 256 work counters, chunks of 24 tiles
  ( cd tools; for v in cur; do [ -x ./kb_$v ] && ./kb_$v 10000000 21 768 512 20 r01_kernel 16 8; done
    [ -x ./kb_old ] && ./kb_old 10000000 21 512 768 20 r02_region 24 256
    for v in s2_hb14 s2_default a_floor a_nolds a_nomaskalg a_nosdwa a_nodigest a_noemit a_noexec a_loads; do [ -x ./kb_$v ] && ./kb_$v 10000000 21 512 768 20 $v 24 256; done
    ./kb_a_loads 10000000 21 512 768 20 loads_8_counters 16 8
    [ -x ./kb_old ] && ./kb_old 10000000 31 512 768 20 r02_region_k31 24 256
    ./kb_s2_hb14 10000000 31 512 768 20 s2_hb14_k31 24 256
    ./kb_a_floor 10000000 31 512 768 20 a_floor_k31 24 256 ) > $O/ablation.txt 2>&1
  ( cd tools; ./ubench ) > $O/ubench.txt 2>&1
  ( cd tools; ./ubench3 ) > $O/ubench3.txt 2>&1
fi
python tools/path_sweep.py 1,4,6,8,11,15,16,17,19,21,22,23,24,27,31,32 > $O/path_sweep.txt 2>&1
python tools/min_grid.py > $O/min_grid.txt 2>&1
python tools/min_generic_bench.py 2>&1 | grep "k=" > $O/min_generic.txt
bash tools/path_pmc.sh $TAG > /dev/null 2>&1
python tools/compat_bench.py > $O/compat_bench.txt 2>&1
python tools/compat_planes_bench.py >> $O/compat_bench.txt 2>&1
[ -x tools/wbw ] && ( cd tools; ./wbw ) > $O/wbw.txt 2>&1
python tools/min_ab.py > $O/min_ab.txt 2>/dev/null
python tools/min_generic_contigs.py > $O/min_generic_contigs.txt 2>&1
python tools/gpu_fuzz.py --seconds 100 --seed 50${RANDOM:0:2} > $O/gpu_fuzz.log 2>&1
python tools/gpu_fuzz.py --seconds 60 --seed 51${RANDOM:0:2} --minimizers-only >> $O/gpu_fuzz.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log
ls $O
