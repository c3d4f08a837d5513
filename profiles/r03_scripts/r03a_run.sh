#!/bin/bash
# Round 3, first GPU call: the new device-count-adaptive tests, the LDS micro-benchmarks and the kernel A/B matrix.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl or self_launch or fallback_is_opt_in" > $O/pytest_new.log 2>&1
timeout 300 python -m pytest tests/test_abi.py -x -q -m gpu >> $O/pytest_new.log 2>&1
tail -5 $O/pytest_new.log
cd $R/tools
timeout 120 ./ubench3 > $O/ubench3.txt 2>&1
cat $O/ubench3.txt
{
for rep in 1 2; do
for v in base append priv priv_append cmpin cmpin_scnt cmpin_app cmpin_scnt_priv cmpin_app_priv floor nolds nodigest; do
  [ -x ./kb_r3_$v ] && timeout 120 ./kb_r3_$v 10000000 21 512 768 20 $v 32 256
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
