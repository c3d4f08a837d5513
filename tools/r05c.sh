set -x
python -m pytest tests/test_gpu_parity.py -x -q -k "minimizer" 2>&1 | tail -5 > gpurun_out/r05c_min_tests.log
cat gpurun_out/r05c_min_tests.log
python tools/min_generic_bench.py > gpurun_out/r05c_min_generic.txt 2>&1
cat gpurun_out/r05c_min_generic.txt
