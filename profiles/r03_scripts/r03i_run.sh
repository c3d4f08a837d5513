#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compat or kats or iterators" > $O/pytest_compat.log 2>&1
tail -3 $O/pytest_compat.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03i/bench.json").read().strip().splitlines()[-1])
print(d["roofline"]["kernel_ms"], d["roofline"]["frac"])
for k, v in d["secondary"].items(): print(k, v)
PY
tail -3 $O/bench.err
