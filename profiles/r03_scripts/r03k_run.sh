#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
cat $O/bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03k/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"]))
for k, v in d["secondary"].items(): print(k, v)
print(json.dumps(d["cpu_baseline"], indent=0)[:3000])
PY
tail -3 $O/bench.err
