# Exercises bench.py's N > 1 path (sharded reads, overlapped all-reduce of the accumulators, max-over-ranks timing) on a 1-GPU
# box: two ranks share cuda:0, gloo carries the collectives.  Checks the reduced result against a single-rank run over the
# same 2 x reads.  Not a measurement.
set -e
R=${READS:-200000}
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 \
  --steps 5 --warmup 2 --reads $R --backend gloo --single-device --no-cpu-baseline 2>/dev/null | tail -1 > ${TMPDIR:-/tmp}/ntk_n2.json
python bench.py --steps 2 --warmup 1 --reads $((2 * R)) --no-cpu-baseline | tail -1 > ${TMPDIR:-/tmp}/ntk_n1.json
python - <<'PY'
import json, os
a, b = json.load(open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ntk_n2.json"))), json.load(open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ntk_n1.json")))
print("N=2:", a["result"], a["n_gpus"]); print("N=1:", b["result"])
assert a["n_gpus"] == 2 and a["result"] == b["result"], "sharded + all-reduced result differs from the single-rank result"
print("n2_on_one_gpu ok")
PY
