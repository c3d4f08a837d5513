# Exercises bench.py's N > 1 path (sharded reads, overlapped all-reduce of the accumulators, max-over-ranks timing) on a 1-GPU
# box: two ranks share cuda:0, gloo carries the collectives.  Checks the reduced result against a single-rank run over the
# same 2 x reads.  Not a measurement.
set -e
R=${READS:-3000000}   # total reads: > 2^20, so that both ranks own batches
# (plain `python bench.py --gpus 2` spawns its own two ranks; config 4: seed 0x5EED0004, batches of 2^20 reads round-robin)
python bench.py --gpus 2 --steps 5 --warmup 2 --reads $R --backend gloo --single-device 2>/dev/null | tail -1 > ${TMPDIR:-/tmp}/ntk_n2.json
python bench.py --workload c4 --steps 2 --warmup 1 --reads $R --no-cpu-baseline --no-secondary | tail -1 > ${TMPDIR:-/tmp}/ntk_n1.json
python - <<'PY'
import json, os
a, b = json.load(open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ntk_n2.json"))), json.load(open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ntk_n1.json")))
print("N=2:", a["result"], a["n_gpus"]); print("N=1:", b["result"])
keys = ("n_total", "n_fwd", "sum", "xor")
assert a["n_gpus"] == 2 and all(a["result"][k] == b["result"][k] for k in keys), "sharded + all-reduced result differs from the single-rank result"
assert a["result"]["verified"] and b["result"]["verified"] and a["scaling"] == "strong"
print("n2_on_one_gpu ok")
PY
