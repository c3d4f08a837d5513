#!/bin/bash
# timeline of the H2D copies of one parallel-producer run (rocprofv3 memory-copy + kernel trace, no counters)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pipe_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
THREADS=24 BATCH_MIB=8 READS=10000000 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o p -- python $R/tools/pipeline_sweep.py > $O/out.txt 2> $O/err.txt
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pipe_trace")
f = glob.glob(O + "/t/**/*memory_copy_trace.csv", recursive=True)
print(open(O + "/out.txt").read())
if not f:
    print("no memory copy trace", glob.glob(O + "/t/**/*", recursive=True)[:10]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", ""))) for r in rows]
h2d = [c for c in cp if "HOST_TO_DEVICE" in c[2].upper() or "H2D" in c[2].upper()]
h2d.sort()
# the last run = the last ~190 copies of ~8 MiB: take copies whose duration > 100 us
big = [c for c in h2d if c[1] - c[0] > 100000]
print(len(h2d), "H2D copies,", len(big), "longer than 100 us")
# split into runs by gaps > 5 ms
runs, cur = [], [big[0]]
for c in big[1:]:
    if c[0] - cur[-1][1] > 5_000_000: runs.append(cur); cur = [c]
    else: cur.append(c)
runs.append(cur)
for r in runs[-3:]:
    span = (r[-1][1] - r[0][0]) / 1e6; busy = sum(c[1] - c[0] for c in r) / 1e6
    # two copy streams: copies may overlap - the time at least one is in flight
    u, end = 0, r[0][0]
    for a, b, _ in sorted(r):
        if b > end: u += b - max(a, end); end = b
    print(f"  at least one copy in flight {u / 1e6:.1f} ms of {span:.1f} ms = {u / 1e6 / span:.2f}; bytes per ms of span: see out.txt")
    gaps = sorted((r[i + 1][0] - r[i][1]) / 1e3 for i in range(len(r) - 1))
    print(f"run of {len(r)} copies: span {span:.1f} ms, busy {busy:.1f} ms, mean copy {busy / len(r) * 1e3:.0f} us, gap p50 {gaps[len(gaps)//2]:.0f} us p90 {gaps[len(gaps)*9//10]:.0f} us max {gaps[-1]:.0f} us")
PY
