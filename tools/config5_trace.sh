#!/bin/bash
# Copy / kernel timeline of configs[4] on the streamed gzip route ("H2D + scan inside the inflate span"): rocprofv3 kernel
# + memory-copy trace (no counters) of tools/config5_stream_run.py, then, for the LAST call, the H2D copies and scan kernels per 20 ms slice.
TAG=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/c5t -o p -- python $R/tools/config5_stream_run.py > $O/c5_out.txt 2> $O/c5_err.txt
cd $R
python3 - "$O" <<'PY' | tee $O/config5_trace.txt
import csv, glob, re, sys
O = sys.argv[1]
out = open(O + "/c5_out.txt").read()
print(out)
calls = [(int(a), int(b)) for a, b in re.findall(r"monotonic_ns (\d+) \.\. (\d+)", out)]
cp = []
for f in glob.glob(O + "/c5t/**/*memory_copy_trace.csv", recursive=True):
    cp += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "HOST_TO_DEVICE" in r.get("Direction", "").upper()]
ks = []
for f in glob.glob(O + "/c5t/**/*kernel_trace.csv", recursive=True):
    ks += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
if not calls or not cp or not ks:
    print("missing data", len(calls), len(cp), len(ks)); raise SystemExit
scans = [k for k in ks if "scan2_kernel" in k[2] or "minimizer_scan_kernel" in k[2]]
# the last call: the copies / kernels after the previous call's end (rocprofv3 timestamps and CLOCK_MONOTONIC share the clock on this stack;
# if they do not, fall back to "the last third of the big copies")
a, b = calls[-1]
big = sorted(c for c in cp if c[1] - c[0] > 50_000)
inside = [c for c in big if a - 5_000_000 <= c[0] <= b + 5_000_000]
if len(inside) < 10:
    inside = big[len(big) * 2 // 3:]
    a, b = inside[0][0], inside[-1][1]
    print("(timestamps not on CLOCK_MONOTONIC: the last third of the copies taken as the last call)")
sc = [k for k in scans if a - 5_000_000 <= k[0] <= b + 5_000_000]
span = (b - a) / 1e6
print(f"last call: {span:.1f} ms of wall time; {len(inside)} H2D copies > 50 us, {len(sc)} scan kernels inside it")
print(f"  first H2D copy starts {(inside[0][0] - a) / 1e6:.1f} ms after the call starts, last scan kernel ends {(b - max(k[1] for k in sc)) / 1e6:.1f} ms before it returns")
print(f"  H2D busy {sum(c[1] - c[0] for c in inside) / 1e6:.1f} ms, scan kernels busy {sum(k[1] - k[0] for k in sc) / 1e6:.1f} ms")
n_sl = 20
print("  per slice of the call (1/20 of its span each): H2D copies started | scan kernels started")
row_c = [0] * n_sl; row_k = [0] * n_sl
for c in inside: row_c[min(n_sl - 1, max(0, int((c[0] - a) * n_sl / (b - a))))] += 1
for k in sc: row_k[min(n_sl - 1, max(0, int((k[0] - a) * n_sl / (b - a))))] += 1
print("   copies :", " ".join(f"{x:3d}" for x in row_c))
print("   kernels:", " ".join(f"{x:3d}" for x in row_k))
PY
