"""Summary of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/compat_bench.py (the batched compat face).  On this stack the
D2H copies into page-locked memory run as blit kernels (__amd_rocclr_copyBuffer in the kernel trace, not SDMA rows of the copy trace):
the "copy engine" of this face is that kernel.  Prints, for the LAST call of the run: its GPU span, the time each kernel had, and how
much of the span the item copies (copyBuffer launches > 0.1 ms) kept the link busy."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
if not rows:
    print("no kernel trace under", d); sys.exit(0)
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
# the last call = the last 1/REPS of the launches that belong to the calls (from the first canonical_bytes_kernel on); argv[2] = REPS
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
first = next(i for i, x in enumerate(ks) if "canonical_bytes_kernel" in x[2])
calls = ks[first - 4 if first >= 4 else 0:]   # (the four uploads in front of the first scan)
last = calls[len(calls) - len(calls) // reps:]
t0, t1 = last[0][0], max(e for _, e, _ in last)
by, cnt = {}, {}
for s, e, n in last:
    by[n] = by.get(n, 0) + (e - s); cnt[n] = cnt.get(n, 0) + 1
item = [(s, e) for s, e, n in last if n == "__amd_rocclr_copyBuffer" and e - s > 100_000]
busy = sum(e - s for s, e in item)
print(f"last call: GPU span {(t1 - t0) / 1e6:.2f} ms, {len(last)} launches")
for n, t in sorted(by.items(), key=lambda x: -x[1]):
    print(f"  {n:48s} x{cnt[n]:3d} {t / 1e6:8.3f} ms")
print(f"item copies (D2H blit kernels > 0.1 ms): {len(item)} launches, {busy / 1e6:.2f} ms = {100.0 * busy / (t1 - t0):.0f} % of the call's GPU span"
      f" ({100.0 * busy / (item[-1][1] - item[0][0]):.0f} % of the span from the first item copy to the last)")
