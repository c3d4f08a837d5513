"""Fused-minimizer grid on the config-2 batch: kernel ms per (k, w) of the fused builds (k = 15..23 x w = 9..12),
a prefix checked against the oracle first; the quality-masked builds of (21, 11) and (15, 10) next to them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, needletail_amd as nt
import oracle as O  # checker
reads, L = 10_000_000, 150
n = reads * (L + 1)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
pre_n = 3000 * (L + 1)
host = seq[:pre_n].cpu().numpy().tobytes()
qual = torch.full((n + 2048,), 73, dtype=torch.uint8, device="cuda")
qual[::7] = 34
hq = qual[:pre_n].cpu().numpy().tobytes()
def run(k, w, q):
    kw = dict(d_qual=qual, quality_cutoff=35) if q else {}
    ctx.reduce_device(seq, pre_n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True, **kw)
    got = ctx.accum_read()
    want = O.minimizers_reduce(O.quality_mask(host, hq, 35) if q else host, k, w, True, True)
    ok = all(int(got[x]) == int(want[x]) for x in ("n_total", "n_fwd", "n_rc", "sum", "xor")) and (got["hist"] == want["hist"]).all()
    for _ in range(40):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True, **kw)
    torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(20):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True, **kw)
    ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
    print(f"k={k:2d} w={w:2d} {'quality-masked ' if q else ''}{ms / 20:.4f} ms per pass ({nl // 20} launch(es))  prefix {'== oracle' if ok else 'DIFFERS FROM THE ORACLE'}", flush=True)
for k in range(15, 24):
    for w in (5, 9, 10, 11, 12):
        run(k, w, False)   # (22, 12), (23, 11), (23, 12): windows of 33 / 34 bytes, the builds with three halo lanes (round 5)
run(21, 11, True); run(15, 10, True)
run(24, 11, False)   # no register-fused build: the generic fused kernel (tools/min_generic_bench.py times it against the two-pass path)
