"""The generic fused minimizer kernel on the config-3 batch (1 M x 10 kb contigs, 10 GB resident): (k, w) = (19, 19), (15, 10) [register-fused], (25, 31),
(31, 19); whole passes timed with events, the (19, 19) result compared with the two-pass path on a 1/16 prefix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import needletail_amd as nt
from needletail_amd import _lib as NL
reads, L = 1_000_000, 10_000
n = reads * (L + 1)
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0003, 0, reads, L, 1, seq)


def run(k, w, nbytes, reps=3):
    for _ in range(2):
        ctx.reduce_device(seq, nbytes, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.reduce_device(seq, nbytes, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, ctx.accum_read()


for k, w in ((19, 19), (15, 10), (25, 31), (31, 19)):
    ms, r = run(k, w, n)
    print(f"k={k:2d} w={w:2d}: {ms:8.3f} ms per 10.0 GB = {reads * L / ms / 1e6:7.1f} Gbases/s = {n / ms / 1e6:6.0f} GB/s  (windows {r['n_total']})", flush=True)
pre = (reads // 16) * (L + 1)
_, a = run(19, 19, pre, 1)
ctx.set_option(NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_TWO_PASS)
ms2, b = run(19, 19, pre, 1)
assert all(a[x] == b[x] for x in ("n_total", "n_fwd", "sum", "xor")) and (a["hist"] == b["hist"]).all()
print(f"(19, 19) on the first {reads // 16} contigs: equal to the two-pass path ({ms2:.1f} ms there)")
