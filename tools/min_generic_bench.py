"""Windowed minimizers outside the register-fused grid, config-2 batch (10 M x 150 bp): the generic fused kernel (run-time k and w, one pass)
against the two-pass path (materialise + window-min, NTK_OPT_MINIMIZER_ROUTE = NTK_ROUTE_NO_REGFUSED | NTK_ROUTE_NO_GENERIC), whole passes timed with events on the ctx stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import needletail_amd as nt
from needletail_amd import _lib as NL
reads, L = 10_000_000, 150
n = reads * (L + 1)
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)


def run(k, w, reps=5):
    for _ in range(2):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, ctx.accum_read()


pairs = [(23, 11), (31, 11), (21, 19), (31, 19), (15, 25), (31, 31), (19, 49), (21, 11), (12, 5)]
for k, w in pairs:
    ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0)
    ms, r = run(k, w)
    line = f"k={k:2d} w={w:2d}: default path {ms:7.3f} ms = {reads * L / ms / 1e6:7.1f} Gbases/s"
    if (k, w) == (21, 11):
        ctx.set_option(NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_NO_REGFUSED)     # skip the register-fused build: the generic kernel on a pair that has one
        ms_g, r_g = run(k, w)
        ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0)
        assert all(r[x] == r_g[x] for x in ("n_total", "n_fwd", "sum", "xor"))
        line += f"   generic kernel on the same pair {ms_g:7.3f} ms"
    ctx.set_option(NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_TWO_PASS)
    ms2, r2 = run(k, w, 3)
    assert all(r[x] == r2[x] for x in ("n_total", "n_fwd", "sum", "xor")), (k, w)
    print(line + f"   two-pass {ms2:7.3f} ms   (n_total {r['n_total']}, equal results)", flush=True)
