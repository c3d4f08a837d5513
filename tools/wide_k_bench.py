"""CanonicalKmers with k = 33..255 on the reduce face (ntk_reduce_device) on the config-2 batch (10 M x 150 bp) and on a config-3 prefix
(100 k x 10 kb contigs): the packed-stream kernel (wide_canonical_reduce_kernel + the byte-walking kernel queued behind its flag) next to the
byte-walking kernel alone (NTK_ROUTE_NO_SPECULATION).  Times are the hipEvent spans the library records.  python tools/wide_k_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import needletail_amd as nt
from needletail_amd import _lib as NL

ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for name, reads, L in (("config-2 batch, 10 M x 150 bp", 10_000_000, 150), ("1.0 Gbases of 10 kb contigs", 100_000, 10_000)):
    n = reads * (L + 1)
    seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
    for k in (33, 51, 64, 127, 255):
        if k > L: continue
        row = []
        for route in (0, NL.ROUTE_NO_SPECULATION):
            ctx.set_option(NL.OPT_MINIMIZER_ROUTE, route)
            for pre in (nt.PRE_NORMALIZE, nt.PRE_NONE):
                for _ in range(2): ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, pre, reset=True)
                torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
                for _ in range(4): ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, pre, reset=True)
                ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
                st = ctx.accum_read()
                row.append((ms / nl, int(st["n_total"]), int(st["n_fwd"]), int(st["hist"].astype("uint64").sum()), hash(st["hist"].tobytes())))
        ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0)
        assert row[0][1:] == row[2][1:] and row[1][1:] == row[3][1:], (k, row)
        print(f"{name}, k = {k}: packed-stream pair {row[0][0]:.3f} ms (normalised) / {row[1][0]:.3f} ms (pre = NONE), byte-walking kernel "
              f"{row[2][0]:.3f} / {row[3][0]:.3f} ms; {n / (row[0][0] * 1e-3) / 1e9:.0f} GB/s; n_total {row[0][1]}, n_fwd {row[0][2]} (equal on both routes)", flush=True)
