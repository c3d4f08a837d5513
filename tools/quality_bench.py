"""Quality-masked scan at config-2 size: kernel time and HBM rate with the quality stream (2 B/base).
Run on the GPU box: python tools/quality_bench.py [--reads N] [--cutoff C]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import needletail_amd as nt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--cutoff", type=int, default=53)   # Phred 20
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
L, k = 150, 21
n = args.reads * (L + 1)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
qual = torch.randint(33, 75, (n + 2048,), dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, args.reads, L, 1, seq)
out = {}
for name, kw in (("plain", {}), ("quality", {"d_qual": qual, "quality_cutoff": args.cutoff})):
    for _ in range(400):   # ~0.3 s of load first: the clocks need ~40 ms to reach their steady state (see bench.py --preheat-ms)
        ctx.accum_reset(); ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, **kw)
    torch.cuda.synchronize()
    ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(args.steps):
        ctx.accum_reset(); ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, **kw)
    ms, launches = ctx.scan_time_ms(); ctx.enable_timing(False)
    ms /= launches
    st = ctx.accum_read()
    bytes_per_launch = n * (2 if kw else 1)
    out[name] = {"kernel_ms": round(ms, 4), "gbases_s": round(args.reads * L / ms / 1e6, 1),
                 "hbm_gb_s": round(bytes_per_launch / ms / 1e6, 1), "n_total": st["n_total"]}
print(json.dumps({"workload": f"{args.reads} x {L} bp, k={k}, cutoff {args.cutoff}", **out}))
