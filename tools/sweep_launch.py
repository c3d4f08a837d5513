#!/usr/bin/env python3
"""Within-process A/B of scan-kernel launch geometries on the bench workload (device-resident 10M x 150 bp)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import needletail_amd as nt

reads, L, k = int(os.environ.get("READS", 10_000_000)), 150, int(os.environ.get("K", 21))
n_bytes = reads * (L + 1)
seq = torch.empty(n_bytes + 2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
torch.cuda.synchronize()
configs = [(0, 1024), (512, 1024), (512, 512), (1024, 512), (1024, 256), (2048, 256), (4096, 256), (256, 512), (256, 256)]
modes = [("bytes_canon_norm", nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), ("bits_canon", nt.PATH_BITS_CANONICAL, nt.PRE_NONE),
         ("bits_fwd", nt.PATH_BITS, nt.PRE_NONE)]
rows = []
for rnd in range(2):
    for blocks, threads in configs:
        for name, path, pre in modes if rnd == 0 else modes[:1]:
            ctx.set_launch(blocks, threads)
            for _ in range(2):
                ctx.accum_reset(); ctx.reduce_device(seq, n_bytes, k, path, pre)
            torch.cuda.synchronize()
            ctx.scan_time_ms(); ctx.enable_timing(True)
            for _ in range(5):
                ctx.accum_reset(); ctx.reduce_device(seq, n_bytes, k, path, pre)
            ms, n = ctx.scan_time_ms(); ctx.enable_timing(False)
            row = {"round": rnd, "mode": name, "blocks": blocks, "threads": threads, "kernel_ms": round(ms / n, 4),
                   "GBps": round(n_bytes / (ms / n * 1e-3) / 1e9, 1), "Gbases_s": round(reads * L / (ms / n * 1e-3) / 1e9, 1)}
            rows.append(row)
            print(json.dumps(row), flush=True)
