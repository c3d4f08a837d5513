#!/usr/bin/env python3
"""Batch-size sweep of the parallel producer at a fixed thread count (FASTQ text in host memory -> parser threads -> pinned batches -> GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import needletail_amd as nt
reads, RL, k = int(os.environ.get("READS", 10_000_000)), 150, 21
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
dev = torch.empty(reads * (RL + 1) + 1024, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, RL, 1, dev)
torch.cuda.synchronize()
seqs = dev[: reads * (RL + 1)].cpu().numpy().reshape(reads, RL + 1)
idw = 9
rec = np.empty((reads, 1 + idw + 1 + RL + 1 + 2 + RL + 1), dtype=np.uint8)
rec[:, 0] = ord("@")
rec[:, 1:1 + idw] = np.frombuffer("".join(np.char.zfill(np.arange(reads).astype(str), idw)).encode(), dtype=np.uint8).reshape(reads, idw)
rec[:, 1 + idw] = 10; rec[:, 2 + idw:2 + idw + RL] = seqs[:, :RL]; rec[:, 2 + idw + RL] = 10
rec[:, 3 + idw + RL] = ord("+"); rec[:, 4 + idw + RL] = 10; rec[:, 5 + idw + RL:5 + idw + 2 * RL] = ord("I"); rec[:, 5 + idw + 2 * RL] = 10
text = rec.tobytes(); del rec, seqs
ctx.accum_reset(); ctx.reduce_device(dev, reads * (RL + 1), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE); want = ctx.accum_read()
from needletail_amd import _lib as NL
for cs, th, bb in [(c_, t_, b_) for c_ in (1, 2) for t_ in (16, 24, 32) for b_ in (4 << 20, 8 << 20, 16 << 20, 32 << 20)]:
    if True:
        ctx.set_option(NL.OPT_COPY_STREAMS, cs)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            st = nt.scan_file_parallel(ctx, None, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=th, batch_bytes=bb, data=text)
            dt = time.perf_counter() - t0
            assert st["n_total"] == want["n_total"] and st["sum"] == want["sum"] and st["n_records"] == reads
            best = dt if best is None else min(best, dt)
        print(f"copy streams {cs} threads {th:3d} batch {bb >> 20:3d} MiB: {best * 1e3:7.1f} ms  {reads * RL / best / 1e9:6.2f} Gbases/s", flush=True)
# where the workers' time goes at the bench's setting (NTK_PIPE_STATS: per-thread phase times on stderr)
