import os, sys
sys.path.insert(0, "/root/repo")
import torch, needletail_amd as nt
reads, L = 10_000_000, 150
n = reads * (L + 1)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
for k in (33, 64, 100, 150):
    for _ in range(3): ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, reset=True)
    torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(5): ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, reset=True)
    t, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
    print(k, round(t / nl, 3), "ms", ctx.accum_read()["n_total"], flush=True)
