import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, needletail_amd as nt
reads, L = 10_000_000, 150
n = reads * (L + 1)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
for k, w in ((21, 11), (17, 11), (21, 9), (21, 12)):
    for _ in range(100):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(30):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
    r = ctx.accum_read()
    print(sys.argv[1] if len(sys.argv) > 1 else "", k, w, round(ms / nl, 4), "ms", r["n_total"], hex(r["sum"]))
