import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, needletail_amd as nt
reads, L = 10_000_000, 150
n = reads*(L+1)
seq = torch.empty(n+2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
for name, path, pre in (("bytes_canonical", nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), ("bits_canonical", nt.PATH_BITS_CANONICAL, nt.PRE_NONE), ("bits_forward", nt.PATH_BITS, nt.PRE_NONE)):
    for k in (tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (4, 11, 16, 17, 21, 22, 23, 27, 31, 32)):
        for _ in range(150):
            ctx.accum_reset(); ctx.reduce_device(seq, n, k, path, pre)
        torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
        for _ in range(30):
            ctx.accum_reset(); ctx.reduce_device(seq, n, k, path, pre)
        ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
        print(name, k, round(ms/nl, 4), "ms", round(n/(ms/nl)/1e6, 1), "GB/s")
