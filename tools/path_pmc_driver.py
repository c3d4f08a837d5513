"""A few launches of each reduce-mode kernel family on the config-2 batch (run under rocprofv3 --pmc by tools/path_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, needletail_amd as nt
reads, L = 10_000_000, 150
n = reads * (L + 1)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
qual = torch.full((n + 2048,), 73, dtype=torch.uint8, device="cuda"); qual[::7] = 34
for path, pre, ks in ((nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, (4, 11, 16, 21, 23, 31)), (nt.PATH_BITS, nt.PRE_NONE, (4, 16, 21, 31))):
    for k in ks:
        for _ in range(4):
            ctx.reduce_device(seq, n, k, path, pre, reset=True)
for _ in range(4):
    ctx.reduce_device(seq, n, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, d_qual=qual, quality_cutoff=35, reset=True)
# (23, 11) / (31, 11): the generic fused minimizer kernel, f64 keys / general keys (two instantiations, w = 11: 61 emitting lanes per tile)
for k, w in ((21, 11), (15, 10), (23, 11), (31, 11)):
    for _ in range(4):
        ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
torch.cuda.synchronize()
