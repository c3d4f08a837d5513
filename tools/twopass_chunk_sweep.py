"""Two-pass windowed minimizers (materialise + window-min; every (k, w) without a fused build): time per pass over the config-2 batch
against the chunk size (NTK_MINIMIZER_CHUNK_BYTES: input bytes per pass; the value scratch is 8 x that).  Does a scratch that fits the
256 MB memory-side cache take the 8 B per position off the HBM?  torch events on the ctx stream (whole passes, both kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
reads, L = 10_000_000, 150
n = reads * (L + 1)
for chunk_mib in (256, 64, 32, 16, 8, 4):
    os.environ["NTK_MINIMIZER_CHUNK_BYTES"] = str(chunk_mib << 20)
    import importlib
    import needletail_amd as nt
    ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)   # the chunk size is read when a ctx is made
    seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
    for k, w in ((23, 11), (31, 11), (21, 19)):
        for _ in range(3):
            ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
        e1.record(); torch.cuda.synchronize()
        r = ctx.accum_read()
        print(f"chunk {chunk_mib:4d} MiB  k={k} w={w}: {e0.elapsed_time(e1) / 5:.3f} ms per pass  (n_total {r['n_total']}, sum {r['sum']:#x})", flush=True)
    ctx.close(); del seq
