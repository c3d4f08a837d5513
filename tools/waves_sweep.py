"""Block size (= waves per SIMD at two blocks per CU) per reduce build on the config-2 batch: 768 threads (6 waves per SIMD, the library's choice
for every build) against 896 / 1024 (7 / 8 waves: only builds with <= 73 / <= 64 VGPRs get two such blocks per CU).  python tools/waves_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, needletail_amd as nt
reads, L = 10_000_000, 150
n = reads * (L + 1)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
def ms(k, path, pre, threads, w=0):
    ctx.set_launch(0, threads)
    for _ in range(30): ctx.reduce_device(seq, n, k, path, pre, reset=True, w=w)
    torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(20): ctx.reduce_device(seq, n, k, path, pre, reset=True, w=w)
    t, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
    return t / nl
print("build                         768      896      1024   (ms per 1.51 GB)")
for name, k, path, pre in [("canonical bytes k=%d" % k, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE) for k in (4, 11, 16, 17, 21, 23, 24, 27, 31, 32)] + \
                          [("canonical bits k=%d" % k, k, nt.PATH_BITS_CANONICAL, nt.PRE_NONE) for k in (21, 31)] + \
                          [("forward-only k=%d" % k, k, nt.PATH_BITS, nt.PRE_NONE) for k in (4, 16, 21, 31)]:
    row = [ms(k, path, pre, t) for t in (768, 896, 1024)]
    print(f"{name:28s} " + " ".join(f"{x:8.4f}" for x in row), flush=True)
ctx.set_launch(0, 0)
