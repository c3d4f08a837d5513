"""Un-normalised byte-path input on the reduce face (ntk_reduce_device, pre = NONE) on the config-2 batch: a third of the bases in lower case,
one lower-case base, none (the clean FASTQ case: the speculative packed-value scan stands) - next to the packed-value scan on the same
batch normalised.  Times are the hipEvent spans around the scan (+ the raw-byte kernel where it is queued).  python tools/raw_bytes_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import needletail_amd as nt
import oracle as O

reads, L = 10_000_000, 150
n = reads * (L + 1)
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
g = torch.Generator(device="cuda"); g.manual_seed(7)
lower = (torch.rand(n, device="cuda", generator=g) < 0.33) & (seq[:n] != 10) & (seq[:n] != ord("N"))
mixed = seq.clone(); mixed[:n][lower] |= 0x20
# a prefix against the oracle's literal chain
pre_reads = 20000
host = mixed[: pre_reads * (L + 1)].cpu().numpy().tobytes()
want = O.reduce_records(host.split(b"\n")[:pre_reads], 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
ctx.accum_reset(); ctx.reduce_device(mixed, pre_reads * (L + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
got = ctx.accum_read()
assert all(int(got[k]) == int(want[k]) for k in ("n_total", "n_fwd", "n_rc", "sum", "xor")), "prefix differs from the oracle"
one = seq.clone(); one[n - 3] |= 0x20   # ONE lower-case base at the end of the batch: the whole launch is redone
for name, buf, pre in (("raw bytes, mixed case, pre = NONE (speculation fails: scan + raw-byte kernel)", mixed, nt.PRE_NONE),
                       ("upper case with one lower-case base, pre = NONE (speculation fails)", one, nt.PRE_NONE),
                       ("upper case, pre = NONE (speculative packed-value scan; the raw-byte kernel returns at once)", seq, nt.PRE_NONE),
                       ("packed-value scan, upper case, pre = NORMALIZE", seq, nt.PRE_NORMALIZE)):
    for _ in range(3): ctx.reduce_device(buf, n, 21, nt.PATH_BYTES_CANONICAL, pre, reset=True)
    torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(5): ctx.reduce_device(buf, n, 21, nt.PATH_BYTES_CANONICAL, pre, reset=True)
    ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
    print(f"{name}: {ms / nl:.3f} ms per 1.51 GB = {n / (ms / nl * 1e-3) / 1e9:.0f} GB/s, n_total {ctx.accum_read()['n_total']}", flush=True)
