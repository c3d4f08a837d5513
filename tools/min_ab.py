"""A/B timing of the generic fused minimizer kernel (and the register-fused (21, 11) build) on the config-2 batch: scan-kernel time per pass from
the library's own events, after a time-based warm-up; one line per (k, w).  NEEDLETAIL_AMD_LIB selects the build (tools/min_ab.sh loops)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import needletail_amd as nt
from needletail_amd import _lib as NL
reads, L = 10_000_000, 150
n = reads * (L + 1)
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(n + 2048, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, L, 1, seq)
pairs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(23, 11), (31, 19), (21, 11), (25, 11), (31, 11), (19, 49), (12, 5)]
ref = {}
out = []
for k, w in pairs:
    route = NL.ROUTE_NO_REGFUSED if (k, w) == (21, 11) and os.environ.get("NTK_AB_GENERIC_ON_2111") else 0
    ctx.set_option(NL.OPT_MINIMIZER_ROUTE, route)
    fn = lambda: ctx.reduce_device(seq, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, reset=True)
    t0 = time.perf_counter()
    i = 0
    while i < 3 or time.perf_counter() - t0 < 0.08:
        fn(); i += 1
        if i % 8 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
    for _ in range(8): fn()
    ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
    r = ctx.accum_read()
    out.append(f"({k},{w}) {ms / nl:.4f} ms  n_total {r['n_total']} sum {r['sum']:#x}")
print(os.path.basename(NL.LIB_PATH), " | ".join(out), flush=True)
