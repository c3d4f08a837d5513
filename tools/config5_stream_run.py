"""configs[4] through the streamed gzip route, for the copy / kernel timeline (tools/config5_trace.sh runs this under rocprofv3
--kernel-trace --memory-copy-trace): the 10 M-read C2 prefix as ONE zlib-6 member -> ntk_scan_file_parallel with (21, 11) minimizers, three
calls; prints each call's wall span and what the gzip front-end reports.  python tools/config5_stream_run.py [reads]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
import needletail_amd as nt

reads, RL = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 150
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(reads * (RL + 1) + 2048, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, RL, 1, seq)
text = B.fastq_text_of(seq, reads, RL) if hasattr(B, "fastq_text_of") else None
if text is None:
    seqs = seq[: reads * (RL + 1)].cpu().numpy().reshape(reads, RL + 1)
    idw = 9
    rec = np.empty((reads, 1 + idw + 1 + RL + 1 + 2 + RL + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    ids = np.arange(reads, dtype=np.int64)
    for d_ in range(idw):
        rec[:, 1 + d_] = (ids // 10 ** (idw - 1 - d_)) % 10 + 48
    rec[:, 1 + idw] = 10
    rec[:, 2 + idw:2 + idw + RL] = seqs[:, :RL]
    rec[:, 2 + idw + RL] = 10
    rec[:, 3 + idw + RL] = ord("+")
    rec[:, 4 + idw + RL] = 10
    rec[:, 5 + idw + RL:5 + idw + 2 * RL] = ord("I")
    rec[:, 5 + idw + 2 * RL] = 10
    text = rec.tobytes()
    del rec, seqs
cpus, _ = B.effective_cpus()
gz = B.gzip_one_member(text, cpus)
ctx.accum_reset()
ctx.reduce_device(seq, reads * (RL + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
want = ctx.accum_read()
print(f"reads {reads}, text {len(text) / 1e9:.2f} GB, gzip {len(gz) / 1e6:.0f} MB, cpus {cpus}", flush=True)
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    path = os.path.join(d, "c5.fastq.gz")
    open(path, "wb").write(gz)
    if os.environ.get("C5_STATS"):
        from needletail_amd import _lib as NL
        ctx.set_option(NL.OPT_PIPE_STATS, 1)
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        st = nt.scan_file_parallel(ctx, path, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=8 << 20, w=11, streaming_fallback=False)
        t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        assert B.stats_equal(st, want) and st["n_records"] == reads
        g = st["gzip"]
        print(f"call {i}: monotonic_ns {t0} .. {t1} = {(t1 - t0) / 1e6:.1f} ms = {reads * RL / (t1 - t0):.2f} Gbases/s; route {g['route']} streamed {g['streamed']} "
              f"first batch after {g['first_batch_s'] * 1e3:.1f} ms, peak backlog {g['peak_backlog_bytes'] / 2**20:.0f} MiB, equal to the resident run", flush=True)
        time.sleep(0.3)
