"""configs[4] end to end, phase by phase: where the time of secondary.config5_gzip_minimizers goes beyond ntk_gunzip itself
(the inflate, the parse + H2D + scan of the inflated text, the buffers in between).  python tools/config5_phases.py [reads]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
import needletail_amd as nt
from needletail_amd import _lib as L

reads, RL = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 150
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(reads * (RL + 1) + 2048, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, RL, 1, seq)
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
print(f"reads {reads}, text {len(text) / 1e9:.2f} GB, gzip {len(gz) / 1e6:.0f} MB, cpus {cpus}", flush=True)
k, w = 21, 11


def best(f, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts), ts


with tempfile.TemporaryDirectory(dir="/tmp") as d:
    path = os.path.join(d, "c5.fastq.gz")
    open(path, "wb").write(gz)
    tpath = os.path.join(d, "c5.fastq")
    open(tpath, "wb").write(text)
    def rss_mb():
        return int(next(l for l in open("/proc/self/status") if l.startswith("VmRSS")).split()[1]) / 1024
    def hwm_mb():
        return int(next(l for l in open("/proc/self/status") if l.startswith("VmHWM")).split()[1]) / 1024
    last = {}
    def run_gz(bb, threads):
        try: open("/proc/self/clear_refs", "w").write("5")   # reset the resident-set high-water mark
        except OSError: pass
        r0 = rss_mb()
        st = nt.scan_file_parallel(ctx, path, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=threads, batch_bytes=bb, w=w, streaming_fallback=False)
        last.update(st["gzip"]); last["rss_peak_above_start_MB"] = round(hwm_mb() - r0, 1)
    for threads in sorted({cpus, cpus + 4, 2 * cpus}):
        for bb in (4 << 20, 8 << 20):
            t, ts = best(lambda: run_gz(bb, threads))
            print(f"gz file   -> scan_file_parallel (streamed), {threads} inflate threads + {last['parse_threads']} parsers, batch {bb >> 20} MiB: {t:.3f} s = {reads * RL / t / 1e9:.2f} Gbases/s  "
                  f"{['%.3f' % x for x in ts]}  route {last['route']} streamed {last['streamed']} first batch after {last['first_batch_s']:.3f} s, peak backlog {last['peak_backlog_bytes'] / 2**20:.0f} MiB, "
                  f"peak RSS above start {last['rss_peak_above_start_MB']:.0f} MB, chunks {last['chunks']} dropped {last['chunks_dropped']} deferred {last['chunks_deferred']}, "
                  f"decode cpu {last['decode_busy_s']:.2f} s, resolve cpu {last['resolve_busy_s']:.2f} s, search {last['search_s']:.3f} s", flush=True)
    t, ts = best(lambda: nt.scan_file_parallel(ctx, tpath, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=4 << 20, w=w))
    print(f"text file -> scan_file_parallel (mmap of the page cache): {t:.3f} s = {reads * RL / t / 1e9:.2f} Gbases/s  {['%.3f' % x for x in ts]}", flush=True)
    t, ts = best(lambda: nt.scan_file_parallel(ctx, None, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=4 << 20, w=w, data=text))
    print(f"text in memory -> scan_buffer_parallel: {t:.3f} s = {reads * RL / t / 1e9:.2f} Gbases/s  {['%.3f' % x for x in ts]}", flush=True)

    def gunzip_keep():
        o, n, info = C.c_void_p(), C.c_uint64(0), L.GunzipInfo()
        L.check(L.lib().ntk_gunzip(gz, len(gz), cpus, C.byref(o), C.byref(n), C.byref(info)), "ntk_gunzip")
        return o, n.value, info
    t0 = time.perf_counter(); o, n, info = gunzip_keep(); t1 = time.perf_counter()
    buf = (C.c_char * n).from_address(o.value)
    t2 = time.perf_counter()
    st = nt.scan_file_parallel(ctx, None, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=4 << 20, w=w, data=buf)
    t3 = time.perf_counter()
    L.lib().ntk_gunzip_free(o, n)
    t4 = time.perf_counter()
    print(f"by hand: ntk_gunzip {t1 - t0:.3f} s (search {info.search_s:.3f}, decode wall {info.decode_s:.3f}, crc {info.crc_s:.3f}), scan of the fresh buffer {t3 - t2:.3f} s, free {t4 - t3:.3f} s", flush=True)
