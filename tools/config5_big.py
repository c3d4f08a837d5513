"""A 41 GB FASTQ text as ONE gzip member through ntk_scan_file_parallel (no 16 GiB cliff, bounded memory).  The stream is
the 10 M-read C2 prefix REPEATS times: the text is deflated twice (once from an empty window, once with its own last 32 KiB as the dictionary:
every later repeat is byte-identical to that one), the pieces are joined with sync flushes into one member and closed by an empty final block.
Expected result = REPEATS x the resident run's (counters, histogram, sum; the xor of an odd number of equal xors).
python tools/config5_big.py [repeats]"""
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
import needletail_amd as nt

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 13
reads, RL = 10_000_000, 150
print("cgroup memory.max:", open("/sys/fs/cgroup/memory.max").read().strip() if os.path.exists("/sys/fs/cgroup/memory.max") else "?", flush=True)
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
mv = memoryview(text)
n = len(mv)
piece = 8 << 20


def body(first_dict):
    """raw deflate of the text in parallel pieces, every piece ending in a sync flush (byte-aligned, not final)"""
    def comp(a):
        d = bytes(mv[a - 32768:a]) if a >= 32768 else first_dict
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, d) if d else zlib.compressobj(6, zlib.DEFLATED, -15)
        return c.compress(mv[a:min(n, a + piece)]) + c.flush(zlib.Z_SYNC_FLUSH)
    with ThreadPoolExecutor(cpus) as ex:
        return b"".join(ex.map(comp, range(0, n, piece)))


t0 = time.perf_counter()
first, later = body(b""), body(bytes(mv[n - 32768:]))
crc = 0
for _ in range(repeats):
    crc = zlib.crc32(mv, crc)
total = n * repeats
print(f"text {total / 1e9:.1f} GB = {repeats} x {n / 1e9:.2f} GB, gzip {(len(first) + (repeats - 1) * len(later)) / 1e9:.2f} GB, built in {time.perf_counter() - t0:.0f} s", flush=True)
k, w = 21, 11
ctx.accum_reset()
ctx.reduce_device(seq, reads * (RL + 1), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w)
one = ctx.accum_read()
want = {key: one[key] * repeats for key in ("n_total", "n_fwd", "n_rc")}
want["sum"] = (one["sum"] * repeats) & (2**64 - 1)
want["xor"] = one["xor"] if repeats & 1 else 0
want["hist"] = one["hist"] * np.uint64(repeats)


def status_mb(key):
    return int(next(l for l in open("/proc/self/status") if l.startswith(key)).split()[1]) / 1024


with tempfile.TemporaryDirectory(dir="/tmp") as d:
    path = os.path.join(d, "big.fastq.gz")
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        f.write(first)
        for _ in range(repeats - 1):
            f.write(later)
        f.write(b"\x03\x00")   # an empty final block (fixed Huffman, end of block)
        f.write(struct.pack("<II", crc & 0xFFFFFFFF, total & 0xFFFFFFFF))
    gz_bytes = os.path.getsize(path)
    del first, later
    for i in range(2):
        try:
            open("/proc/self/clear_refs", "w").write("5")
        except OSError:
            pass
        rss0 = status_mb("VmRSS")
        t0 = time.perf_counter()
        st = nt.scan_file_parallel(ctx, path, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=8 << 20, w=w, streaming_fallback=False)
        dt = time.perf_counter() - t0
        ok = B.stats_equal(st, want) and st["n_records"] == reads * repeats
        g = st["gzip"]
        print(f"run {i}: {dt:.2f} s = {reads * repeats * RL / dt / 1e9:.2f} Gbases/s, {total / dt / 1e9:.1f} GB/s of text; equal to {repeats} x the resident run: {ok}; "
              f"route {g['route']} streamed {g['streamed']} members {g['members']} chunks {g['chunks']} dropped {g['chunks_dropped']} deferred {g['chunks_deferred']}; "
              f"text {g['text_bytes'] / 1e9:.1f} GB, peak text waiting for a parser {g['peak_backlog_bytes'] / 2**20:.0f} MiB; "
              f"peak RSS above the call's start {status_mb('VmHWM') - rss0:.0f} MB (the .gz file is {gz_bytes / 1e6:.0f} MB; its pages are dropped behind the decoder)", flush=True)
        assert ok
