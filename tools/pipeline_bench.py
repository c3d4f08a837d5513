#!/usr/bin/env python3
"""End-to-end (host-buffer, PCIe-inclusive) rate of the pinned-batch face: FASTQ text in host memory -> C++ reader ->
pinned batches -> hipMemcpyAsync (copy stream) overlapped with the scan kernels.  Also the materialise-mode rate.
Not the headline metric (bench.py times device-resident batches); DESIGN.md quotes these numbers."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import needletail_amd as nt
from needletail_amd import _lib as L
import ctypes as C

reads, RL, k = int(os.environ.get("READS", 2_000_000)), 150, 21
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
dev = torch.empty(reads * (RL + 1) + 1024, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, RL, 1, dev)
torch.cuda.synchronize()
seqs = dev[: reads * (RL + 1)].cpu().numpy().reshape(reads, RL + 1)
# FASTQ text: "@r<i>\n" + seq + "\n+\n" + 'I'*150 + "\n" with fixed-width ids so that numpy can build it
idw = 9
rec = np.empty((reads, 1 + idw + 1 + RL + 1 + 2 + RL + 1), dtype=np.uint8)
rec[:, 0] = ord("@")
ids = np.char.zfill(np.arange(reads).astype(str), idw)
rec[:, 1:1 + idw] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(reads, idw)
rec[:, 1 + idw] = 10
rec[:, 2 + idw:2 + idw + RL] = seqs[:, :RL]
rec[:, 2 + idw + RL] = 10
rec[:, 3 + idw + RL] = ord("+")
rec[:, 4 + idw + RL] = 10
rec[:, 5 + idw + RL:5 + idw + 2 * RL] = ord("I")
rec[:, 5 + idw + 2 * RL] = 10
text = rec.tobytes()
del rec

def run(batch_bytes, n_batches):
    rd = nt.FastxReader(data=text)
    ctx.accum_reset()
    p = L.Params(k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, 0)
    nrec, nb = C.c_uint64(0), C.c_uint64(0)
    t0 = time.perf_counter()
    rc = L.lib().ntk_scan_reader(ctx._h, rd._h, C.byref(p), batch_bytes, n_batches, C.byref(nrec), C.byref(nb))
    st = ctx.accum_read()
    dt = time.perf_counter() - t0
    assert rc == 0 and nrec.value == reads
    return dt, st

# reference result from the device-resident path
ctx.accum_reset(); ctx.reduce_device(dev, reads * (RL + 1), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE); want = ctx.accum_read()
out = {"reads": reads, "fastq_text_bytes": len(text)}
for bb, nbat in ((8 << 20, 3), (64 << 20, 3)):
    best = None
    for _ in range(3):
        dt, st = run(bb, nbat)
        assert st["n_total"] == want["n_total"] and st["sum"] == want["sum"] and st["xor"] == want["xor"]
        best = dt if best is None else min(best, dt)
    out[f"pipeline_batch{bb >> 20}MiB"] = {"seconds": round(best, 4), "Gbases_s": round(reads * RL / best / 1e9, 3),
                                           "fastq_GB_s": round(len(text) / best / 1e9, 3)}
for th in (8, 16, 32, 48, 64, 128):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        st = nt.scan_file_parallel(ctx, None, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=th, batch_bytes=8 << 20, data=text)
        dt = time.perf_counter() - t0
        assert st["n_total"] == want["n_total"] and st["sum"] == want["sum"] and st["xor"] == want["xor"] and st["n_records"] == reads
        best = dt if best is None else min(best, dt)
    out[f"pipeline_parallel_{th}threads"] = {"seconds": round(best, 4), "Gbases_s": round(reads * RL / best / 1e9, 3),
                                            "fastq_GB_s": round(len(text) / best / 1e9, 3)}
# parser alone (no GPU work): upper bound of the single-threaded producer
t0 = time.perf_counter(); rd = nt.FastxReader(data=text); n = 0
rec_ = L.Record()
while L.lib().ntk_reader_next(rd._h, C.byref(rec_)) == 0:
    n += 1
dt = time.perf_counter() - t0
out["parser_only"] = {"seconds": round(dt, 4), "fastq_GB_s": round(len(text) / dt / 1e9, 3)}
# materialise mode on the device-resident batch
nbytes = reads * (RL + 1)
vals = torch.empty((nbytes + 15) // 16 * 16, dtype=torch.int64, device="cuda")
v16 = torch.empty((nbytes + 15) // 16, dtype=torch.int16, device="cuda"); r16 = torch.empty_like(v16)
for _ in range(2):
    ctx.materialize_device(dev, nbytes, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, vals, v16, r16)
torch.cuda.synchronize(); ctx.scan_time_ms(); ctx.enable_timing(True)
for _ in range(5):
    ctx.materialize_device(dev, nbytes, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, vals, v16, r16)
ms, nl = ctx.scan_time_ms(); ctx.enable_timing(False)
out["materialize"] = {"kernel_ms": round(ms / nl, 4), "Gbases_s": round(reads * RL / (ms / nl * 1e-3) / 1e9, 1),
                      "GB_s_read_plus_write": round((nbytes + nbytes * 8 + nbytes / 4) / (ms / nl * 1e-3) / 1e9, 1)}
# windowed minimizers (w = 11, k = 21) on the device-resident batch: materialise + window-min + fold
del vals, v16, r16
torch.cuda.empty_cache()
for _ in range(2):
    ctx.accum_reset(); ctx.reduce_device(dev, nbytes, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ctx.accum_reset(); ctx.reduce_device(dev, nbytes, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
out["minimizers_w11_resident"] = {"ms": round(dt * 1e3, 3), "Gbases_s": round(reads * RL / dt / 1e9, 1)}

# BASELINE.json configs[4]: gzip FASTQ stream (zlib level 6) + minimizers (w=11, k=21); inflate + parse on the CPU thread,
# overlapped with H2D + kernels.  Bounded to 1 M reads so that compressing the fixture stays short.
import gzip
import tempfile
gz_reads = min(reads, 1_000_000)
rec_bytes = len(text) // reads
with tempfile.NamedTemporaryFile(suffix=".fq.gz", delete=False) as f:
    f.write(gzip.compress(text[: gz_reads * rec_bytes], compresslevel=6))
    gz_path = f.name
gz_size = os.path.getsize(gz_path)
best = None
for _ in range(2):
    t0 = time.perf_counter()
    st = nt.scan_file(ctx, gz_path, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=8 << 20, w=11)
    dt = time.perf_counter() - t0
    assert st["n_records"] == gz_reads
    best = dt if best is None else min(best, dt)
best_par = None
for _ in range(2):   # the same file through the parallel producer: libdeflate inflates it into memory, then parallel parsing
    t0 = time.perf_counter()
    try:
        st = nt.scan_file_parallel(ctx, gz_path, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=16, batch_bytes=8 << 20, w=11)
    except nt.NtkError:
        break
    dt = time.perf_counter() - t0
    assert st["n_records"] == gz_reads
    best_par = dt if best_par is None else min(best_par, dt)
# block gzip (BGZF, what bgzip writes): members are located by their size fields and inflated by 32 threads
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _fastx import bgzf_compress
with tempfile.NamedTemporaryFile(suffix=".fq.bgz", delete=False) as f:
    f.write(bgzf_compress(text[: gz_reads * rec_bytes], block=65000))
    bgz_path = f.name
best_bgz = None
for _ in range(3):
    t0 = time.perf_counter()
    try:
        st = nt.scan_file_parallel(ctx, bgz_path, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=32, batch_bytes=8 << 20, w=11)
    except nt.NtkError:
        break
    dt = time.perf_counter() - t0
    assert st["n_records"] == gz_reads
    best_bgz = dt if best_bgz is None else min(best_bgz, dt)
bgz_size = os.path.getsize(bgz_path)
os.unlink(bgz_path)
t0 = time.perf_counter()
with gzip.open(gz_path, "rb") as g:
    while g.read(1 << 24):
        pass
inflate_s = time.perf_counter() - t0
os.unlink(gz_path)
out["config5_gzip_minimizers"] = {"reads": gz_reads, "gz_bytes": gz_size, "seconds": round(best, 4),
                                  "Gbases_s": round(gz_reads * RL / best / 1e9, 3),
                                  "inflate_only_seconds_python_zlib": round(inflate_s, 4),
                                  "libdeflate_then_parallel_seconds": round(best_par, 4) if best_par else None,
                                  "libdeflate_then_parallel_Gbases_s": round(gz_reads * RL / best_par / 1e9, 3) if best_par else None,
                                  "bgzf_bytes": bgz_size,
                                  "bgzf_parallel_inflate_seconds": round(best_bgz, 4) if best_bgz else None,
                                  "bgzf_parallel_inflate_Gbases_s": round(gz_reads * RL / best_bgz / 1e9, 3) if best_bgz else None}
print(json.dumps(out))
