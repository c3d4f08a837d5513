"""The bit-plane compat face alone (ntk_canonical_kmers_batch_planes), 1 M x 150 bp reads in page-locked memory: seconds per call, best of 5,
next to the item-array form.  Usage (GPU box): python tools/compat_planes_bench.py [reads]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import needletail_amd as nt  # noqa: E402
from needletail_amd import _lib as L  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k, read_len = 21, 150
ctx = nt.Context(0)
rng = np.random.default_rng(5)
src = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, reads * read_len)]
src[rng.integers(0, len(src), len(src) // 1024)] = ord("N")


def pinned(n_bytes, dtype):
    p = C.c_void_p()
    L.check(L.lib().ntk_pinned_alloc(max(n_bytes, 8), C.byref(p)), "ntk_pinned_alloc")
    return np.frombuffer((C.c_uint8 * n_bytes).from_address(p.value), dtype=dtype)


flat = pinned(src.nbytes, np.uint8); flat[:] = src
offs = pinned((reads + 1) * 8, np.uint64); offs[:] = np.arange(reads + 1, dtype=np.uint64) * np.uint64(read_len)
cap_w = reads * read_len // 16 + reads + 1
rec_bit, v16, r16 = pinned((reads + 1) * 8, np.uint64), pinned(cap_w * 2, np.uint16), pinned(cap_w * 2, np.uint16)
nw, tot = C.c_uint64(0), C.c_uint64(0)
best = None
for _ in range(6):
    t0 = time.perf_counter()
    L.check(L.lib().ntk_canonical_kmers_batch_planes(ctx._h, C.cast(flat.ctypes.data, C.c_char_p), offs.ctypes.data, reads, k, rec_bit.ctypes.data,
                                                     v16.ctypes.data, r16.ctypes.data, cap_w, C.byref(nw), C.byref(tot)), "planes")
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(f"planes: {reads} reads, {tot.value} items, {best * 1e3:.2f} ms = {reads * read_len / best / 1e9:.1f} Gbases/s, {src.nbytes / best / 1e9:.1f} GB/s in, "
      f"{nw.value * 4 / best / 1e9:.1f} GB/s out")
for chunk in (4 << 20, 8 << 20, 32 << 20, 64 << 20):
    ctx.set_option(L.OPT_COMPAT_CHUNK_BYTES, chunk)
    b2 = None
    for _ in range(5):
        t0 = time.perf_counter()
        L.check(L.lib().ntk_canonical_kmers_batch_planes(ctx._h, C.cast(flat.ctypes.data, C.c_char_p), offs.ctypes.data, reads, k, rec_bit.ctypes.data,
                                                         v16.ctypes.data, r16.ctypes.data, cap_w, C.byref(nw), C.byref(tot)), "planes")
        dt = time.perf_counter() - t0
        b2 = dt if b2 is None else min(b2, dt)
    print(f"   chunk {chunk >> 20:3d} MiB: {b2 * 1e3:.2f} ms = {reads * read_len / b2 / 1e9:.1f} Gbases/s")
ctx.set_option(L.OPT_COMPAT_CHUNK_BYTES, 0)
# sequence::minimizer per record for the same batch (ntk_minimizer_batch): one upload, one wave per record, n_records x m bytes back
for m in (21, 31):
    mout = pinned(reads * m, np.uint8)
    mpos, mflg = pinned(reads * 8, np.uint64), pinned(reads, np.uint8)
    bad = C.c_uint64(0)
    b3 = None
    for _ in range(5):
        t0 = time.perf_counter()
        L.check(L.lib().ntk_minimizer_batch(ctx._h, C.cast(flat.ctypes.data, C.c_char_p), offs.ctypes.data, reads, m, mout.ctypes.data, mpos.ctypes.data,
                                            mflg.ctypes.data, C.byref(bad)), "ntk_minimizer_batch")
        dt = time.perf_counter() - t0
        b3 = dt if b3 is None else min(b3, dt)
    print(f"minimizer_batch m = {m}: {reads} reads, {b3 * 1e3:.2f} ms = {reads * read_len / b3 / 1e9:.1f} Gbases/s ({reads / b3 / 1e6:.1f} M records/s)")
