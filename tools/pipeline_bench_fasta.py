#!/usr/bin/env python3
"""Host-buffer rate of the pipeline on wrapped FASTA (config-3 shape: 10 kb contigs, 80-column lines, k = 31 bit path):
FASTA text in host memory -> C++ reader -> pinned batches (line feeds deleted by the packer) -> H2D + scan."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import needletail_amd as nt

contigs, CL, W = int(os.environ.get("CONTIGS", 100_000)), 10_000, 80
rng = np.random.default_rng(3)
lines = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(contigs, CL // W, W))]
body = np.concatenate([lines, np.full((contigs, CL // W, 1), 10, dtype=np.uint8)], axis=2).reshape(contigs, -1)
hdr = np.frombuffer(b"".join(b">c%08d\n" % i for i in range(contigs)), dtype=np.uint8).reshape(contigs, 11)
text = np.concatenate([hdr, body], axis=1).tobytes()
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
out = {"contigs": contigs, "fasta_text_bytes": len(text)}
want = None
for label, kw in (("single_thread", None), ("parallel_8", 8), ("parallel_16", 16), ("parallel_32", 32)):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        if kw is None:
            rd = nt.FastxReader(data=text)
            import ctypes as C
            from needletail_amd import _lib as L
            ctx.accum_reset()
            p = L.Params(31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS, 0)
            nrec, nb = C.c_uint64(0), C.c_uint64(0)
            rc = L.lib().ntk_scan_reader(ctx._h, rd._h, C.byref(p), 16 << 20, 3, C.byref(nrec), C.byref(nb))
            assert rc == 0
            st = ctx.accum_read(); st["n_records"] = nrec.value
        else:
            st = nt.scan_file_parallel(ctx, None, 31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS, threads=kw, batch_bytes=16 << 20, data=text)
        dt = time.perf_counter() - t0
        assert st["n_records"] == contigs
        if want is None:
            want = (st["n_total"], st["sum"], st["xor"])
        assert (st["n_total"], st["sum"], st["xor"]) == want
        best = dt if best is None else min(best, dt)
    out[label] = {"seconds": round(best, 4), "Gbases_s": round(contigs * CL / best / 1e9, 2), "text_GB_s": round(len(text) / best / 1e9, 2)}
assert want[0] == contigs * (CL - 30)
print(json.dumps(out))
