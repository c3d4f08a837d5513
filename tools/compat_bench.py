"""The batched compat face alone (ntk_canonical_kmers_batch: the items of Sequence::canonical_kmers for a reader batch in one call),
1 M x 150 bp records, k = 21, page-locked arrays in and out.  Run under
    rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/<tag>/compat_trace -o p -- python tools/compat_bench.py
to see what the call's time is made of (tools/compat_trace_summary.py reads the trace)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import needletail_amd as nt
from needletail_amd import _lib as L

reads, read_len, k = int(os.environ.get("READS", "1000000")), 150, 21
ctx = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = torch.empty(reads * (read_len + 1) + 2048, dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(0x5EED0002, 0, reads, read_len, 1, seq)
host = seq[: reads * (read_len + 1)].cpu().numpy().reshape(reads, read_len + 1)
cap = reads * (read_len - k + 1)

def pinned(n_bytes, dtype):
    p = C.c_void_p()
    L.check(L.lib().ntk_pinned_alloc(max(n_bytes, 8), C.byref(p)), "ntk_pinned_alloc")
    return np.frombuffer((C.c_uint8 * n_bytes).from_address(p.value), dtype=dtype)

flat = pinned(reads * read_len, np.uint8); flat[:] = np.ascontiguousarray(host[:, :read_len]).reshape(-1)
offs = pinned((reads + 1) * 8, np.uint64); offs[:] = np.arange(reads + 1, dtype=np.uint64) * np.uint64(read_len)
counts, pos, flg = pinned(reads * 8, np.uint64), pinned(cap * 8, np.uint64), pinned(cap, np.uint8)
tot = C.c_uint64(0)
times = []
for rep in range(int(os.environ.get("REPS", "6"))):
    t0 = time.perf_counter()
    L.check(L.lib().ntk_canonical_kmers_batch(ctx._h, C.cast(flat.ctypes.data, C.c_char_p), offs.ctypes.data, reads, k, counts.ctypes.data,
                                              pos.ctypes.data, flg.ctypes.data, cap, C.byref(tot)), "ntk_canonical_kmers_batch")
    times.append(time.perf_counter() - t0)
best = min(times)
print(f"records {reads} items {tot.value} best {best * 1e3:.2f} ms of {[round(t * 1e3, 2) for t in times]} -> {reads * read_len / best / 1e9:.2f} Gbases/s, "
      f"{tot.value * 9 / best / 1e9:.1f} GB/s of items")
