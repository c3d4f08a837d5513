"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards by records with no data-path collective: every rank scans its own record batches into its
own device accumulators (NTK_ACC_WORDS u64 words).  The only exchange is ONE all-reduce (sum) of that small
buffer at the end of a pass.  The xor digest cannot be summed, so the fold kernel also keeps it as 64 bit
counters (NTK_ACC_XOR_BITS) whose parities survive a sum."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import ACC_HIST, ACC_N_FWD, ACC_N_RC, ACC_N_TOTAL, ACC_SUM, ACC_WORDS, ACC_XOR, ACC_XOR_BITS, HIST_BINS

COMM_ID_BYTES = 128
BATCH_RECORDS = 1 << 20   # record i -> GPU (i / BATCH_RECORDS) mod n_gpus


def round_robin_batches(n_records: int, rank: int, world: int, batch_records: int = BATCH_RECORDS):
    """[(first_record, n_records), ...] of the record batches `rank` owns when batches of `batch_records` records are
    dealt round-robin to `world` GPUs (config 4 of BASELINE.json)."""
    if not (0 <= rank < world) or batch_records < 1:
        raise ValueError("bad rank / batch size")
    out = []
    n_batches = (n_records + batch_records - 1) // batch_records
    for b in range(rank, n_batches, world):
        first = b * batch_records
        out.append((first, min(batch_records, n_records - first)))
    return out


class Communicator:
    """RCCL communicator behind the C ABI (ntk_comm_*): the accumulators of all ranks are summed by ONE ncclAllReduce
    (ncclUint64, ncclSum, NTK_ACC_WORDS words) on each context's stream.

    `Communicator.for_rank(ctx, n_ranks, rank, id_bytes)`: one process per GPU; rank 0 makes `unique_id()` and ships it.
    `Communicator.all_local(ctxs)`: one process driving several GPUs (ncclCommInitAll)."""

    def __init__(self, handle, ctxs):
        self._h = handle
        self._ctxs = list(ctxs)   # keep the contexts alive

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        L.check(L.lib().ntk_comm_unique_id(buf), "ntk_comm_unique_id")
        return buf.raw

    @classmethod
    def for_rank(cls, ctx, n_ranks: int, rank: int, id_bytes: bytes) -> "Communicator":
        if len(id_bytes) != COMM_ID_BYTES:
            raise ValueError("the communicator id has 128 bytes")
        h = C.c_void_p()
        L.check(L.lib().ntk_comm_init_rank(ctx._h, n_ranks, rank, id_bytes, C.byref(h)), "ntk_comm_init_rank")
        return cls(h, [ctx])

    @classmethod
    def all_local(cls, ctxs) -> "Communicator":
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        h = C.c_void_p()
        L.check(L.lib().ntk_comm_init_all(arr, len(ctxs), C.byref(h)), "ntk_comm_init_all")
        return cls(h, ctxs)

    @property
    def size(self) -> int:
        return int(L.lib().ntk_comm_size(self._h))

    def allreduce_accumulators(self):
        L.check(L.lib().ntk_allreduce_accumulators(self._h), "ntk_allreduce_accumulators")

    def allreduce_time_ms(self):
        """(total ms, calls) of the all-reduces issued while the first local ctx had timing enabled, since the last call
        (ntk_comm_allreduce_time_ms: hipEvents on the stream the collective runs on)."""
        ms, n = C.c_double(0), C.c_uint64(0)
        L.check(L.lib().ntk_comm_allreduce_time_ms(self._h, C.byref(ms), C.byref(n)), "ntk_comm_allreduce_time_ms")
        return ms.value, n.value

    def close(self):
        if self._h:
            L.lib().ntk_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [begin, end) of `n_items` records (or record batches) owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_accumulators(acc, group=None):
    """In-place sum all-reduce of an accumulator tensor (int64[ACC_WORDS]) over the process group."""
    import torch.distributed as dist
    if acc.numel() != ACC_WORDS:
        raise ValueError(f"accumulator tensor must have {ACC_WORDS} words")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc


def decode_accumulators(acc) -> dict:
    """Host view of an accumulator buffer (after any all-reduce): the xor digest is rebuilt from its bit counters."""
    a = acc.detach().cpu().numpy() if hasattr(acc, "detach") else np.asarray(acc)
    a = a.view(np.uint64)
    xr = 0
    for i in range(64):
        xr |= (int(a[ACC_XOR_BITS + i]) & 1) << i
    return {"n_total": int(a[ACC_N_TOTAL]), "n_fwd": int(a[ACC_N_FWD]), "n_rc": int(a[ACC_N_RC]),
            "sum": int(a[ACC_SUM]), "xor": xr, "hist": a[ACC_HIST: ACC_HIST + HIST_BINS].copy()}


def encode_accumulators(stats: dict) -> np.ndarray:
    """Inverse of decode (a single rank's result in accumulator layout) - used by the CPU multi-process tests."""
    a = np.zeros(ACC_WORDS, dtype=np.uint64)
    a[ACC_N_TOTAL], a[ACC_N_FWD], a[ACC_N_RC] = stats["n_total"], stats["n_fwd"], stats["n_rc"]
    a[ACC_SUM], a[ACC_XOR] = stats["sum"], stats["xor"]
    a[ACC_HIST: ACC_HIST + HIST_BINS] = stats["hist"]
    for i in range(64):
        a[ACC_XOR_BITS + i] = (stats["xor"] >> i) & 1
    return a.view(np.int64)
