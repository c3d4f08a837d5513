"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards by records with no data-path collective: every rank scans its own record batches into its
own device accumulators (NTK_ACC_WORDS u64 words).  The only exchange is ONE all-reduce (sum) of that small
buffer at the end of a pass.  The xor digest cannot be summed, so the fold kernel also keeps it as 64 bit
counters (NTK_ACC_XOR_BITS) whose parities survive a sum."""
from __future__ import annotations

import numpy as np

from ._lib import ACC_HIST, ACC_N_FWD, ACC_N_RC, ACC_N_TOTAL, ACC_SUM, ACC_WORDS, ACC_XOR, ACC_XOR_BITS, HIST_BINS


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
