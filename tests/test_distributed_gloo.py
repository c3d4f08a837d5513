"""world_size-2 (and 3) gloo test of the multi-GPU host path on CPU: record sharding + the single sum
all-reduce of the accumulator buffer + decoding (xor via bit counters).  Per-rank accumulators are produced by
the oracle here (no GPU in this container); on the GPU box the same functions run over RCCL in bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
from needletail_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_reads, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = D.shard_range(n_reads, rank, world)
        buf = O.synth_reads(0x5EED0004, b, e - b, 150, 4)
        st = O.reduce_fused(buf, 21, True, True, True)
        acc = torch.from_numpy(D.encode_accumulators(st).copy())
        D.allreduce_accumulators(acc)
        got = D.decode_accumulators(acc)
        if rank == 0:
            ret.put({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in got.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reduce_equals_whole(world):
    n_reads = 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = O.reduce_fused(O.synth_reads(0x5EED0004, 0, n_reads, 150, 4), 21, True, True, True)
    for k in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
        assert got[k] == whole[k], k
    assert np.array_equal(np.array(got["hist"], dtype=np.uint64), whole["hist"])


def test_shard_range_partitions():
    for n in (0, 1, 7, 100, 12_500_000):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)


def test_encode_decode_roundtrip():
    st = O.reduce_fused(O.synth_reads(1, 0, 50, 150, 8), 21, True, True, True)
    got = D.decode_accumulators(D.encode_accumulators(st))
    assert got["xor"] == st["xor"] and got["sum"] == st["sum"] and np.array_equal(got["hist"], st["hist"])
