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
        # config-4 partition: record batches dealt round-robin (record i -> rank (i / batch) mod world), SURVEY.md 8d
        parts = [O.reduce_fused(O.synth_reads(0x5EED0004, first, n, 150, 4), 21, True, True, True)
                 for first, n in D.round_robin_batches(n_reads, rank, world, batch_records=64)]
        st = {"n_total": 0, "n_fwd": 0, "n_rc": 0, "sum": 0, "xor": 0, "hist": np.zeros(4096, dtype=np.uint64)}
        for q in parts:
            for key in ("n_total", "n_fwd", "n_rc"):
                st[key] += q[key]
            st["sum"] = (st["sum"] + q["sum"]) & (2 ** 64 - 1)
            st["xor"] ^= q["xor"]
            st["hist"] = st["hist"] + q["hist"]
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


def test_round_robin_batches_partition_the_records():
    for n, world, batch in ((1001, 2, 64), (1 << 12, 8, 1 << 9), (5, 3, 2), (0, 2, 4), (100, 4, 1000)):
        seen = []
        for r in range(world):
            got = D.round_robin_batches(n, r, world, batch)
            for first, cnt in got:
                assert cnt >= 1 and first % batch == 0 and (first // batch) % world == r
                seen.extend(range(first, first + cnt))
        assert sorted(seen) == list(range(n))
    with pytest.raises(ValueError):
        D.round_robin_batches(10, 2, 2)
    b, e = D.shard_range(10, 1, 3)
    assert (b, e) == (4, 7)


def test_bench_fails_fast_when_the_rendezvous_cannot_happen():
    """bench.py N > 1 start-up has deadlines: a rank whose MASTER_PORT nobody listens on must exit
    non-zero within seconds, naming itself and the phase - not sit in the rendezvous until the driver's lease runs out.
    (gloo, no GPU needed: the process group is joined before any device is touched.)"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                        "--init-timeout-s", "4", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=170)
    took = time.time() - t0
    assert r.returncode != 0 and took < 120, (r.returncode, took)
    assert "rank 1" in r.stderr and "phase 'process-group rendezvous (gloo)'" in r.stderr, r.stderr[-1500:]
    assert r.stdout.strip() == ""   # no JSON line from a run that did not happen


def test_bench_refuses_more_gpus_than_are_visible():
    """`python bench.py --gpus N` with fewer than N usable devices is an immediate, named error (no launcher, no hang)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 64 but" in r.stderr and "usable gfx950 device(s) are visible" in r.stderr, r.stderr[-800:]


def test_bench_reference_crate_harness_protocol(tmp_path):
    """bench.py's `--cpu-reference-bin` path (prefer the real crate when cargo + a vendored registry exist; the
    harness is rust/cpu_baseline, source only here): the reads go in as a file of fixed-stride records, one JSON line comes back.
    A stand-in executable that speaks the same protocol through the oracle exercises the plumbing and the toolchain note."""
    import json
    import stat
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "fake_ntk_cpu_baseline"
    fake.write_text(f"""#!{sys.executable}
import json, sys, time
sys.path.insert(0, {root!r})
import numpy as np
import oracle as O
path, n_reads, read_len, k, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
buf = np.fromfile(path, dtype=np.uint8)[: n_reads * (read_len + 1)]
offs = np.arange(n_reads + 1, dtype=np.uint64) * (read_len + 1)
t0 = time.time()
st = O.reduce_batch(buf, offs, 1, k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, threads)
print(json.dumps({{"n_total": st["n_total"], "n_fwd": st["n_fwd"], "sum": st["sum"], "xor": st["xor"], "seconds": time.time() - t0}}))
""")
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    n_reads, read_len, k = 3000, 150, 21
    host = O.synth_reads(0x5EED0002, 0, n_reads, read_len, 1)
    got = bench.cpu_reference_run(str(fake), host, n_reads, read_len, k, 2)
    want = O.reduce_fused(host, k, True, True, True)
    assert all(int(got[x]) == int(want[x]) for x in ("n_total", "n_fwd", "sum", "xor")) and got["seconds"] > 0
    note = bench.reference_toolchain_note()
    assert note.startswith("absent (") and "cargo" in note   # this image has no cargo: the C port is the baseline, and the line says why
    eff, info = bench.effective_cpus()
    assert 1 <= eff <= (os.cpu_count() or 1) and info["logical_cpus"] == os.cpu_count()
