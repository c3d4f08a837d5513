#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

A "step" is one pass of the hot path (canonical k=21 extraction, reduce mode: counters + 4096-bin prefix
histogram + sum/xor digests of every canonical k-mer) over one batch of synthetic 150 bp reads that is
already resident in HBM (configs[1]: 10 M reads per GPU, SplitMix64 seed 0x5EED0002, N rate 1/1024), plus, for
N > 1, the single RCCL all-reduce of the histogram/counters over xGMI.  Records shard across ranks with no other
collective ("weak" scaling: every GPU holds its own 10 M reads).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric/value/unit/... + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x5EED0002
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(k: int, read_len: int, n_per_1024: int, budget_s: float):
    """The reference benchmark's own per-record loop (normalize -> reverse_complement -> canonical_kmers, count
    items; reference benches/benchmark.rs:32-41) as restated by the oracle, on all host cores, on a bounded
    prefix of the same synthetic read set."""
    import numpy as np

    import oracle as O  # the checker / CPU port: only timed here, never part of the GPU path

    threads = os.cpu_count() or 1
    probe = 100_000
    buf = O.synth_reads(SEED, 0, probe, read_len, n_per_1024)
    offs = np.arange(probe + 1, dtype=np.uint64) * (read_len + 1)
    t0 = time.perf_counter()
    O.count_batch(buf, offs, 1, k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, threads)
    rate = probe / max(time.perf_counter() - t0, 1e-6)  # reads/s
    n = int(min(max(rate * budget_s, probe), 20_000_000))
    buf = O.synth_reads(SEED, 0, n, read_len, n_per_1024)
    offs = np.arange(n + 1, dtype=np.uint64) * (read_len + 1)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        nt_, nf_ = O.count_batch(buf, offs, 1, k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    t0 = time.perf_counter()
    O.count_batch(buf[: (n // 8) * (read_len + 1)], offs[: n // 8 + 1], 1, k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, 1)
    dt1 = time.perf_counter() - t0
    return {
        "value": round(n * read_len / best / 1e9, 4),
        "unit": "Gbases/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {n} reads of the same synthetic set ({n * read_len / 1e6:.0f} Mbases), needletail-equivalent "
                  f"CPU path (C restatement of normalize->reverse_complement->CanonicalKmers counting loop; Rust "
                  f"toolchain unavailable), {threads} threads, best of 2",
        "single_thread_value": round((n // 8) * read_len / dt1 / 1e9, 4),
        "n_total_sample": nt_,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed GPU activity before the warm-up steps: a step lasts 0.7 ms and the clocks take ~40 ms of load to "
                         "reach their steady state (0.75 ms per scan cold, 0.63-0.65 ms warm on the same box)")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--n-per-1024", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="threads per block (0 = the library's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--sync-allreduce", action="store_true",
                    help="keep the per-step all-reduce on the scan's critical path instead of overlapping it")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend; gloo + --single-device exercise the N > 1 code path on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true",
                    help="test affordance: every rank uses cuda:0 (NOT a measurement: ranks share one GPU)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, "needletail_amd", "libneedletail_amd.so")) and \
            int(os.environ.get("RANK", "0")) == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        import subprocess  # a snapshot without the built library: compile it (hipcc is in the image); no other path exists
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "needletail_amd", "csrc")], stdout=sys.stderr)
    import needletail_amd as nt
    from needletail_amd import _lib as ntl
    from needletail_amd import distributed as nd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run the RCCL path is exercised even at N = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")

    stride = args.read_len + 1
    n_bytes = args.reads * stride
    seq = torch.empty(n_bytes + 2048, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(ntl.ACC_WORDS, dtype=torch.int64, device="cuda")
    ctx = nt.Context(dev_index, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_launch(args.blocks, args.threads)
    ctx.accum_bind_device(acc)
    first_read, _ = nd.shard_range(args.reads * world, rank, world)
    ctx.synth_reads_device(SEED, first_read, args.reads, args.read_len, args.n_per_1024, seq)
    torch.cuda.synchronize()

    path, pre = nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE

    decode = nd.decode_accumulators

    # parity gate before timing (rank 0, small prefix): the bench refuses to time a wrong kernel
    if not args.no_verify and rank == 0:
        import oracle as O  # checker only
        sample = min(20_000, args.reads)
        ctx.accum_reset()
        ctx.reduce_device(seq, sample * stride, args.k, path, pre)
        torch.cuda.synchronize()
        got = decode(acc)
        want = O.reduce_fused(O.synth_reads(SEED, first_read, sample, args.read_len, args.n_per_1024),
                              args.k, True, True, True)
        for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
            if got[key] != want[key]:
                raise SystemExit(f"parity check failed on {key}: gpu {got[key]} != oracle {want[key]}")
        if not np.array_equal(got["hist"], want["hist"]):
            raise SystemExit("parity check failed on the histogram")

    # Two accumulator sets: the RCCL all-reduce of step i (on RCCL's own stream) overlaps the scan of step i+1, which
    # writes the other set.  Every step's accumulators are still all-reduced, and the last one is waited for inside
    # the timed region.  --sync-allreduce keeps the collective on the scan's critical path (A/B).
    accs = [acc, torch.zeros_like(acc)]
    pending = [None, None]
    state = {"i": 0}

    def step():
        j = state["i"] & 1
        state["i"] += 1
        if pending[j] is not None:
            pending[j].wait()  # the scan stream waits for the older all-reduce of this set
            pending[j] = None
        ctx.accum_bind_device(accs[j])
        ctx.accum_reset()
        ctx.reduce_device(seq, n_bytes, args.k, path, pre)
        if use_dist:
            # ONE RCCL sum all-reduce over xGMI: histogram + counters + digests
            if args.sync_allreduce:
                dist.all_reduce(accs[j], op=dist.ReduceOp.SUM)
            else:
                pending[j] = dist.all_reduce(accs[j], op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for j in (0, 1):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None

    # steady-state clocks before anything is timed (setup, like generating the reads; the W warm-up and K timed steps follow)
    # (a fixed number of steps, not a wall-clock loop: every rank must issue the same number of collectives)
    for i in range(int(args.preheat_ms / 0.6)):
        step()
        if i % 32 == 31:
            drain()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.scan_time_ms()  # drop warm-up events (none recorded yet)
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.enable_timing(False)
    kern_ms, launches = ctx.scan_time_ms()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([kern_ms / max(launches, 1)], dtype=torch.float64, device="cuda")
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kern_avg_ms = float(km.item())
    else:
        kern_avg_ms = kern_ms / max(launches, 1)

    res = decode(accs[(state["i"] - 1) & 1])
    if state["i"] >= 2 and not torch.equal(accs[0], accs[1]):
        raise SystemExit("the two accumulator sets disagree: a step's all-reduce was lost or doubled")
    total_reads = args.reads * world
    ok = res["n_total"] == res["n_fwd"] + res["n_rc"] == int(res["hist"].sum()) and \
        0 < res["n_total"] <= total_reads * (args.read_len - args.k + 1)
    if not ok:
        raise SystemExit(f"inconsistent reduced result: {res['n_total']} {res['n_fwd']} {res['n_rc']}")

    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the number
    # comes from the committed rocprofv3 --pmc passes over this very command (tools/profile_round.sh ->
    # profiles/<round>/pmc_scan_kernel.json; FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) when the workload is the default one.
    traffic, traffic_source = args.traffic_bytes, ("--traffic-bytes" if args.traffic_bytes is not None else None)
    if traffic is None and (args.reads, args.read_len, args.k, args.n_per_1024) == (10_000_000, 150, 21, 1):
        import glob
        cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", "pmc_scan_kernel.json")))
        if cands:
            try:
                with open(cands[-1]) as fh:
                    pmc = json.load(fh)
                traffic = pmc["hbm_read_bytes_per_launch_corrected"] + pmc.get("hbm_write_bytes_per_launch", 0.0)
                traffic_source = os.path.relpath(cands[-1], os.path.dirname(os.path.abspath(__file__))) + \
                    " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command)"
            except (OSError, KeyError, ValueError):
                traffic, traffic_source = None, None

    if rank == 0:
        bases = total_reads * args.read_len
        value = bases * args.steps / elapsed / 1e9
        achieved = n_bytes / (kern_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "Gbases/s canonical k=21 on 150 bp FASTQ; % HBM-read roofline at 1/2/4/8 GPU",
            "value": round(value, 3),
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat_ms": args.preheat_ms,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"Synthetic FASTQ {args.reads / 1e6:g}M x {args.read_len} bp per GPU, k={args.k} canonical "
                            f"k-mers (normalize -> reverse_complement -> canonical_kmers), reduce mode, "
                            f"device-resident batch",
                "reads_per_gpu": args.reads, "read_len": args.read_len, "k": args.k,
                "seed": hex(SEED), "n_rate": f"{args.n_per_1024}/1024",
                "outputs": "n_total,n_fwd,n_rc,4096-bin prefix histogram,sum64,xor64",
                "parallelism": f"records sharded over {world} GPU(s), one RCCL all-reduce per step" if world > 1
                               else "single GPU",
                "launch": {"blocks": args.blocks or "auto", "threads": args.threads or "auto"},
                **({"test_mode": f"{args.backend} backend, all ranks on cuda:0 - NOT a measurement"}
                   if (args.single_device or args.backend != "nccl") else {}),
            },
            "result": {"n_total": res["n_total"], "n_fwd": res["n_fwd"], "sum": hex(res["sum"]), "xor": hex(res["xor"])},
            "roofline": {
                "bound": "hbm",
                "kernel": (f"ntk::scan2_kernel<{args.k}, true, true, false, 14>" if args.k > 16
                           else f"ntk::scan_kernel<1, true, true, true, true, {args.k}, true, false>"),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": n_bytes,
                "kernel_ms": round(kern_avg_ms, 5),
                "launches_timed": launches,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.k, args.read_len, args.n_per_1024, args.cpu_budget_s)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
