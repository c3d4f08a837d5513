#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configs.

A "step" is one pass of the hot path (canonical k=21 extraction, reduce mode: counters + 4096-bin prefix histogram +
sum/xor digests of every canonical k-mer) over synthetic 150 bp reads that are already resident in HBM, plus, for
N > 1, the single RCCL all-reduce of the accumulators over xGMI (through the C ABI: ntk_allreduce_accumulators).

    N = 1   configs[1]: 10 M reads, SplitMix64 seed 0x5EED0002, N rate 1/1024              ("scaling": "weak")
    N > 1   configs[3]: 100 M reads in total, seed 0x5EED0004, record batches of 2^20 reads dealt round-robin to the
            GPUs (record i -> GPU (i / 2^20) mod N); every GPU keeps its shard resident     ("strong")

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a torch.distributed environment re-launches itself as
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
        (one process per GPU); launched that way by the driver it runs as is.

Rank 0 prints ONE JSON line: metric/value/unit/... + "roofline" + "cpu_baseline" + "secondary" (N = 1).
Before anything is timed every rank compares its WHOLE shard bit-exactly with the oracle (all five scalars and the
4096 bins); after timing, the all-reduced result is compared with the sum of the ranks' oracle results.
"""
import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED_C2, SEED_C3, SEED_C4 = 0x5EED0002, 0x5EED0003, 0x5EED0004
C4_TOTAL_READS = 100_000_000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SCALARS = ("n_total", "n_fwd", "n_rc", "sum", "xor")


def effective_cpus():
    """CPUs this process can actually use: the logical CPUs, cut down by the affinity mask and by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs under a 16-CPU quota: 256 busy threads there are 16 cores' worth of time slices)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return eff, {"logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota}


def stats_equal(a, b):
    import numpy as np
    return all(int(a[k]) == int(b[k]) for k in SCALARS) and np.array_equal(np.asarray(a["hist"], dtype=np.uint64),
                                                                           np.asarray(b["hist"], dtype=np.uint64))


def stats_sum(items):
    import numpy as np
    out = {"n_total": 0, "n_fwd": 0, "n_rc": 0, "sum": 0, "xor": 0, "hist": np.zeros(4096, dtype=np.uint64)}
    for s in items:
        for k in ("n_total", "n_fwd", "n_rc"):
            out[k] += int(s[k])
        out["sum"] = (out["sum"] + int(s["sum"])) & (2 ** 64 - 1)
        out["xor"] ^= int(s["xor"])
        out["hist"] = out["hist"] + np.asarray(s["hist"], dtype=np.uint64)
    return out


def oracle_shard(batches, seed, read_len, n_per_1024, k, path, pre, threads, literal):
    """The oracle's reduced result of a shard given as [(first_read, n_reads), ...].  literal: the reference's per-record
    chain with its allocations (normalize -> reverse_complement -> CanonicalKmers) on `threads` threads - 0.15 Gbases/s on
    256 threads, used for the 10 M-read single-GPU shard; otherwise the oracle's second formulation (rolling values, pinned
    against the literal chain by tests/test_oracle_golden.py) on `threads` threads, which keeps the 100 M-read runs short."""
    import numpy as np

    import oracle as O  # the checker
    parts = []
    for first, n in batches:
        buf = O.synth_reads(seed, first, n, read_len, n_per_1024)
        if literal:
            offs = np.arange(n + 1, dtype=np.uint64) * (read_len + 1)
            parts.append(O.reduce_batch(buf, offs, 1, k, path, pre, threads))
        else:
            parts.append(O.reduce_fused_parallel(buf, read_len + 1, k, True, path == O.PATH_BYTES_CANONICAL, pre >= O.PRE_NORMALIZE, threads))
    return stats_sum(parts)


def reference_toolchain_note():
    """if `cargo` AND a needletail crate with a vendored registry are on this box, the real crate is the baseline to
    prefer (rust/cpu_baseline/ holds the harness that reads the same synthetic reads and prints the same reduced result; build:
    `cargo build --release --offline` there with NEEDLETAIL_SRC pointing at the crate).  This image has neither, so the C port
    is timed; the note says which of the two was missing."""
    import shutil
    cargo = shutil.which("cargo")
    src = os.environ.get("NEEDLETAIL_SRC", "")
    vendored = bool(src) and os.path.isfile(os.path.join(src, "Cargo.toml")) and os.path.isdir(os.path.join(src, "vendor"))
    if cargo and vendored:
        return (f"cargo at {cargo} and a vendored crate at {src}: build rust/cpu_baseline (cargo build --release --offline) and pass "
                "--cpu-reference-bin to time the real crate (kind: reference)")
    missing = [w for w, ok in (("cargo on PATH", cargo), ("NEEDLETAIL_SRC = a needletail crate with vendor/", vendored)) if not ok]
    return "absent (" + "; ".join("no " + m for m in missing) + "): the C port is the baseline"


def cpu_reference_run(binary, host, n_reads, read_len, k, threads):
    """Times the reference crate's own chain through rust/cpu_baseline's harness (a binary built elsewhere with cargo): the reads
    go in as a file of fixed-stride records, the harness prints one JSON line with the reduced result and its seconds."""
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".reads") as f:
        f.write(memoryview(host)); f.flush()
        r = subprocess.run([binary, f.name, str(n_reads), str(read_len), str(k), str(threads)], capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise SystemExit(f"cpu_baseline: {binary} failed: {r.stderr[-400:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def cpu_baseline(gpu_ctx, seq, k, read_len, n_per_1024, n_reads, budget_s, reference_bin=None):
    """the needletail-equivalent CPU path (C restatement of the reference's per-record chain, allocations
    included; the Rust toolchain is unavailable) built -O3 -march=native on this host, four variants, each asserted equal to
    the GPU result on the same reads: bytes / bits x 1 thread / all host threads.  The N-thread variants run the GPU's
    whole read set; the 1-thread variants a prefix sized to the time budget."""
    import numpy as np

    import needletail_amd as nt
    import oracle as O  # the checker / CPU port: only timed here, never part of the GPU path
    flags = O.use_native_build()
    threads, host_info = effective_cpus()
    stride = read_len + 1
    host = O.synth_reads(SEED_C2, 0, n_reads, read_len, n_per_1024)
    offs = np.arange(n_reads + 1, dtype=np.uint64) * stride

    def gpu_result(n, path, pre):
        gpu_ctx.accum_reset()
        gpu_ctx.reduce_device(seq, n * stride, k, path, pre)
        return gpu_ctx.accum_read()

    variants = {}
    # (the reference's bit path feeds strip_returns output; the synthetic reads carry no line breaks inside a record)
    for name, path, pre in (("bytes", O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), ("bits", O.PATH_BITS_CANONICAL, O.PRE_STRIP_RETURNS)):
        gpath = nt.PATH_BYTES_CANONICAL if name == "bytes" else nt.PATH_BITS_CANONICAL
        gpre = nt.PRE_NORMALIZE if name == "bytes" else nt.PRE_STRIP_RETURNS
        # N threads, whole set, median of up to 5 repetitions inside the budget
        times, got = [], None
        t_begin = time.perf_counter()
        for rep in range(5):
            t0 = time.perf_counter()
            got = O.reduce_batch(host, offs, 1, k, path, pre, threads)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s / 4 and rep >= 2:
                break
        if not stats_equal(got, gpu_result(n_reads, gpath, gpre)):
            raise SystemExit(f"cpu_baseline: cpu-{name}-Nt result differs from the GPU result")
        variants[f"cpu-{name}-{threads}t"] = {"Gbases_s": round(n_reads * read_len / sorted(times)[len(times) // 2] / 1e9, 4),
                                             "reads": n_reads, "threads": threads, "repetitions": len(times),
                                             "equal_to_gpu": True}
        # 1 thread, a prefix
        rate_nt = n_reads / sorted(times)[len(times) // 2]
        n1 = int(min(n_reads, max(50_000, rate_nt / threads * 4 * (budget_s / 8))))
        t0 = time.perf_counter()
        got1 = O.reduce_batch(host[: n1 * stride], offs[: n1 + 1], 1, k, path, pre, 1)
        dt1 = time.perf_counter() - t0
        if not stats_equal(got1, gpu_result(n1, gpath, gpre)):
            raise SystemExit(f"cpu_baseline: cpu-{name}-1t result differs from the GPU result")
        variants[f"cpu-{name}-1t"] = {"Gbases_s": round(n1 * read_len / dt1 / 1e9, 4), "reads": n1, "threads": 1,
                                      "repetitions": 1, "equal_to_gpu": True}
        # informational: every thread keeps its normalize / reverse-complement buffers across records - NOT the reference's
        # behaviour (it allocates three Vecs per record); shows what the allocator costs in the number above
        t0 = time.perf_counter()
        got_a = O.reduce_batch(host, offs, 1, k, path, pre, threads, reuse_buffers=True)
        dta = time.perf_counter() - t0
        if not stats_equal(got_a, got):
            raise SystemExit(f"cpu_baseline: cpu-{name}-Nt-arena result differs")
        variants[f"cpu-{name}-{threads}t-arena"] = {"Gbases_s": round(n_reads * read_len / dta / 1e9, 4), "reads": n_reads, "threads": threads,
                                                   "repetitions": 1, "equal_to_gpu": True,
                                                   "note": "per-thread buffers reused across records: not the reference's behaviour"}
    head = variants[f"cpu-bytes-{threads}t"]
    if reference_bin:   # the real crate: its result must equal the GPU's too
        ref = cpu_reference_run(reference_bin, host, n_reads, read_len, k, threads)
        gpu = gpu_result(n_reads, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
        if not all(int(ref[x]) == int(gpu[x]) for x in ("n_total", "n_fwd", "sum", "xor")):
            raise SystemExit("cpu_baseline: the reference crate's result differs from the GPU result")
        variants[f"needletail-crate-{threads}t"] = {"Gbases_s": round(n_reads * read_len / float(ref["seconds"]) / 1e9, 4), "reads": n_reads,
                                                  "threads": threads, "equal_to_gpu": True}
        head = variants[f"needletail-crate-{threads}t"]
        return {"value": head["Gbases_s"], "unit": "Gbases/s", "cores": threads, "kind": "reference",
                "sample": f"the GPU's own {n_reads} reads through the needletail crate (rust/cpu_baseline harness)", "variants": variants}
    return {
        "value": head["Gbases_s"],
        "unit": "Gbases/s",
        "cores": threads,
        "kind": "port",
        "reference_toolchain": reference_toolchain_note(),
        "sample": f"the GPU's own {n_reads} reads ({n_reads * read_len / 1e6:.0f} Mbases) for the {threads}-thread variants, a "
                  f"prefix for the 1-thread ones; needletail-equivalent CPU path (C restatement of normalize -> "
                  f"reverse_complement -> CanonicalKmers / strip_returns -> BitNuclKmer per record, allocations included; Rust "
                  f"toolchain unavailable), reduced outputs asserted equal to the GPU's",
        "build": f"gcc {flags}",
        "host": host_info,
        "bound": "the scalar per-record chain itself (three heap allocations per record as the reference makes them, src/sequence.rs:20,202-208, "
                 "then a byte-at-a-time window walk); the -arena variants (buffers reused per thread) show the allocator's share",
        "variants": variants,
    }


def secondary_measurements(ctx, nt, torch, k21_seq, k21_bytes, reads, read_len):
    """Driver-visible numbers for the paths next to the headline: config-3 bit path, materialise mode,
    quality-masked scan, and the H2D-inclusive pipeline.  Each one is checked against the oracle on a prefix first."""
    import numpy as np

    import oracle as O  # checker
    out = {}

    def kernel_ms(fn, reps):
        # warm-up by time, not by count: each of these measurements follows an oracle check on the CPU (an idle GPU clocks down)
        # and a sub-millisecond kernel needs some tens of launches to be back at its steady-state clock (the headline has --preheat-ms)
        t_warm = time.perf_counter()
        n_warm = 0
        while n_warm < 3 or time.perf_counter() - t_warm < 0.05:
            fn()
            n_warm += 1
            if n_warm % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ctx.scan_time_ms()
        ctx.enable_timing(True)
        for _ in range(reps):
            fn()
        ms, nl = ctx.scan_time_ms()
        ctx.enable_timing(False)
        return ms / max(nl, 1) * (nl / reps)   # ms per pass (a pass may be several launches)

    # configs[2]: 1 M x 10 kb contigs, k = 31, bit-packed canonical path (BitNuclKmer)
    n_contigs, clen = 1_000_000, 10_000
    cbytes = n_contigs * (clen + 1)
    cseq = torch.empty(cbytes + 2048, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(SEED_C3, 0, n_contigs, clen, 1, cseq)
    torch.cuda.synchronize()
    pre_n = 200
    ctx.accum_reset()
    ctx.reduce_device(cseq, pre_n * (clen + 1), 31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS)
    want = O.reduce_fused(O.synth_reads(SEED_C3, 0, pre_n, clen, 1), 31, True, False, False)
    if not stats_equal(ctx.accum_read(), want):
        raise SystemExit("secondary: config-3 prefix differs from the oracle")
    ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(cseq, cbytes, 31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS)), 10)
    out["config3_bits_k31"] = {"workload": "1 M x 10 kb contigs, k=31, BitNuclKmer canonical, reduce mode, resident",
                               "kernel_ms": round(ms, 4), "Gbases_s": round(n_contigs * clen / (ms * 1e-3) / 1e9, 1),
                               "GB_s": round(cbytes / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(cbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del cseq
    torch.cuda.empty_cache()

    # forward-only bit path (BitNuclKmer with canonical = false, reference src/bitkmer.rs:80-108) on the config-2 batch: what the
    # kernel reaches when the contract asks for no strand choice (no rc stream, no compare, no strand counter)
    pre_r = 2000
    ctx.accum_reset()
    ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 21, nt.PATH_BITS, nt.PRE_NONE)
    hs = k21_seq[: pre_r * (read_len + 1)].cpu().numpy().tobytes()
    if not stats_equal(ctx.accum_read(), O.reduce_fused(hs, 21, False, False, False)):
        raise SystemExit("secondary: forward-only prefix differs from the oracle")
    ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BITS, nt.PRE_NONE)), 20)
    out["bits_forward_k21"] = {"workload": "config-2 batch, k=21, BitNuclKmer canonical=false, reduce mode, resident", "kernel_ms": round(ms, 4),
                               "GB_s": round(k21_bytes / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(k21_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # round 6: the byte path on input nobody normalised (Sequence::canonical_kmers on a clean record: normalize returns None, reference
    # src/sequence.rs:57-61): the speculative packed-value scan + the raw-byte kernel that returns at once when no lower-case byte was seen.
    # The C2 batch is upper case with N: the result must equal the oracle's raw-byte chain (prefix) and the normalised run's (whole batch:
    # no U, no lower case in it); hipEvents span both kernels of the pair.
    ctx.accum_reset()
    ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
    if not stats_equal(ctx.accum_read(), O.reduce_records(hs.split(b"\n")[:pre_r], 21, O.PATH_BYTES_CANONICAL, O.PRE_NONE)):
        raise SystemExit("secondary: un-normalised byte-path prefix differs from the oracle's raw-byte chain")
    ctx.accum_reset(); ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    want_norm = ctx.accum_read()
    ctx.accum_reset(); ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
    if not stats_equal(ctx.accum_read(), want_norm):
        raise SystemExit("secondary: the speculative scan of the un-normalised batch differs from the normalised run")
    ms_n = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)), 20)
    ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)), 20)
    out["bytes_unnormalised_k21"] = {"workload": "config-2 batch as it is (upper case), NTK_PATH_BYTES_CANONICAL with pre = NONE: speculative packed-value scan, "
                                                 "raw-byte kernel queued behind it (returns at once: no lower-case byte)",
                                     "kernel_ms": round(ms, 4), "GB_s": round(k21_bytes / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(k21_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "normalised_run_same_loop_ms": round(ms_n, 4), "round5_raw_byte_kernel_ms": 5.1}
    # k > 32 on the reduce face (CanonicalKmers takes k: u8): counters + histogram, no sum / xor; a prefix against the literal iterator
    pre_w = 300
    recs_w = hs.split(b"\n")[:pre_w]
    ctx.accum_reset(); ctx.reduce_device(k21_seq, pre_w * (read_len + 1), 64, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    got_w = ctx.accum_read()
    code_w = {65: 0, 67: 1, 71: 2, 84: 3}
    nt_w = nrc_w = 0
    hist_w = np.zeros(4096, dtype=np.uint64)
    for r_ in recs_w:
        r_ = O.normalize(r_)[0]
        rc_ = O.reverse_complement(r_)
        pos_, flg_ = O.canonical_kmers_arrays(r_, rc_, 64)
        for p_, f_ in zip(pos_.tolist(), flg_.tolist()):
            sl = rc_[len(rc_) - p_ - 64: len(rc_) - p_] if f_ else r_[p_: p_ + 64]
            b_ = 0
            for ch in sl[:6]:
                b_ = b_ * 4 + code_w[ch]
            hist_w[b_] += 1
        nt_w += len(pos_); nrc_w += int(flg_.sum())
    if not (got_w["n_total"] == nt_w == got_w["n_undigested"] and got_w["n_rc"] == nrc_w and np.array_equal(got_w["hist"], hist_w) and got_w["sum"] == 0):
        raise SystemExit("secondary: k = 64 reduce differs from the oracle's literal iterator on the prefix")
    ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 64, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)), 6)
    out["bytes_k64_counts"] = {"workload": "config-2 batch, CanonicalKmers k = 64 (33 <= k <= 255: counters + 6-base histogram, no sum / xor), resident",
                               "kernel": "wide_canonical_reduce_kernel<true> (strand on the first 32 bases of the packed streams; canonical_bytes_reduce_kernel<true> "
                                         "queued behind its flag, returns at once on this batch)", "kernel_ms": round(ms, 4), "GB_s": round(k21_bytes / (ms * 1e-3) / 1e9, 1)}

    # materialise mode (dense u64 per window + two flag planes), config-2 batch
    vals = torch.empty((k21_bytes + 15) // 16 * 16, dtype=torch.int64, device="cuda")
    v16 = torch.empty((k21_bytes + 15) // 16, dtype=torch.int16, device="cuda")
    r16 = torch.empty_like(v16)
    ms = kernel_ms(lambda: ctx.materialize_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, vals, v16, r16), 10)
    moved = k21_bytes + 8 * k21_bytes + k21_bytes / 4
    out["materialize_k21"] = {"kernel_ms": round(ms, 4), "GB_s_read_plus_write": round(moved / (ms * 1e-3) / 1e9, 1),
                              "frac_of_8TBs": round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del vals, v16, r16
    torch.cuda.empty_cache()

    # quality-masked scan (two byte streams: 2 B per base), cutoff '#' + 2
    qual = torch.full((k21_bytes + 2048,), 73, dtype=torch.uint8, device="cuda")
    qual[::7] = 34
    torch.cuda.synchronize()
    pre_r = 2000
    ctx.accum_reset()
    ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, d_qual=qual, quality_cutoff=35)
    hs = k21_seq[: pre_r * (read_len + 1)].cpu().numpy().tobytes()
    hq = qual[: pre_r * (read_len + 1)].cpu().numpy().tobytes()
    if not stats_equal(ctx.accum_read(), O.reduce_fused(O.quality_mask(hs, hq, 35), 21, True, True, True)):
        raise SystemExit("secondary: quality-masked prefix differs from the oracle")
    ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE,
                                                                 d_qual=qual, quality_cutoff=35)), 10)
    out["quality_masked_k21"] = {"kernel_ms": round(ms, 4), "GB_s_two_streams": round(2 * k21_bytes / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8TBs": round(2 * k21_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del qual
    torch.cuda.empty_cache()

    # fused minimizers (configs[4] kernel side): w = 11, k = 21, resident; a prefix against the oracle first
    try:
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
        if not stats_equal(ctx.accum_read(), O.minimizers_reduce(hs, 21, 11, True, True)):
            raise SystemExit("secondary: fused minimizers differ from the oracle on the prefix")
        ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)), 10)
        out["minimizers_w11_k21_resident"] = {"kernel_ms": round(ms, 4), "Gbases_s": round(reads * read_len / (ms * 1e-3) / 1e9, 1)}
        # a (k, w) outside the register-fused grid: the generic fused kernel (run-time k <= 31, w <= 49; one pass as well)
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 31, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=19)
        if not stats_equal(ctx.accum_read(), O.minimizers_reduce(hs, 31, 19, True, True)):
            raise SystemExit("secondary: generic fused minimizers differ from the oracle on the prefix")
        ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 31, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=19)), 6)
        out["minimizers_w19_k31_generic_resident"] = {"kernel": "minimizer_scan_kernel (any k <= 31, w <= 49; 26 <= k: keys value << 1 | strand)",
                                                       "kernel_ms": round(ms, 4), "Gbases_s": round(reads * read_len / (ms * 1e-3) / 1e9, 1)}
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 23, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
        if not stats_equal(ctx.accum_read(), O.minimizers_reduce(hs, 23, 11, True, True)):
            raise SystemExit("secondary: fused minimizers (k = 23) differ from the oracle on the prefix")
        ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 23, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)), 6)
        out["minimizers_w11_k23_resident"] = {"kernel": "scan2_kernel<23, ..., W = 11> (register-fused since round 5: windows of 33 bytes, three halo lanes; rounds 4 / 5a: "
                                                        "the generic kernel, key minimizers_w11_k23_generic_resident)",
                                              "kernel_ms": round(ms, 4), "Gbases_s": round(reads * read_len / (ms * 1e-3) / 1e9, 1)}
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, pre_r * (read_len + 1), 25, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
        if not stats_equal(ctx.accum_read(), O.minimizers_reduce(hs, 25, 11, True, True)):
            raise SystemExit("secondary: generic fused minimizers (k = 25) differ from the oracle on the prefix")
        ms = kernel_ms(lambda: (ctx.accum_reset(), ctx.reduce_device(k21_seq, k21_bytes, 25, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)), 6)
        out["minimizers_w11_k25_generic_resident"] = {"kernel": "minimizer_scan_kernel (k <= 25: one v_min_f64 per minimum on (value, position, strand) keys)",
                                                       "kernel_ms": round(ms, 4), "Gbases_s": round(reads * read_len / (ms * 1e-3) / 1e9, 1)}
    except nt.NtkError as e:  # pragma: no cover
        out["minimizers_w11_k21_resident"] = {"error": str(e)}

    # H2D-inclusive pipeline: FASTQ text in host memory -> parallel record parser -> pinned batches -> overlapped copies + scans
    p_reads = reads   # the whole read set (a 2 M-read sample under-reports by ~27 %: the ramp-up of the first batches dominates it)
    seqs = k21_seq[: p_reads * (read_len + 1)].cpu().numpy().reshape(p_reads, read_len + 1)
    idw = 9
    rec = np.empty((p_reads, 1 + idw + 1 + read_len + 1 + 2 + read_len + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    ids = np.arange(p_reads, dtype=np.int64)
    for d_ in range(idw):   # zero-padded decimal record numbers, digit by digit
        rec[:, 1 + d_] = (ids // 10 ** (idw - 1 - d_)) % 10 + 48
    del ids
    rec[:, 1 + idw] = 10
    rec[:, 2 + idw:2 + idw + read_len] = seqs[:, :read_len]
    rec[:, 2 + idw + read_len] = 10
    rec[:, 3 + idw + read_len] = ord("+")
    rec[:, 4 + idw + read_len] = 10
    rec[:, 5 + idw + read_len:5 + idw + 2 * read_len] = ord("I")
    rec[:, 5 + idw + 2 * read_len] = 10
    text = rec.tobytes()
    # the batched compat face (what an `impl Sequence` binds: Sequence::canonical_kmers for every record of a reader batch in
    # ONE call): host records in, (counts, pos, is_rc) out - packing, upload, scan, device-side compaction and download
    # included; the call pipelines 16 MiB chunks (two in flight).  Twice: plain (pageable) numpy arrays, and page-locked arrays
    # from ntk_pinned_alloc (what a host that owns its buffers would pass).
    try:
        import ctypes as C

        from needletail_amd import _lib as L
        c_reads = min(p_reads, 1_000_000)
        cap = c_reads * (read_len - 21 + 1)
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, c_reads * (read_len + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
        ref = ctx.accum_read()

        def pinned(n_bytes, dtype):
            p = C.c_void_p()
            L.check(L.lib().ntk_pinned_alloc(max(n_bytes, 8), C.byref(p)), "ntk_pinned_alloc")
            return p, np.frombuffer((C.c_uint8 * n_bytes).from_address(p.value), dtype=dtype)

        def run(flat, offs, counts, pos, flg):
            tot = C.c_uint64(0)
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                L.check(L.lib().ntk_canonical_kmers_batch(ctx._h, C.cast(flat.ctypes.data, C.c_char_p), offs.ctypes.data, c_reads, 21, counts.ctypes.data,
                                                          pos.ctypes.data, flg.ctypes.data, cap, C.byref(tot)), "ntk_canonical_kmers_batch")
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            if not (tot.value == int(counts.sum()) == ref["n_total"] and int(flg[: tot.value].sum()) == ref["n_rc"]):
                raise SystemExit("secondary: the batched compat face differs from the resident scan")
            return best, int(tot.value)

        src = np.ascontiguousarray(seqs[:c_reads, :read_len]).reshape(-1)
        offs = (np.arange(c_reads + 1, dtype=np.uint64) * np.uint64(read_len))
        best_pg, items = run(src, offs, np.zeros(c_reads, dtype=np.uint64), np.empty(cap, dtype=np.uint64), np.empty(cap, dtype=np.uint8))
        handles = []
        arrs = []
        for nb, dt_ in ((src.nbytes, np.uint8), (offs.nbytes, np.uint64), (c_reads * 8, np.uint64), (cap * 8, np.uint64), (cap, np.uint8)):
            h, a = pinned(nb, dt_)
            handles.append(h); arrs.append(a)
        arrs[0][:] = src; arrs[1][:] = offs
        best_pin, _ = run(*arrs)
        # the same items as two bit planes (ntk_canonical_kmers_batch_planes): the records are uploaded as they lie, 1/4 byte per
        # sequence byte comes back, the host iterator walks the bits (reference src/kmer.rs:114-129 needs nothing else)
        cap_w = c_reads * read_len // 16 + c_reads + 1
        hp = [pinned(nb, dt_) for nb, dt_ in (((c_reads + 1) * 8, np.uint64), (cap_w * 2, np.uint16), (cap_w * 2, np.uint16))]
        rec_bit, v16, r16 = (a for _, a in hp)
        nw, tot = C.c_uint64(0), C.c_uint64(0)
        best_pl = None
        for _ in range(4):
            t0 = time.perf_counter()
            L.check(L.lib().ntk_canonical_kmers_batch_planes(ctx._h, C.cast(arrs[0].ctypes.data, C.c_char_p), arrs[1].ctypes.data, c_reads, 21,
                                                             rec_bit.ctypes.data, v16.ctypes.data, r16.ctypes.data, cap_w, C.byref(nw), C.byref(tot)),
                    "ntk_canonical_kmers_batch_planes")
            dt = time.perf_counter() - t0
            best_pl = dt if best_pl is None else min(best_pl, dt)
        popc = lambda a: int(np.unpackbits(a[: nw.value].view(np.uint8)).sum())
        if not (tot.value == popc(v16) == ref["n_total"] and popc(r16) == ref["n_rc"]):
            raise SystemExit("secondary: the bit-plane compat face differs from the resident scan")
        # element-wise against the item form on a prefix of the records (both against the oracle in tests/test_gpu_parity.py)
        o = 0
        for i in range(2000):
            n_i = int(arrs[2][i])
            b0 = int(rec_bit[i])
            bits = np.unpackbits(v16[b0 >> 4: (b0 + read_len + 15) >> 4].astype(">u2").view(np.uint8))[b0 & 15: (b0 & 15) + read_len - 20]
            rbits = np.unpackbits(r16[b0 >> 4: (b0 + read_len + 15) >> 4].astype(">u2").view(np.uint8))[b0 & 15: (b0 & 15) + read_len - 20]
            pos_i = np.flatnonzero(bits)
            if not (np.array_equal(pos_i, arrs[3][o:o + n_i].astype(np.int64)) and np.array_equal(rbits[pos_i], arrs[4][o:o + n_i])):
                raise SystemExit("secondary: the bit planes and the item arrays disagree on record %d" % i)
            o += n_i
        # Sequence::bit_kmers(21, true) in the same form (ntk_bit_kmers_batch_planes): planes + DENSE values (8.25 B per position back), and the
        # planes alone (the host packs an emitted window's value from its bases); counts against the resident bit-path scan, values of a
        # prefix of the records against the item arrays of ntk_bit_kmers_batch
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, c_reads * (read_len + 1), 21, nt.PATH_BITS_CANONICAL, nt.PRE_NONE)
        ref_b = ctx.accum_read()
        hv, vals = pinned(cap_w * 16 * 8, np.uint64)
        bit_lines = {}
        for with_values in (True, False):
            best_b = None
            for _ in range(4):
                t0 = time.perf_counter()
                L.check(L.lib().ntk_bit_kmers_batch_planes(ctx._h, C.cast(arrs[0].ctypes.data, C.c_char_p), arrs[1].ctypes.data, c_reads, 21, 1,
                                                           rec_bit.ctypes.data, v16.ctypes.data, r16.ctypes.data, vals.ctypes.data if with_values else None,
                                                           cap_w, C.byref(nw), C.byref(tot)), "ntk_bit_kmers_batch_planes")
                dt = time.perf_counter() - t0
                best_b = dt if best_b is None else min(best_b, dt)
            if not (tot.value == popc(v16) == ref_b["n_total"] and popc(r16) == ref_b["n_rc"]):
                raise SystemExit("secondary: the bit-path plane face differs from the resident scan")
            bit_lines["with_dense_values" if with_values else "planes_only"] = {
                "seconds": round(best_b, 4), "Gbases_s": round(c_reads * read_len / best_b / 1e9, 2),
                "bytes_out_per_position": 8.25 if with_values else 0.25}
        # the values: every emitted window of the first 2000 records against the oracle's iterator
        for i in range(2000):
            b0 = int(rec_bit[i])
            rec_i = arrs[0][int(arrs[1][i]): int(arrs[1][i + 1])].tobytes()
            for p_, (v_, _k), f_ in O.bit_kmers(rec_i, 21, True):
                if int(vals[b0 + p_]) != v_:
                    raise SystemExit("secondary: ntk_bit_kmers_batch_planes value differs from the oracle's BitNuclKmer on record %d" % i)
        L.lib().ntk_pinned_free(hv)
        del vals
        out["compat_bit_planes_k21"] = {"call": "ntk_bit_kmers_batch_planes", "records": c_reads, "items": int(ref_b["n_total"]), **bit_lines,
                                        "note": "Sequence::bit_kmers(21, true) for a batch: emitted / was_rc planes per window start (+ one u64 per position), "
                                                "page-locked arrays, PCIe-inclusive; the item-array form (ntk_bit_kmers_batch) returns 17 B per item"}
        # sequence::minimizer (reference src/sequence.rs:139-152) for every record of the same batch in one call (ntk_minimizer_batch);
        # a prefix against the oracle's per-record function
        mh = [pinned(nb, dt_) for nb, dt_ in ((c_reads * 21, np.uint8), (c_reads * 8, np.uint64), (c_reads, np.uint8))]
        m_out, m_pos, m_flg = (a for _, a in mh)
        bad = C.c_uint64(0)
        best_mb = None
        for _ in range(4):
            t0 = time.perf_counter()
            L.check(L.lib().ntk_minimizer_batch(ctx._h, C.cast(arrs[0].ctypes.data, C.c_char_p), arrs[1].ctypes.data, c_reads, 21, m_out.ctypes.data,
                                                m_pos.ctypes.data, m_flg.ctypes.data, C.byref(bad)), "ntk_minimizer_batch")
            dt = time.perf_counter() - t0
            best_mb = dt if best_mb is None else min(best_mb, dt)
        for i in range(500):
            rec_i = arrs[0][int(arrs[1][i]): int(arrs[1][i + 1])].tobytes()
            if m_out[i * 21:(i + 1) * 21].tobytes() != O.minimizer(rec_i, 21):
                raise SystemExit("secondary: ntk_minimizer_batch differs from the oracle's sequence::minimizer on record %d" % i)
        minimizer_batch_line = {"call": "ntk_minimizer_batch", "records": c_reads, "length": 21, "seconds": round(best_mb, 4),
                                "Mrecords_s": round(c_reads / best_mb / 1e6, 1), "Gbases_s": round(c_reads * read_len / best_mb / 1e9, 2),
                                "note": "sequence::minimizer per record, page-locked arrays, PCIe-inclusive (bound by the upload); 500 records checked against the oracle"}
        for h, _ in mh:
            L.lib().ntk_pinned_free(h)
        for h, _ in hp:
            L.lib().ntk_pinned_free(h)
        for h in handles:
            L.lib().ntk_pinned_free(h)
        del arrs, handles, src, offs, hp, rec_bit, v16, r16, mh, m_out, m_pos, m_flg
        # one key per call: the item arrays ARE the reference iterator's items; the planes need a host bit-walk to become items
        # (CanonicalKmersPlanes.iter / arrays: not timed here) - different results, different keys
        out["compat_batch_face_k21"] = {"call": "ntk_canonical_kmers_batch", "records": c_reads, "items": items,
                                        "seconds": round(best_pin, 4), "Gbases_s": round(c_reads * read_len / best_pin / 1e9, 2),
                                        "Mitems_s": round(items / best_pin / 1e6, 1), "bytes_out_per_item": 9,
                                        "GB_s_out": round(items * 9 / best_pin / 1e9, 1),
                                        "note": "(counts, pos, is_rc) item arrays, page-locked host arrays, PCIe-inclusive: bound by the 9 B per item coming back",
                                        "pageable_arrays": {"seconds": round(best_pg, 4), "Gbases_s": round(c_reads * read_len / best_pg / 1e9, 2)}}
        out["compat_batch_planes_k21"] = {"call": "ntk_canonical_kmers_batch_planes", "records": c_reads, "items": items,
                                          "seconds": round(best_pl, 4), "Gbases_s": round(c_reads * read_len / best_pl / 1e9, 2),
                                          "Mitems_s": round(items / best_pl / 1e6, 1), "bytes_out_per_base": 0.25,
                                          "GB_s_in": round(c_reads * read_len / best_pl / 1e9, 1),
                                          "note": "valid / is_rc bit planes per window start + rec_bit[]; records uploaded as they lie (no packing pass), "
                                                  "page-locked host arrays (ntk_pinned_alloc), PCIe-inclusive: bound by the upload; the host-side walk of "
                                                  "the bits into items is NOT part of this time"}
        out["minimizer_batch_m21"] = minimizer_batch_line
    except (nt.NtkError, AttributeError) as e:  # pragma: no cover
        out["compat_batch_face_k21"] = {"error": str(e)}
    ctx.accum_reset()
    ctx.reduce_device(k21_seq, p_reads * (read_len + 1), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    want = ctx.accum_read()
    del rec, seqs
    # 24 parser threads on the 16 granted CPUs, 8 MiB batches, two copy streams: the best cell of tools/pipeline_batch_sweep.py
    # (profiles/r05c/pipeline_sweep.txt; the parser threads are the bound, the copies hide behind them)
    th = min(24, os.cpu_count() or 1)
    best = None
    for _ in range(6):
        t0 = time.perf_counter()
        st = nt.scan_file_parallel(ctx, None, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=th, batch_bytes=8 << 20, data=text)
        dt = time.perf_counter() - t0
        if not (stats_equal(st, want) and st["n_records"] == p_reads):
            raise SystemExit("secondary: the pipeline result differs from the resident scan")
        best = dt if best is None else min(best, dt)
    out["pipeline_fastq_h2d_inclusive"] = {"reads": p_reads, "parser_threads": th, "batch_MiB": 8, "copy_streams": 2, "seconds": round(best, 4),
                                           "Gbases_s": round(p_reads * read_len / best / 1e9, 2),
                                           "fastq_GB_s": round(len(text) / best / 1e9, 2)}
    try:
        out["config5_gzip_minimizers"] = config5_gzip_minimizers(ctx, nt, text, k21_seq, p_reads, read_len)
    except (nt.NtkError, OSError, MemoryError) as e:  # pragma: no cover
        out["config5_gzip_minimizers"] = {"error": str(e)}
    return out


def gzip_one_member(text, threads, level=6):
    """`text` as ONE gzip member at zlib level `level`, compressed by `threads` threads the way pigz does it: pieces of the input are
    deflated concurrently, each primed with the 32 KiB before it (so matches cross the piece boundaries, as in a stream written in one
    go) and closed with a sync flush; one header, one CRC-32 / ISIZE trailer.  (Python's zlib releases the GIL.)"""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(text)
    n = len(mv)
    piece = max(1 << 23, -(-n // (threads * 4)))
    cuts = list(range(0, n, piece)) or [0]

    def comp(a):
        b = min(n, a + piece)
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, bytes(mv[a - 32768:a])) if a >= 32768 else \
            zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(mv[a:b]) + c.flush(zlib.Z_FINISH if b == n else zlib.Z_SYNC_FLUSH)

    with ThreadPoolExecutor(threads) as ex:
        crc_f = ex.submit(zlib.crc32, mv)
        parts = list(ex.map(comp, cuts))
        crc = crc_f.result()
    return b"".join([b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"] + parts + [struct.pack("<II", crc & 0xFFFFFFFF, n & 0xFFFFFFFF)])


def bgzf_members(text, threads, block=60000):
    """Block gzip as bgzip / htslib write it: independent members of <= 64 KiB, each with its compressed size in a 'BC' extra subfield,
    closed by the empty EOF member."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(text)
    n = len(mv)
    group = block * 256

    def comp(a):
        out = bytearray()
        for o in range(a, min(n, a + group), block):
            ch = mv[o:min(n, o + block)]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = c.compress(ch) + c.flush()
            out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
            out += body + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch))
        return bytes(out)

    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(comp, range(0, n, group)))
    eof = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\0\0\0\0\0\0\0\0"
    return b"".join(parts + [eof])


def config5_gzip_minimizers(ctx, nt, text, k21_seq, reads, read_len):
    """BASELINE.json configs[4] end to end: the first 10 M reads of C2 as FASTQ text, ONE gzip member at zlib level 6,
    as a file -> inflate on the host's cores, the parser threads taking the text while it is inflated -> pinned batches -> overlapped H2D +
    fused (21, 11) minimizers; the result must equal the accumulators of the resident minimizer run over the same reads.  Routes: (a) the
    ordinary .gz through ntk_scan_file_parallel (speculative parallel inflate of the one deflate stream, all granted CPUs, streamed: bounded
    memory), (b) the same file through the
    streaming reader (zlib on one thread: what the reference does, src/parser/mod.rs:95-108; a 1 M-read sample), (c) block gzip
    (bgzip-style members).  Plus ntk_gunzip alone per thread count."""
    import ctypes as C
    import tempfile

    from needletail_amd import _lib as L
    cpus, host = effective_cpus()
    rec_bytes = len(text) // reads
    k, w = 21, 11
    ctx.accum_reset()
    ctx.reduce_device(k21_seq, reads * (read_len + 1), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w)
    want = ctx.accum_read()
    line = {"workload": f"first {reads} reads of C2 as FASTQ text ({len(text) / 1e9:.2f} GB), one gzip member, zlib level 6 -> file -> "
                        f"inflate + parse on the host -> pinned batches -> H2D + fused minimizers (k = {k}, w = {w})",
            "cpus_granted": cpus, "host": host}
    t0 = time.perf_counter()
    gz = gzip_one_member(text, cpus)
    line["gzip_bytes"] = len(gz)
    line["compress_s_untimed"] = round(time.perf_counter() - t0, 2)

    def gunzip(buf, threads):
        o, n, info = C.c_void_p(), C.c_uint64(0), L.GunzipInfo()
        t0 = time.perf_counter()
        L.check(L.lib().ntk_gunzip(buf, len(buf), threads, C.byref(o), C.byref(n), C.byref(info)), "ntk_gunzip")
        dt = time.perf_counter() - t0
        L.lib().ntk_gunzip_free(o, n.value)
        return dt, int(n.value), info

    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        path = os.path.join(d, "c5.fastq.gz")
        with open(path, "wb") as f:
            f.write(gz)
        best, ginfo, rss_peak, all_s = None, None, None, []

        def status_mb(key):
            return int(next(l for l in open("/proc/self/status") if l.startswith(key)).split()[1]) / 1024
        for _ in range(3):
            try:
                open("/proc/self/clear_refs", "w").write("5")   # reset VmHWM: the resident-set peak of THIS call
            except OSError:
                pass
            rss0 = status_mb("VmRSS")
            t0 = time.perf_counter()
            st = nt.scan_file_parallel(ctx, path, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=8 << 20, w=w, streaming_fallback=False)
            dt = time.perf_counter() - t0
            if not (stats_equal(st, want) and st["n_records"] == reads):
                raise SystemExit("secondary: config 5 (gzip + minimizers) differs from the resident minimizer run")
            all_s.append(round(dt, 3))
            if best is None or dt < best:
                best, ginfo, rss_peak = dt, st["gzip"], status_mb("VmHWM") - rss0
        line["plain_gz_parallel_inflate"] = {"call": "ntk_scan_file_parallel (the text is consumed WHILE it is inflated)", "inflate_threads": cpus,
                                             "parser_threads": ginfo["parse_threads"], "seconds": round(best, 3), "seconds_of_three_calls_back_to_back": all_s,
                                             "Gbases_s": round(reads * read_len / best / 1e9, 3), "text_GB_s": round(len(text) / best / 1e9, 2),
                                             "equal_to_resident_run": True, "route": ginfo["route"], "streamed": ginfo["streamed"],
                                             "first_batch_submitted_after_s": round(ginfo["first_batch_s"], 4),
                                             "peak_text_waiting_for_a_parser_MB": round(ginfo["peak_backlog_bytes"] / 1e6, 1),
                                             "peak_rss_above_call_start_MB": round(rss_peak, 1),
                                             "peak_rss_note": "includes the pages of the mmap-ed .gz file the call touched (its size: gzip_bytes)",
                                             "chunks": ginfo["chunks"], "chunks_dropped": ginfo["chunks_dropped"], "chunks_deferred": ginfo["chunks_deferred"],
                                             "decode_cpu_s": round(ginfo["decode_busy_s"], 3), "resolve_cpu_s": round(ginfo["resolve_busy_s"], 3)}
        # the inflate alone, per thread count (the whole file for 1 and all granted CPUs, a 2 M-read member for the ones between)
        table = []
        dt, n_out, info = gunzip(gz, cpus)
        assert n_out == len(text)
        table.append({"threads": cpus, "input": "whole file", "seconds": round(dt, 3), "text_GB_s": round(n_out / dt / 1e9, 2), "route": info.route,
                      "chunks": info.chunks, "chunks_dropped": info.chunks_dropped, "search_s": round(info.search_s, 3),
                      "decode_wall_s": round(info.decode_s, 3), "decode_cpu_s": round(info.decode_busy_s, 3),
                      "marker_share": round(info.marker_symbols / max(n_out, 1), 3),
                      "per_thread_text_MB_s": round(n_out / max(info.decode_busy_s, 1e-9) / 1e6, 1)})
        sample_reads = min(reads, 2_000_000)
        sample = gzip_one_member(memoryview(text)[: sample_reads * rec_bytes], cpus)
        tlist = sorted({1, 2, 4, 8, cpus} & set(range(1, cpus + 1)))
        for t in tlist:
            dt, n_out, info = min((gunzip(sample, t) for _ in range(2)), key=lambda x: x[0])
            table.append({"threads": t, "input": f"{sample_reads} reads", "seconds": round(dt, 3), "text_GB_s": round(n_out / dt / 1e9, 2), "route": info.route,
                          "decode_cpu_s": round(info.decode_busy_s, 3), "per_thread_text_MB_s": round(n_out / max(info.decode_busy_s, 1e-9) / 1e6, 1)})
        line["gunzip_alone"] = table
        # (b) the streaming reader on a 1 M-read member: one zlib thread, copies and kernels overlapped behind it
        s_reads = min(reads, 1_000_000)
        spath = os.path.join(d, "c5_sample.fastq.gz")
        with open(spath, "wb") as f:
            f.write(gzip_one_member(memoryview(text)[: s_reads * rec_bytes], cpus))
        ctx.accum_reset()
        ctx.reduce_device(k21_seq, s_reads * (read_len + 1), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w)
        want_s = ctx.accum_read()
        t0 = time.perf_counter()
        st = nt.scan_file(ctx, spath, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w)
        dt = time.perf_counter() - t0
        if not (stats_equal(st, want_s) and st["n_records"] == s_reads):
            raise SystemExit("secondary: config 5 through the streaming reader differs from the resident minimizer run")
        line["plain_gz_stream_one_thread"] = {"call": "ntk_scan_reader (zlib inflate on the reader thread: the reference's MultiGzDecoder arrangement)",
                                              "reads": s_reads, "seconds": round(dt, 3), "Gbases_s": round(s_reads * read_len / dt / 1e9, 3)}
        # (c) block gzip
        bpath = os.path.join(d, "c5.fastq.bgz")
        with open(bpath, "wb") as f:
            f.write(bgzf_members(text, cpus))
        bbest = None
        for _ in range(2):
            t0 = time.perf_counter()
            st = nt.scan_file_parallel(ctx, bpath, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=cpus, batch_bytes=8 << 20, w=w, streaming_fallback=False)
            dt = time.perf_counter() - t0
            if not (stats_equal(st, want) and st["n_records"] == reads):
                raise SystemExit("secondary: config 5 (block gzip + minimizers) differs from the resident minimizer run")
            bbest = dt if bbest is None else min(bbest, dt)
        line["block_gzip"] = {"call": "ntk_scan_file_parallel", "threads": cpus, "seconds": round(bbest, 3), "Gbases_s": round(reads * read_len / bbest / 1e9, 3)}
    line["Gbases_s"] = line["plain_gz_parallel_inflate"]["Gbases_s"]
    line["note"] = ("PCIe- and host-inclusive, never `value`; the GPU work (fused minimizers: secondary.minimizers_w11_k21_resident) hides behind the host's "
                    "inflate, which is the bound: gunzip_alone lists its rate per thread count")
    return line


class phase_deadline:
    """Fail fast instead of hanging: a phase that may block inside C (process-group rendezvous, ncclCommInitRank, the first
    collective) gets a deadline; when it passes, a watchdog thread prints the rank and the phase and ends the process with a
    non-zero status (os._exit: the main thread may be stuck in a call that never returns).  `with phase_deadline(...)`."""

    def __init__(self, phase: str, seconds: float, rank: int, fatal: bool = True):
        import threading
        self.phase, self.seconds, self.rank, self.fatal = phase, seconds, rank, fatal
        self.done = threading.Event()
        self.thread = threading.Thread(target=self._watch, daemon=True)

    def _watch(self):
        if not self.done.wait(self.seconds):
            sys.stderr.write(f"bench.py: rank {self.rank}: phase '{self.phase}' did not finish within {self.seconds:g} s - giving up "
                             f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                             f"WORLD_SIZE={os.environ.get('WORLD_SIZE')})\n")
            sys.stderr.flush()
            os._exit(3)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, etype, evalue, tb):
        self.done.set()
        if etype is not None and self.fatal and not issubclass(etype, (SystemExit, KeyboardInterrupt)):
            raise SystemExit(f"bench.py: rank {self.rank}: phase '{self.phase}' failed: {etype.__name__}: {evalue}")
        return False


def pmc_passes(args, n_bytes):
    """Counters of the scan kernel, collected live: one child run of this script per pass under `rocprofv3 --pmc` (no trace domains),
    5 steps each, per scan-kernel launch.  Pass 1 / 2: FETCH_SIZE / WRITE_SIZE (they do not fit one pass) -> HBM bytes; FETCH_SIZE counts
    64-byte units of 128-byte requests on gfx950: a wide coalesced streaming read shows up at exactly half its bytes
    (MI355X_MICROARCH.md, HBM) -> doubled; both counters are in KB.  Pass 3: SQ_INSTS_VALU / SALU / LDS, SQ_ACTIVE_INST_VALU,
    GRBM_GUI_ACTIVE -> instructions per tile and shader cycles per tile per SIMD.  The kernel's NAME is taken from the counter rows
    (the scan2_kernel instantiation with the most dispatches), not assumed.  Returns a dict; keys are missing for what could not be
    collected (no rocprofv3, a failed pass)."""
    import csv
    import glob
    import shutil
    import tempfile
    tool = shutil.which("rocprofv3")
    if not tool:
        return {}
    env = dict(os.environ, NTK_BENCH_PMC_CHILD="1", TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "1", "--preheat-ms", "5", "--no-verify", "--no-cpu-baseline",
             "--no-secondary", "--no-pmc", "--reads", str(args.reads or 10_000_000), "--read-len", str(args.read_len), "--k", str(args.k),
             "--n-per-1024", str(args.n_per_1024), "--blocks", str(args.blocks), "--threads", str(args.threads)]
    got, names = {}, collections.Counter()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "GRBM_GUI_ACTIVE"]):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            try:
                r = subprocess.run([tool, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child, cwd="/tmp", env=env,
                                   capture_output=True, text=True, timeout=240)
            except (OSError, subprocess.TimeoutExpired):
                continue
            vals = collections.defaultdict(list)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "scan2_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") in counters:
                        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                        names[row["Kernel_Name"]] += 1
            if r.returncode == 0:
                for c, v in vals.items():
                    got[c] = sum(v) / len(v)
    out = {}
    if names:
        out["kernel"] = names.most_common(1)[0][0]
    if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
        read_b, write_b = got["FETCH_SIZE"] * 1024 * 2, got["WRITE_SIZE"] * 1024
        out["traffic"] = read_b + write_b
        out["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child passes of this command, "
                                 f"per scan2_kernel launch; read {read_b:.0f} B = FETCH_SIZE KB x 1024 x 2 (gfx950 correction), write {write_b:.0f} B)")
    if all(c in got for c in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")):
        n_tiles = -(-(-(-n_bytes // 16)) // 62)                    # tiles of 62 emitting 16-byte slots (992 bases)
        cycles = got["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
        per_tile = cycles / (n_tiles / 1024.0)                     # 256 CUs x 4 SIMDs
        ipt = got["SQ_INSTS_VALU"] / n_tiles
        out["valu"] = {
            "insts_per_tile": round(ipt, 1),
            "salu_per_tile": round(got.get("SQ_INSTS_SALU", 0.0) / n_tiles, 1),
            "lds_per_tile": round(got.get("SQ_INSTS_LDS", 0.0) / n_tiles, 1),
            "shader_cycles_per_launch": round(cycles),
            "cycles_per_tile_per_simd": round(per_tile, 1),
            "cycles_per_inst": round(per_tile / ipt, 3),
            **({"dual_issued_per_tile": round(got["SQ_ACTIVE_INST_VALU2"] / n_tiles, 1),
                "issue_slots_per_tile": round(ipt - got["SQ_ACTIVE_INST_VALU2"] / n_tiles, 1),
                "slot_busy": round((ipt - got["SQ_ACTIVE_INST_VALU2"] / n_tiles) * 4.1 / per_tile, 3),
                "slot_note": "SQ_ACTIVE_INST_VALU2 = quad-cycles with a SECOND VALU instruction active: two full-rate instructions of two waves share "
                             "one 4.1-cycle issue slot (profiles/r04a); slots = instructions - pairs; slot_busy = slots x 4.1 cycles / measured cycles"}
               if "SQ_ACTIVE_INST_VALU2" in got else {}),
            "port_busy_by_SQ_ACTIVE_INST_VALU_x4": round(got.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (cycles * 1024), 4),
            "port_busy_note": "SQ_ACTIVE_INST_VALU counts ONE quad-cycle per instruction whatever its issue class (it equals SQ_INSTS_VALU): "
                              "this ratio is cycles_per_inst / 4 restated, not an occupancy (profiles/r04a/README.md)",
            "source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE "
                      "(one child pass, per scan2_kernel launch; tile = 992 bases per wave)",
        }
    return out


def roofline_bound(pmc, frac, k):
    """What limits the dominant kernel, from this run's counters: the yardstick of the metric stays the HBM-read roofline (peak / frac),
    `bound` says what the kernel is actually held by.  HBM-bound would mean running near the HBM rate; this kernel moves
    1.01 x its algorithmic bytes at ~0.4 of the HBM rate while its waves issue ~175 VALU instructions per 992-base tile: it is bound by
    VALU issue (half-rate instructions at 4.1 cycles, full-rate at 2.05: profiles/r04a/README.md), and says so."""
    out = {"bound": "hbm", "kernel": pmc.get("kernel") or f"(not observed: no counter pass ran) ntk::scan2_kernel<{k}, ...>"}
    valu = pmc.get("valu")
    if not valu:
        return out
    valu = dict(valu)
    cls_path = os.path.join(ROOT, "profiles", "r04b", f"valu_classes_k{k}.json")
    if os.path.exists(cls_path):
        try:
            with open(cls_path) as fh:
                cls = json.load(fh)
            issue = cls["half_rate"] * cls["cycles_half"] + cls["full_rate"] * cls["cycles_full"]
            valu["issue_classes"] = {"half_rate_per_tile": cls["half_rate"], "full_rate_per_tile": cls["full_rate"],
                                     "cycles_half": cls["cycles_half"], "cycles_full": cls["cycles_full"],
                                     "issue_cycles_per_tile": round(issue, 1),
                                     "issue_share_of_measured_cycles": round(issue / valu["cycles_per_tile_per_simd"], 3),
                                     "source": "static: " + os.path.relpath(cls_path, ROOT) + " (tools/isa_census.py --json of the shipped build; "
                                               "class costs measured by tools/ubench*.hip, profiles/r04a/)"}
        except (OSError, KeyError, ValueError):
            pass
    share = valu.get("issue_classes", {}).get("issue_share_of_measured_cycles")
    if frac < 0.7 and (share is None or share > 0.6):   # (at >= 0.7 of the HBM rate the memory system is the limit whatever the waves do)
        out["bound"] = "valu-issue"
        out["bound_note"] = ("the HBM-read roofline stays the yardstick (peak, frac); the kernel itself is held by VALU issue: "
                             f"{valu['insts_per_tile']} VALU instructions per 992-base tile in {valu['cycles_per_tile_per_simd']} shader cycles per "
                             "tile per SIMD" + (f", {share:.0%} of which is the instructions' own issue time" if share else ""))
    out["valu"] = valu
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed GPU activity before the warm-up steps: a step lasts ~0.5 ms and the clocks take ~40 ms of load to "
                         "reach their steady state")
    ap.add_argument("--reads", type=int, default=None, help="reads in total (default: 10 M at N = 1, 100 M at N > 1)")
    ap.add_argument("--workload", default="auto", choices=("auto", "c2", "c4"),
                    help="auto: configs[1] at N = 1, configs[3] at N > 1; c4 at N = 1 scans the whole config-4 read set on one GPU "
                         "(what the N > 1 result must equal)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--n-per-1024", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="threads per block (0 = the library's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=24.0)
    ap.add_argument("--cpu-reference-bin", default=os.environ.get("NTK_CPU_REFERENCE_BIN"),
                    help="rust/cpu_baseline's harness built against the real needletail crate (needs cargo + a vendored registry, "
                         "absent from this image): timed as cpu_baseline with kind 'reference'")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl: RCCL through the C ABI (ntk_comm_*); gloo + --single-device exercise the N > 1 host path on a 1-GPU "
                         "box with a torch.distributed all-reduce of the same accumulator words")
    ap.add_argument("--single-device", action="store_true",
                    help="test affordance: every rank uses cuda:0 (NOT a measurement: ranks share one GPU)")
    ap.add_argument("--init-timeout-s", type=float, default=120.0,
                    help="deadline for each start-up phase of an N > 1 run (process-group rendezvous, communicator init, first "
                         "collective): past it the rank exits non-zero naming itself and the phase instead of hanging")
    ap.add_argument("--allow-collective-fallback", action="store_true",
                    help="if the library's own RCCL communicator (ntk_comm_*) cannot come up, reduce with torch.distributed's "
                         "all_reduce instead (noted in config.collective); without this flag that is a fatal error")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic with rocprofv3 --pmc child passes (N = 1)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        if not args.single_device:
            from needletail_amd import _lib as ntl0
            have = ntl0.device_count()
            if have < args.gpus:
                raise SystemExit(f"--gpus {args.gpus} but {have} usable gfx950 device(s) are visible")
        # one process per GPU: re-launch under torch.distributed.run and pass its output through
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    # stdout carries the ONE JSON line and nothing else: librccl prints a version banner and gloo its connection notes
    # through C stdio on fd 1, so fd 1 is pointed at stderr for the life of the process and the line goes to the saved fd
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch one process per GPU (or plain `python bench.py --gpus N`)")
    if not os.path.exists(os.path.join(ROOT, "needletail_amd", "libneedletail_amd.so")) and rank == 0 and world == 1:
        # a snapshot without the built library: compile it (hipcc is in the image); no other path exists
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "needletail_amd", "csrc")], stdout=sys.stderr)
    import needletail_amd as nt
    from needletail_amd import _lib as ntl
    from needletail_amd import distributed as nd

    dev_index = 0 if args.single_device else local_rank
    use_dist = "RANK" in os.environ
    import datetime
    pg_timeout = datetime.timedelta(seconds=max(1.0, args.init_timeout_s))
    if use_dist and not (args.backend == "nccl" and not args.single_device):
        # gloo (test mode) needs no device: rendezvous first, so that a wrong MASTER_ADDR / MASTER_PORT fails fast everywhere
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        with phase_deadline("process-group rendezvous (gloo)", args.init_timeout_s + 5, rank):
            dist.init_process_group(backend="gloo", timeout=pg_timeout)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(dev_index)
    if use_dist and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        with phase_deadline("process-group rendezvous (nccl)", args.init_timeout_s + 5, rank):
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), timeout=pg_timeout)

    # ---- workload ------------------------------------------------------------------------------------------------
    stride = args.read_len + 1
    if args.workload == "c2" and world > 1:
        raise SystemExit("--workload c2 is the single-GPU configuration")
    if world == 1 and args.workload != "c4":
        seed, total_reads = SEED_C2, (args.reads or 10_000_000)
        batches = [(0, total_reads)]
        workload = (f"configs[1]: synthetic FASTQ {total_reads / 1e6:g}M x {args.read_len} bp, k={args.k} canonical k-mers "
                    f"(normalize -> reverse_complement -> canonical_kmers), reduce mode, device-resident batch")
    else:
        seed, total_reads = SEED_C4, (args.reads or C4_TOTAL_READS)
        batches = nd.round_robin_batches(total_reads, rank, world)
        workload = (f"configs[3]: synthetic FASTQ {total_reads / 1e6:g}M x {args.read_len} bp in total, k={args.k} canonical k-mers, "
                    f"record batches of 2^20 reads dealt round-robin to {world} GPU(s), one RCCL all-reduce of the accumulators per step")
    my_reads = sum(n for _, n in batches)
    n_bytes = my_reads * stride
    seq = torch.empty(n_bytes + 2048, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(ntl.ACC_WORDS, dtype=torch.int64, device="cuda")
    ctx = nt.Context(dev_index, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_launch(args.blocks, args.threads)
    ctx.accum_bind_device(acc)
    pos = 0
    for first, n in batches:   # the shard is the concatenation of its batches (every record ends with its break byte)
        ctx.synth_reads_device(seed, first, n, args.read_len, args.n_per_1024, seq[pos:])
        pos += n * stride
    torch.cuda.synchronize()
    path, pre = nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE

    # ---- the collective --------------------------------------------------------------------------------------------
    comm = None
    rccl = use_dist and args.backend == "nccl" and not args.single_device
    collective_note = None
    if rccl:
        # the product path: the library's own communicator (ntk_comm_*, RCCL through the C ABI).  Should it fail to come up on
        # ANY rank (say a librccl the process cannot load) that is FATAL, on every rank, with the reason - unless
        # --allow-collective-fallback was given: then every rank reduces the same accumulator words with torch.distributed's
        # all-reduce (still RCCL, still on the scan stream) and the JSON line says so.  Every phase has a deadline.
        err = "NTK_BENCH_FORCE_COLLECTIVE_FALLBACK is set (test hook)" if os.environ.get("NTK_BENCH_FORCE_COLLECTIVE_FALLBACK") else None
        try:
            ids = [nd.Communicator.unique_id() if rank == 0 and not err else None]
        except nt.NtkError as e:
            ids, err = [None], str(e)
        with phase_deadline("broadcast of the communicator id", args.init_timeout_s, rank):
            dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None:
            try:
                with phase_deadline("ntk_comm_init_rank (ncclCommInitRank)", args.init_timeout_s, rank, fatal=False):
                    comm = nd.Communicator.for_rank(ctx, world, rank, ids[0])
            except nt.NtkError as e:
                err = str(e)
        else:
            err = err or "rank 0 could not create a communicator id"
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
        with phase_deadline("agreement on the communicator (first torch collective)", args.init_timeout_s, rank):
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            all_ok = int(ok[0]) == 1
        if not all_ok:
            if comm is not None:
                comm.close()
            comm = None
            errs = [None] * world
            dist.all_gather_object(errs, err)
            why = next(e for e in errs if e)
            if not args.allow_collective_fallback:
                raise SystemExit(f"rank {rank}: the library's RCCL communicator did not come up ({why}); "
                                 "--allow-collective-fallback would reduce through torch.distributed instead")
            collective_note = "FALLBACK torch.distributed all_reduce (ntk_comm_* failed: %s)" % why
        else:
            with phase_deadline("first ntk_allreduce_accumulators (ncclAllReduce)", args.init_timeout_s, rank):
                ctx.accum_reset()
                comm.allreduce_accumulators()
                ctx.synchronize()
            if comm.size != world:
                raise SystemExit(f"rank {rank}: ntk_comm_size says {comm.size} ranks, WORLD_SIZE is {world}")

    def allreduce():
        if comm is not None:
            comm.allreduce_accumulators()          # ncclAllReduce(ncclUint64, ncclSum) on the scan stream + xor rebuild
        elif rccl:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)   # fallback (see above): int64 sums of the same words, xor rebuilt on read
        elif use_dist:
            t = acc.cpu()                          # test mode (gloo): the same words, summed on the host
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            acc.copy_(t)

    # ---- parity gate: the WHOLE shard against the oracle, before anything is timed -------------------------------------
    want_mine = None
    if not args.no_verify:
        ctx.accum_reset()
        ctx.reduce_device(seq, n_bytes, args.k, path, pre)
        got = ctx.accum_read()
        t0 = time.perf_counter()
        threads = max(1, min(os.cpu_count() or 1, 4 * effective_cpus()[0]) // world)
        import oracle as O  # checker only
        literal = my_reads <= 12_000_000 and world == 1
        want_mine = oracle_shard(batches, seed, args.read_len, args.n_per_1024, args.k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, threads,
                                 literal)
        verify_s = time.perf_counter() - t0
        if not stats_equal(got, want_mine):
            bad = [k for k in SCALARS if int(got[k]) != int(want_mine[k])]
            raise SystemExit(f"rank {rank}: parity check failed on the whole shard ({my_reads} reads): {bad or 'histogram'}")

    def step():   # a new result every step: the accumulators are zeroed by the scan's own launch (NTK_FLAG_RESET), then scan + fold
        ctx.reduce_device(seq, n_bytes, args.k, path, pre, reset=True)
        allreduce()

    # steady-state clocks before anything is timed (a fixed number of steps: every rank issues the same collectives)
    est_ms = max(0.5, my_reads / 10_000_000 * 0.5)
    for i in range(max(1, int(args.preheat_ms / est_ms))):
        step()
        if i % 32 == 31:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.scan_time_ms()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.enable_timing(False)
    kern_ms, launches = ctx.scan_time_ms()
    kern_avg_ms = kern_ms / max(launches, 1)
    ar_avg_ms = None
    if comm is not None:
        ar_ms, ar_calls = comm.allreduce_time_ms()   # the collective's own duration on this rank (events on the stream it runs on)
        ar_avg_ms = ar_ms / max(ar_calls, 1)
    per_rank = None
    if use_dist:
        # every rank's own numbers, so that a scaling line explains itself: scan kernel and all-reduce per step, wall per step
        mine = torch.tensor([kern_avg_ms, ar_avg_ms if ar_avg_ms is not None else -1.0, elapsed / args.steps * 1e3, float(my_reads)],
                            dtype=torch.float64, device="cuda" if rccl else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": i, "reads": int(v[3]), "kernel_ms": round(float(v[0]), 5),
                     "allreduce_ms": (round(float(v[1]), 5) if float(v[1]) >= 0 else None), "ms_per_step": round(float(v[2]), 5)}
                    for i, v in enumerate(t_.cpu() for t_ in allr)]
        t = torch.tensor([elapsed, kern_avg_ms], dtype=torch.float64, device="cuda" if rccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_avg_ms = float(t[0]), float(t[1])

    res = ctx.accum_read() if comm is not None or not use_dist else nd.decode_accumulators(acc)
    if not (res["n_total"] == res["n_fwd"] + res["n_rc"] == int(np.asarray(res["hist"]).sum())
            and 0 < res["n_total"] <= total_reads * (args.read_len - args.k + 1)):
        raise SystemExit(f"inconsistent reduced result: {res['n_total']} {res['n_fwd']} {res['n_rc']}")
    verified = None
    if want_mine is not None:
        if use_dist:
            parts = [None] * world
            dist.all_gather_object(parts, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in want_mine.items()})
            want_all = stats_sum(parts)
        else:
            want_all = want_mine
        if not stats_equal(res, want_all):
            raise SystemExit("the all-reduced result differs from the sum of the ranks' oracle results")
        verified = (f"bit-exact vs the oracle ({'literal per-record chain' if literal else 'rolling formulation'}) on all {total_reads} "
                    f"reads (5 scalars + 4096 bins; {verify_s:.1f} s of CPU per rank, untimed)")

    traffic, traffic_source = args.traffic_bytes, ("--traffic-bytes" if args.traffic_bytes is not None else None)
    pmc = {}
    if world == 1 and rank == 0 and not args.no_pmc and not os.environ.get("NTK_BENCH_PMC_CHILD"):
        # measured in THIS run: three separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; the SQ / GRBM counters) over short child
        # runs of this same command, per launch of the scan kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950
        pmc = pmc_passes(args, n_bytes)
        if traffic is None and "traffic" in pmc:
            traffic, traffic_source = pmc["traffic"], pmc["traffic_source"]
    if traffic is None and world == 1 and seed == SEED_C2 and (total_reads, args.read_len, args.k, args.n_per_1024) == (10_000_000, 150, 21, 1):
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_scan_kernel.json")))
        if cands:
            try:
                with open(cands[-1]) as fh:
                    pmc = json.load(fh)
                traffic = pmc["hbm_read_bytes_per_launch_corrected"] + pmc.get("hbm_write_bytes_per_launch", 0.0)
                traffic_source = "CARRIED from " + os.path.relpath(cands[-1], ROOT) + " (an earlier profile round; the live rocprofv3 --pmc passes of this run were unavailable)"
            except (OSError, KeyError, ValueError):
                traffic, traffic_source = None, None

    out = None
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
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "reads_total": total_reads, "reads_this_gpu": my_reads, "read_len": args.read_len, "k": args.k,
                "seed": hex(seed), "n_rate": f"{args.n_per_1024}/1024",
                "outputs": "n_total,n_fwd,n_rc,4096-bin prefix histogram,sum64,xor64",
                "parallelism": (f"records sharded over {world} GPUs (round-robin batches of 2^20), one ncclAllReduce(ncclUint64, ncclSum, "
                                f"{ntl.ACC_WORDS} words) per step through ntk_allreduce_accumulators") if world > 1 else "single GPU",
                "launch": {"blocks": args.blocks or "auto", "threads": args.threads or "auto"},
                "rccl_ranks": (comm.size if comm is not None else 0),   # ntk_comm_size: ranks the library's communicator spans (0 = none in use)
                "devices_visible": torch.cuda.device_count(),
                **({"per_rank": per_rank,
                    "per_rank_note": "kernel_ms = scan kernel per step, allreduce_ms = ncclAllReduce + xor rebuild per step (hipEvents on the "
                                     "ctx stream, ntk_comm_allreduce_time_ms; it includes waiting for the slowest rank's scan), ms_per_step = "
                                     "that rank's wall time per step; value and ms_per_step above use the MAX over ranks"} if per_rank else {}),
                **({"collective": collective_note} if collective_note else {}),
                **({"test_mode": f"{args.backend} backend, all ranks on cuda:0 - NOT a measurement"}
                   if (args.single_device or (use_dist and not rccl)) else {}),
            },
            "result": {"n_total": res["n_total"], "n_fwd": res["n_fwd"], "sum": hex(res["sum"]), "xor": hex(res["xor"]),
                       "verified": verified},
            "roofline": {
                **roofline_bound(pmc, achieved / HBM_PEAK_GBS, args.k),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_6290": round(achieved / 6290.0, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": n_bytes,
                "kernel_ms": round(kern_avg_ms, 5),
                "launches_timed": launches,
            },
        }
    if world == 1 and rank == 0 and seed == SEED_C2:
        out["cpu_baseline"] = None if args.no_cpu_baseline else \
            cpu_baseline(ctx, seq, args.k, args.read_len, args.n_per_1024, total_reads, args.cpu_budget_s, args.cpu_reference_bin)
        if not args.no_secondary:
            out["secondary"] = secondary_measurements(ctx, nt, torch, seq, n_bytes, total_reads, args.read_len)
    elif rank == 0:
        out["cpu_baseline"] = None   # timed at N = 1 only
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    if comm is not None:
        comm.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
