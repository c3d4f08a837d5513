"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.
Bit-exact is the bar (integer k-mers, flags, counts).  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import needletail_amd as nt  # noqa: E402
import oracle as O  # noqa: E402  (the checker)
from _fastx import bgzf_compress, fasta_raw_seqs, fastq_raw_seqs  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "these tests need a GPU"
    # kernels are enqueued on torch's current stream so that tensor initialisation (torch) and scans (library) are ordered
    c = nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def to_dev(buf: bytes):
    n = len(buf)
    t = torch.full(((n + 1023) // 1024 * 1024 + 1024,), 0x41, dtype=torch.uint8, device="cuda")  # 'A' padding: must be ignored
    if n:
        t[:n] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    return t


class redone_launches:
    """Context manager: binds an accumulator buffer of the test's own and reports NTK_ACC_REDONE (speculative launches since the last reset whose
    result came from the byte-walking kernel queued behind them) - the route a launch took is otherwise invisible in its (equal) result."""
    def __init__(self, ctx):
        self.ctx = ctx
        self.acc = torch.zeros(NL.ACC_WORDS, dtype=torch.int64, device="cuda")
    def __enter__(self):
        self.ctx.accum_bind_device(self.acc)
        return self
    def __exit__(self, *exc):
        self.ctx.accum_bind_device(None)
    def count(self):
        self.ctx.synchronize()
        return int(self.acc[NL.ACC_REDONE])


def gpu_reduce(ctx, buf: bytes, k, path, pre):
    t = to_dev(buf)
    ctx.accum_reset()
    ctx.reduce_device(t, len(buf), k, path, pre)
    return ctx.accum_read()


from contextlib import contextmanager  # noqa: E402
from needletail_amd import _lib as NL  # noqa: E402


@pytest.fixture
def restore_options(ctx):
    """Tests that switch ntk_ctx_set_option on the module's shared ctx: every option back to its default afterwards."""
    yield
    for o in (NL.OPT_COMPAT_CHUNK_BYTES, NL.OPT_MINIMIZER_CHUNK_BYTES, NL.OPT_MINIMIZER_ROUTE, NL.OPT_COMPAT_PACK_THREADS):
        ctx.set_option(o, 0)


@contextmanager
def ctx_option(c, option, value):
    """ntk_ctx_set_option for the duration of a block (the module's ctx is shared: the default is restored)."""
    c.set_option(option, value)
    try:
        yield
    finally:
        c.set_option(option, 0)


def assert_stats_equal(a, b, what=""):
    for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
        assert a[key] == b[key], (what, key, a[key], b[key])
    assert np.array_equal(a["hist"], b["hist"]), what


MODES = [  # (path, pre, canonical, tie_rc, accept_u) for the oracle's fused formulation
    (nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, True, True, True),
    (nt.PATH_BITS_CANONICAL, nt.PRE_NONE, True, False, False),
    (nt.PATH_BITS, nt.PRE_STRIP_RETURNS, False, False, False),
    (nt.PATH_BITS_CANONICAL, nt.PRE_NORMALIZE_IUPAC, True, False, True),
]


# ---- compat face: reads like the reference's unit tests ------------------------------------------

def test_normalize_kats(ctx):
    # reference src/sequence.rs:316-344, :219-224
    assert nt.normalize_opt(b"ACGTU", False, ctx) == (b"ACGTT", True)
    assert nt.normalize_opt(b"acgtu", False, ctx) == (b"ACGTT", True)
    assert nt.normalize_opt(b"N.N-N~N N", False, ctx) == (b"N-N-N-NN", True)
    assert nt.normalize_opt(b"BDHVRYSWKM", True, ctx) == (b"BDHVRYSWKM", False)
    assert nt.normalize_opt(b"bdhvryswkm", True, ctx) == (b"BDHVRYSWKM", True)
    assert nt.normalize_opt(b"BDHVRYSWKM", False, ctx) == (b"NNNNNNNNNN", True)
    assert nt.normalize_opt(b"bdhvryswkm", False, ctx) == (b"NNNNNNNNNN", True)
    assert nt.normalize(b"ADGH", False, ctx) == b"ANGN"
    assert nt.normalize(b"ADGH", True, ctx) == b"ADGH"
    assert nt.normalize(b"ACGU", True, ctx) == b"ACGT"
    assert nt.normalize(b"", False, ctx) == b""


def test_python_facade_literals(ctx):
    # reference test_python.py:101-149, :36-41
    n = lambda s, iupac=False: nt.normalize_seq(s, iupac, ctx)
    assert n("ACGTU") == "ACGTT" and n("acgtu") == "ACGTT"
    assert n("BDHVRYSWKM") == "NNNNNNNNNN" and n("BDHVRYSWKM", True) == "BDHVRYSWKM" and n("bdhvryswkm", True) == "BDHVRYSWKM"
    assert n("N-N-N-N") == "N-N-N-N" and n("N.N.N.N") == "N-N-N-N" and n("N~N~N~N") == "N-N-N-N"
    for ws in " \t\n\r":
        assert n(ws.join("NNNN")) == "NNNN"
    for junk in "!@#$%^&*|":
        assert n(junk.join("NNNN")) == "NNNNNNN"
    assert n("N9N5N1N") == "NNNNNNN"
    assert n("AGCTGYrtcga", True) == "AGCTGYRTCGA" and n("AGCTGYRTCGA") == "AGCTGNNTCGA"
    rc = lambda s: nt.reverse_complement(s, ctx)
    assert rc("a") == "t" and rc("c") == "g" and rc("g") == "c" and rc("n") == "n"
    assert rc("atcg") == "cgat" and rc("ATCG") == "CGAT"
    assert nt.reverse_complement(b"AACC", ctx) == b"GGTT"  # reference src/sequence.rs:200
    # Record.normalize (reference test_python.py:42-47)
    rec = nt.Record("test", "AGCTGYrtcga")
    rec.normalize(iupac=True)
    assert rec.seq == "AGCTGYRTCGA"
    rec.normalize()
    assert rec.seq == "AGCTGNNTCGA"


def test_normalize_strip_revcomp_random(ctx):
    rng = np.random.default_rng(3)
    for n in (1, 15, 16, 17, 4095, 4096, 4097, 70001):
        seq = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        for iupac in (False, True):
            assert nt.normalize_opt(seq, iupac, ctx) == O.normalize(seq, iupac)
        s, borrowed = O.strip_returns(seq)
        assert nt.strip_returns(seq, ctx) == (seq if borrowed else s)
        assert nt.reverse_complement(seq, ctx) == O.reverse_complement(seq)
    clean = b"ACGTNACGT-" * 1000
    assert nt.normalize_opt(clean, False, ctx) == (clean, False)
    assert nt.strip_returns(clean, ctx) is clean


def test_canonical_kmers_kats(ctx):
    # reference src/kmer.rs:171-226
    seq = b"AGCT"
    got = nt.canonical_kmers(seq, 1, nt.reverse_complement(seq, ctx), ctx)
    assert [(k, f) for _, k, f in got] == [(b"A", False), (b"C", True), (b"C", False), (b"A", True)]
    seq = b"AGCTA"
    assert [k for _, k, _ in nt.canonical_kmers(seq, 2, nt.reverse_complement(seq, ctx), ctx)] == [b"AG", b"GC", b"AG", b"TA"]
    seq = b"AGNTA"
    assert [(p, k) for p, k, _ in nt.canonical_kmers(seq, 2, nt.reverse_complement(seq, ctx), ctx)] == [(0, b"AG"), (3, b"TA")]
    # palindrome reports true on the byte path; mixed case compares raw bytes
    assert nt.canonical_kmers(b"AGCT", 4, b"AGCT", ctx) == [(0, b"AGCT", True)]
    seq = b"acgTT"
    assert nt.canonical_kmers(seq, 3, O.reverse_complement(seq), ctx) == [(0, b"acg", False), (1, b"Acg", True), (2, b"AAc", True)]


def test_bit_kmers_kats(ctx):
    # reference src/bitkmer.rs:193-251
    assert [v for _, (v, _), _ in nt.bit_kmers(b"AGCT", 1, False, ctx)] == [0, 2, 1, 3]
    assert [v for _, (v, _), _ in nt.bit_kmers(b"ACNGT", 2, False, ctx)] == [0b0001, 0b1011]
    assert [v for _, (v, _), _ in nt.bit_kmers(b"ACNG", 2, False, ctx)] == [1]
    assert [v for _, (v, _), _ in nt.bit_kmers(b"AC", 2, False, ctx)] == [1]
    assert nt.bit_kmers(b"ACGTA", 3, False, ctx) == [(0, (6, 3), False), (1, (27, 3), False), (2, (44, 3), False)]
    assert nt.bit_kmers(b"TA", 3, False, ctx) == []
    assert nt.bit_kmers(b"AGCT", 4, True, ctx) == [(0, (39, 4), False)]  # palindrome reports false on the bit path


def test_compat_iterators_random(ctx):
    rng = np.random.default_rng(17)
    alphabet = np.frombuffer(b"ACGTACGTACGTacgtNnUuRY-\n", dtype=np.uint8)
    for trial in range(40):
        n = int(rng.integers(0, 3000))
        seq = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        k = int(rng.integers(1, 33))
        canonical = bool(rng.integers(0, 2))
        p, v, f = nt.bit_kmers_arrays(seq, k, canonical, ctx)
        op, ov, of = O.bit_kmers_arrays(seq, k, canonical)
        assert np.array_equal(p, op) and np.array_equal(v, ov) and np.array_equal(f, of), (trial, n, k)
        kb = int(rng.integers(1, 80))
        p, f = nt.canonical_kmers_arrays(seq, kb, ctx)
        op, of = O.canonical_kmers_arrays(seq, O.reverse_complement(seq), kb)
        assert np.array_equal(p, op) and np.array_equal(f, of), (trial, n, kb)
    with pytest.raises(ValueError):
        nt.bit_kmers(b"ACGT", 0, True, ctx)
    with pytest.raises(ValueError):
        nt.bit_kmers(b"ACGT", 33, True, ctx)


# ---- batch face, reduce mode ------------------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 20, 21, 22, 1023, 1024, 1025, 2047, 5000, 65536 + 7])
def test_reduce_edge_lengths(ctx, n):
    rng = np.random.default_rng(n + 1)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTACGTN\n", dtype=np.uint8)
    buf = bytes(alphabet[rng.integers(0, len(alphabet), n)])
    for k in (1, 4, 16, 17, 21, 31, 32):
        for path, pre, canon, tie, u in MODES:
            want = O.reduce_fused(buf, k, canon, tie, u)
            assert_stats_equal(gpu_reduce(ctx, buf, k, path, pre), want, (n, k, path, pre))


@pytest.mark.parametrize("k", list(range(1, 33)))
def test_reduce_all_k_synthetic(ctx, k):
    buf = O.synth_reads(0x5EED0002, 0, 3000, 150, 4).tobytes()
    for path, pre, canon, tie, u in MODES[:3]:
        want = O.reduce_fused(buf, k, canon, tie, u)
        assert_stats_equal(gpu_reduce(ctx, buf, k, path, pre), want, (k, path))


def test_reduce_random_bytes_all_classes(ctx):
    rng = np.random.default_rng(99)
    for trial in range(6):
        buf = bytes(rng.integers(0, 256, 200_000, dtype=np.uint8))
        mix = np.frombuffer(b"ACGTacgtUuNn\n \t", dtype=np.uint8)
        buf2 = bytes(mix[rng.integers(0, len(mix), 200_000)])
        for b in (buf, buf2):
            for k in (3, 21, 32):
                for path, pre, canon, tie, u in MODES:
                    assert_stats_equal(gpu_reduce(ctx, b, k, path, pre), O.reduce_fused(b, k, canon, tie, u), (trial, k, path))


def test_launch_geometry_invariance(ctx):
    buf = O.synth_reads(0x5EED0002, 100, 20000, 150, 1).tobytes()
    want = O.reduce_fused(buf, 21, True, True, True)
    try:
        for blocks, threads in ((0, 1024), (1, 256), (3, 512), (7, 1024), (512, 256), (2048, 256), (100000, 256), (0, 768), (5, 896), (0, 0)):
            ctx.set_launch(blocks, threads)
            assert_stats_equal(gpu_reduce(ctx, buf, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), want, (blocks, threads))
    finally:
        ctx.set_launch(0, 0)   # back to the automatic geometry: the ctx is shared by the module's tests


def test_every_kernel_family_with_many_tiles_per_wave(ctx):
    """Few blocks on a few MB: every wave runs MANY tiles and several chunks, so whatever the compiler keeps live across the masked asm regions
    from one tile to the next (tile offsets, 64-bit carries in SCC, exec) is exercised for every kernel family - the default geometry gives
    a wave ONE tile on inputs of this size.  Regression: the forward-only and fused-minimizer regions once lost their "scc" clobber and the
    64-bit tile offset's carry with it (found by tools/gpu_fuzz.py under a forced 7-block launch, profiles/r05e/)."""
    buf = O.synth_reads(0x5EED0002, 7, 14000, 150, 1).tobytes()   # ~2.1 MB, an N every ~1024 bases
    n = len(buf)
    t = to_dev(buf)
    rng = np.random.default_rng(77)
    qual = rng.integers(33, 75, size=n, dtype=np.uint8).tobytes()
    tq = _qual_dev(qual)
    masked = O.quality_mask(buf, qual, 40)
    kmers = [(k, mode) for k in (3, 6, 7, 11, 15, 16, 17, 21, 23, 24, 31, 32) for mode in range(3)]
    mins = [(21, 11, 0), (15, 10, 0), (19, 12, 1), (23, 11, 0), (23, 12, 1), (22, 12, 0), (15, 5, 0), (19, 5, 1), (24, 11, 0), (25, 33, 1), (31, 19, 0), (27, 11, 1), (12, 5, 0), (21, 19, 1)]
    try:
        for launch in ((7, 0), (2, 512), (0, 0)):
            ctx.set_launch(*launch)
            for k, mode in kmers:
                path, pre, canon, tie_rc, accept_u = MODES[mode]
                ctx.accum_reset(); ctx.reduce_device(t, n, k, path, pre)
                assert_stats_equal(ctx.accum_read(), O.reduce_fused(buf, k, canon, tie_rc, accept_u), ("k-mers", launch, k, mode))
            for k, w, bits in mins:
                path, pre, tie, u = (nt.PATH_BITS_CANONICAL, nt.PRE_NONE, False, False) if bits else (nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, True, True)
                ctx.accum_reset(); ctx.reduce_device(t, n, k, path, pre, w=w)
                assert_stats_equal(ctx.accum_read(), O.minimizers_reduce(buf, k, w, accept_u=u, tie_rc=tie), ("minimizers", launch, k, w, bits))
            for k in (4, 21, 31):
                ctx.accum_reset(); ctx.reduce_device(t, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, d_qual=tq, quality_cutoff=40)
                assert_stats_equal(ctx.accum_read(), O.reduce_fused(masked, k, True, True, True), ("quality", launch, k))
            for k, w in ((21, 11), (23, 11)):
                ctx.accum_reset(); ctx.reduce_device(t, n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, d_qual=tq, quality_cutoff=40)
                assert_stats_equal(ctx.accum_read(), O.minimizers_reduce(masked, k, w, True, True), ("quality minimizers", launch, k, w))
    finally:
        ctx.set_launch(0, 0)


def test_reduce_on_bytes_that_were_not_normalised(ctx):
    """ntk_reduce_device / the pinned-batch face with NTK_PATH_BYTES_CANONICAL and pre = NONE or STRIP_RETURNS: the reference iterator
    compares RAW bytes (src/kmer.rs:121-128), so on mixed-case input the strand is not the smaller 2-bit value (`acgTT`, k = 3, SURVEY.md
    A.5).  Against the oracle's literal chain per record (reverse_complement -> CanonicalKmers on the bytes as they are)."""
    rng = np.random.default_rng(4242)
    def records(n_rec, lo, hi, p_lower, p_junk):
        out = []
        for _ in range(n_rec):
            n = int(rng.integers(lo, hi))
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
            m = rng.random(n)
            a[m < p_lower] |= 0x20
            j = m > 1 - p_junk
            a[j] = np.frombuffer(b"NnUuRYKM-.*\x00\xff", dtype=np.uint8)[rng.integers(0, 13, int(j.sum()))]
            out.append(a.tobytes())
        return out
    assert [x for x in O.canonical_kmers(b"acgTT", O.reverse_complement(b"acgTT"), 3)] == [(0, b"acg", False), (1, b"Acg", True), (2, b"AAc", True)]
    sets = [records(300, 0, 400, 0.3, 0.02), records(40, 3000, 9000, 0.5, 0.001), records(200, 1, 80, 0.1, 0.1), [b"acgTT", b"", b"A", b"ACGT" * 50, b"acgt" * 50]]
    try:
        for si, recs in enumerate(sets):
            buf = b"\n".join(recs) + b"\n"
            t = to_dev(buf)
            for k in (1, 3, 4, 6, 7, 11, 16, 17, 21, 31, 32):
                for pre in (nt.PRE_NONE, nt.PRE_STRIP_RETURNS):
                    want = O.reduce_records(recs, k, nt.PATH_BYTES_CANONICAL, pre)
                    for launch in ((0, 0), (3, 0)):
                        ctx.set_launch(*launch)
                        ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, nt.PATH_BYTES_CANONICAL, pre)
                        assert_stats_equal(ctx.accum_read(), want, ("raw bytes", si, k, pre, launch))
        ctx.set_launch(0, 0)
        # where every base has one case the packed-value scan and the raw-byte kernel must agree
        up = b"\n".join(r.upper() for r in sets[0]) + b"\n"
        a = gpu_reduce(ctx, up, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
        b = gpu_reduce(ctx, up, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
        nou = O.reduce_records([r.upper() for r in sets[0]], 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
        assert_stats_equal(a, nou, "upper case, raw kernel")
        if b"U" not in up:
            assert_stats_equal(b, nou, "upper case, packed-value scan")
        # lower case in the padding behind the input's last byte is nobody's base (the speculative kernels keep it out of their bit-5 watch)
        cl = b"\n".join(records(120, 0, 400, 0.0, 0.0)) + b"\n"   # ACGT and the separator only: no byte of the input has bit 5 set
        for n_cut in (len(cl), len(cl) - 3, len(cl) - 18):
            t = to_dev(cl[:n_cut]); t[n_cut:] = 0x61
            recs_cut = cl[:n_cut].split(b"\n")
            ctx.reduce_device(t, n_cut, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, reset=True)
            assert_stats_equal(ctx.accum_read(), O.reduce_records(recs_cut, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE), ("padding", n_cut))
            ctx.reduce_device(t, n_cut, 40, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, reset=True)
            assert_stats_equal(ctx.accum_read(), _wide_reference(recs_cut, 40, False), ("padding, k = 40", n_cut))
            with redone_launches(ctx) as rl:   # ... and neither launch was handed to the byte-walking kernel
                ctx.reduce_device(t, n_cut, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, reset=True)
                ctx.reduce_device(t, n_cut, 40, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
                assert rl.count() == 0, n_cut
                t[n_cut - 2] |= 0x20   # one lower-case base INSIDE the input: both are
                ctx.reduce_device(t, n_cut, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, reset=True)
                ctx.reduce_device(t, n_cut, 40, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
                assert rl.count() == 2, n_cut
        # the pinned-batch face with the reset flag
        st = _run_records(ctx, sets[0], 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
        assert_stats_equal(st, O.reduce_records(sets[0], 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE), "batch face")
    finally:
        ctx.set_launch(0, 0)


def test_speculative_scan_of_bytes_that_were_not_normalised(ctx):
    """Round 6: un-normalised byte-path input is scanned by the packed-value build that watches for lower case; the raw-byte kernel queued
    behind it redoes the launch only if a byte with bit 5 was seen (reference src/sequence.rs:57-61: normalize returns None on a clean read;
    src/kmer.rs:121-128: the compare is on raw bytes).  Both routes (speculation on / off) against the oracle's literal chain, on launches
    that alternate between clean and soft-masked batches - each launch has its own flag word in a ring that later launches re-arm."""
    rng = np.random.default_rng(606)
    def reads(n_rec, p_lower, extra=b""):
        out = []
        for _ in range(n_rec):
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 150)].copy()
            a[rng.random(150) < 1 / 256] = ord("N")
            a[rng.random(150) < p_lower] |= 0x20
            out.append(a.tobytes())
        return out + ([extra] if extra else [])
    clean, soft = reads(700, 0.0), reads(700, 0.02)
    one_low = [r for r in clean]; one_low[-1] = one_low[-1][:-1] + one_low[-1][-1:].lower()       # ONE lower-case base, in the last tile
    first_low = [clean[0][:1].lower() + clean[0][1:]] + clean[1:]                                  # ... in the first
    false_pos = reads(300, 0.0, b"ACGT-ACGT.ACGT*9 nACGTACGTACGTACGTACGTACGTACGT")                 # bit 5 on bytes that are no bases: the slow route, same result
    sets = {"clean": clean, "soft": soft, "one_low": one_low, "first_low": first_low, "false_pos": false_pos}
    bufs = {name: (b"\n".join(r) + b"\n") for name, r in sets.items()}
    devs = {name: to_dev(b) for name, b in bufs.items()}
    order = ["clean", "soft", "clean", "clean", "one_low", "first_low", "clean", "false_pos", "soft", "soft", "clean"]
    try:
        for k in (4, 16, 21, 31, 32):
            want = {name: O.reduce_records(r, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE) for name, r in sets.items()}
            for route in (0, NL.ROUTE_NO_SPECULATION):
                ctx.set_option(NL.OPT_MINIMIZER_ROUTE, route)
                for geometry in ((0, 0), (2, 0)):
                    ctx.set_launch(*geometry)
                    for name in order:
                        ctx.reduce_device(devs[name], len(bufs[name]), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, reset=True)
                        assert_stats_equal(ctx.accum_read(), want[name], ("speculative scan", k, route, geometry, name))
        ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0); ctx.set_launch(0, 0)
        # more launches than the ring has flag words, accumulating (no reset): every third one soft-masked
        k = 21
        ctx.accum_reset()
        tot = {key: 0 for key in ("n_total", "n_fwd", "n_rc", "sum", "xor")}; hist = np.zeros(4096, dtype=np.uint64)
        wc, ws = (O.reduce_records(sets[x], k, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE) for x in ("clean", "soft"))
        for i in range(150):
            name, w = ("soft", ws) if i % 3 == 1 else ("clean", wc)
            ctx.reduce_device(devs[name], len(bufs[name]), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE)
            for key in ("n_total", "n_fwd", "n_rc"): tot[key] += w[key]
            tot["sum"] = (tot["sum"] + w["sum"]) & (2**64 - 1); tot["xor"] ^= w["xor"]; hist += w["hist"]
        tot["hist"] = hist
        assert_stats_equal(ctx.accum_read(), tot, "150 launches on one ring")
        # the pinned-batch face: clean and soft-masked records in separate batches and mixed
        for name, recs in (("clean", clean), ("soft", soft), ("mixed", clean[:300] + soft[:300] + clean[300:])):
            st = _run_records(ctx, recs, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_STRIP_RETURNS, batch_bytes=1 << 15)
            assert_stats_equal(st, O.reduce_records(recs, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_STRIP_RETURNS), ("batch face", name))
    finally:
        ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0); ctx.set_launch(0, 0)


def _wide_reference(recs, k, normalized):
    """CanonicalKmers with 33 <= k <= 255 per record through the oracle's literal iterator: counters + the histogram of the leading six
    bases of every emitted slice (the 2-bit value itself has more than 64 bits: no sum / xor)."""
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"): code[ch] = i; code[ch | 0x20] = i
    st = {"n_total": 0, "n_fwd": 0, "n_rc": 0, "sum": 0, "xor": 0, "hist": np.zeros(4096, dtype=np.uint64)}
    for r in recs:
        if normalized:
            r = O.normalize(r)[0]
        rc = O.reverse_complement(r)
        pos, flg = O.canonical_kmers_arrays(r, rc, k)
        for p, f in zip(pos.tolist(), flg.tolist()):
            sl = rc[len(rc) - p - k: len(rc) - p] if f else r[p: p + k]
            b = 0
            for ch in sl[:6]: b = b * 4 + int(code[ch])
            st["hist"][b] += 1
        st["n_total"] += len(pos); st["n_rc"] += int(flg.sum()); st["n_fwd"] += len(pos) - int(flg.sum())
    return st


def test_k_above_32_on_the_reduce_face(ctx, golden_dir):
    """Round 6: CanonicalKmers takes k: u8 (reference src/kmer.rs:48-82, src/sequence.rs:237-239); for 33 <= k <= 255 the device-resident
    reduce face counts the items, splits them by strand and bins them by their leading six bases; sum / xor are not defined on values of
    more than 64 bits (ntk_result.n_undigested says how many k-mers they do not cover)."""
    rng = np.random.default_rng(3355)
    recs28 = fasta_raw_seqs(open(os.path.join(golden_dir, "28S.fasta"), "rb").read())
    rnd = []
    for _ in range(60):
        n = int(rng.integers(0, 900))
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        m = rng.random(n)
        a[m < 0.2] |= 0x20
        a[m > 0.997] = np.frombuffer(b"NnUu-", dtype=np.uint8)[rng.integers(0, 5, int((m > 0.997).sum()))]
        rnd.append(a.tobytes())
    pal = [b"ACGT" * 80, b"acgt" * 80, b"AT" * 40 + b"at" * 40, b"A" * 300 + b"T" * 300]   # reverse-complement palindromes: ties -> rc
    # upper case only: the input on which the packed-stream kernel's result stands (no lower case, no k-mer equal to its reverse complement
    # over 32 bases) - record lengths around k and around the 16-byte slots / 4096-byte tiles of wide_canonical_reduce_kernel, breaks anywhere
    clean = []
    for n in [0, 1, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 254, 255, 256, 257, 271, 272, 4095, 4096, 4097, 4351, 4352, 9000, 20000] + [int(x) for x in rng.integers(0, 1500, 40)]:
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        a[rng.random(n) > 0.9985] = ord("N")
        clean.append(a.tobytes())
    try:
        for name, recs in (("28S", recs28), ("random", rnd), ("palindromes", pal), ("clean", clean)):
            buf = b"\n".join(r.replace(b"\n", b"").replace(b"\r", b"") for r in recs) + b"\n"
            flat = [r.replace(b"\n", b"").replace(b"\r", b"") for r in recs]
            t = to_dev(buf)
            for k in (33, 64, 255):
                for pre, normalized in ((nt.PRE_NONE, False), (nt.PRE_NORMALIZE, True)):
                    want = _wide_reference(flat, k, normalized)
                    for geometry in ((0, 0), (3, 0)):
                        ctx.set_launch(*geometry)
                        ctx.reduce_device(t, len(buf), k, nt.PATH_BYTES_CANONICAL, pre, reset=True)
                        got = ctx.accum_read()
                        assert_stats_equal(got, want, ("k > 32", name, k, pre, geometry))
                        assert got["n_undigested"] == got["n_total"]
                    with redone_launches(ctx) as rl:   # which kernel's result it was: the packed-stream one unless the batch holds lower case (input not normalised) or an inverted repeat
                        ctx.reduce_device(t, len(buf), k, nt.PATH_BYTES_CANONICAL, pre, reset=True)
                        bit5 = any(c & 0x20 for c in set(buf)) and not normalized   # (the watch is on every byte of the input, base or not)
                        # (ACGT)n, (AT)n: an even-length window can equal its reverse complement, an odd one cannot; A..AT..T: a window with >= 32 of each
                        # agrees with its reverse complement over the first 32 bases whatever its length
                        tie32 = name == "palindromes" and (k % 2 == 0 or k >= 65)
                        assert rl.count() == (1 if bit5 or tie32 else 0), (name, k, pre)
                    if name in ("clean", "palindromes"):   # the byte-walking kernel alone (the route the pair falls back to) says the same
                        try:
                            ctx.set_option(NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_NO_SPECULATION)
                            ctx.reduce_device(t, len(buf), k, nt.PATH_BYTES_CANONICAL, pre, reset=True)
                            assert_stats_equal(ctx.accum_read(), want, ("k > 32, direct route", name, k, pre))
                        finally:
                            ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0)
            for k in (40, 49, 97, 128, 200):   # other word / bit offsets of the window's start
                want = _wide_reference(flat, k, True)
                ctx.reduce_device(t, len(buf), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, reset=True)
                assert_stats_equal(ctx.accum_read(), want, ("k > 32", name, k))
        ctx.set_launch(0, 0)
        # a k <= 32 scan after it: digests again, nothing undigested
        buf = b"\n".join(rnd) + b"\n"
        ctx.reduce_device(to_dev(buf), len(buf), 31, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, reset=True)
        got = ctx.accum_read()
        assert_stats_equal(got, O.reduce_records(rnd, 31, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE), "k = 31 after k > 32")
        assert got["n_undigested"] == 0
        # the pinned-batch face at k = 64
        st = _run_records(ctx, rnd, 64, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1 << 14)
        assert_stats_equal(st, _wide_reference(rnd, 64, True), "batch face, k = 64")
        # everything else stays k <= 32
        t = to_dev(b"ACGT" * 100)
        vals = torch.zeros(512, dtype=torch.int64, device="cuda"); v16 = torch.zeros(64, dtype=torch.int16, device="cuda"); r16 = torch.zeros_like(v16)
        for call in (lambda: ctx.reduce_device(t, 400, 256, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE),
                     lambda: ctx.reduce_device(t, 400, 33, nt.PATH_BITS_CANONICAL, nt.PRE_NONE),
                     lambda: ctx.reduce_device(t, 400, 33, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=5),
                     lambda: ctx.materialize_device(t, 400, 33, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, vals, v16, r16)):
            with pytest.raises(nt.NtkError) as e:
                call()
            assert e.value.status == 1
    finally:
        ctx.set_launch(0, 0)


def test_ctx_options_round_trip(ctx):
    """ntk_ctx_set_option / ntk_ctx_get_option (round 6: the producer reads its knobs through the getter; nothing reads the environment)."""
    try:
        assert ctx.get_option(NL.OPT_BATCH_WAIT_POLL_US) == 50 and ctx.get_option(NL.OPT_GZ_STREAM_WINDOW_BYTES) == 0
        for opt, val, back in ((NL.OPT_BATCH_WAIT_POLL_US, 200, 200), (NL.OPT_BATCH_WAIT_POLL_US, NL.POLL_BLOCK, NL.POLL_BLOCK),
                               (NL.OPT_BATCH_WAIT_POLL_US, 10**9, 10_000_000), (NL.OPT_GZ_STREAM_WINDOW_BYTES, 64 << 20, 64 << 20),
                               (NL.OPT_GZ_INMEM_LIMIT_BYTES, 1 << 33, 1 << 33), (NL.OPT_PIPE_STATS, 1, 1),
                               (NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_NO_SPECULATION | NL.ROUTE_NO_F64, NL.ROUTE_NO_SPECULATION | NL.ROUTE_NO_F64)):
            ctx.set_option(opt, val)
            assert ctx.get_option(opt) == back, (opt, val)
        with pytest.raises(nt.NtkError):
            ctx.get_option(99)
        with pytest.raises(nt.NtkError):
            ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 16)
    finally:
        for opt in (NL.OPT_BATCH_WAIT_POLL_US, NL.OPT_GZ_STREAM_WINDOW_BYTES, NL.OPT_GZ_INMEM_LIMIT_BYTES, NL.OPT_PIPE_STATS, NL.OPT_MINIMIZER_ROUTE):
            ctx.set_option(opt, 0)
    # a blocking wait (NTK_POLL_BLOCK) through the pinned-batch face gives the same result as the polling one
    recs = [b"ACGTTGCAAGCTTGCATGCAAGTCGATCGATTAGC" * 3] * 50
    want = O.reduce_records(recs, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
    try:
        ctx.set_option(NL.OPT_BATCH_WAIT_POLL_US, NL.POLL_BLOCK)
        assert_stats_equal(_run_records(ctx, recs, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1 << 12), want, "blocking wait")
    finally:
        ctx.set_option(NL.OPT_BATCH_WAIT_POLL_US, 0)


def test_unsupported_and_bad_args_are_errors(ctx):
    t = to_dev(b"ACGT" * 100)
    # un-normalised byte-path input: reduce mode has its raw-byte kernel (test_reduce_on_bytes_that_were_not_normalised); dense values and
    # windowed minimizers on such input are not built and say so
    with pytest.raises(nt.NtkError) as e:
        ctx.reduce_device(t, 400, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, w=11)
    assert e.value.status == 6
    vals = torch.zeros(512, dtype=torch.int64, device="cuda"); v16 = torch.zeros(64, dtype=torch.int16, device="cuda"); r16 = torch.zeros_like(v16)
    with pytest.raises(nt.NtkError) as e:
        ctx.materialize_device(t, 400, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NONE, vals, v16, r16)
    assert e.value.status == 6
    for k in (0, 33):
        with pytest.raises(nt.NtkError) as e:
            ctx.reduce_device(t, 400, k, nt.PATH_BITS, nt.PRE_NONE)
        assert e.value.status == 1
    with pytest.raises(nt.NtkError) as e:
        ctx.reduce_device(t.data_ptr() + 1, 100, 5, nt.PATH_BITS, nt.PRE_NONE)
    assert e.value.status == 2


# ---- whole-file pins through the pinned-batch face (reference benches/benchmark.rs:43-44,66-67) -----------

def _run_records(ctx, recs, k, path, pre, batch_bytes=1 << 18):
    ctx.accum_reset()
    batches = [ctx.batch(batch_bytes, 4096) for _ in range(3)]
    cur = 0
    for r in recs:
        if not batches[cur].append(r, pre):
            batches[cur].submit(k, path, pre)
            cur = (cur + 1) % len(batches)
            batches[cur].wait()
            assert batches[cur].append(r, pre)
    batches[cur].submit(k, path, pre)
    for b in batches:
        b.wait()
        b.release()
    return ctx.accum_read()


def test_28s_whole_file_pins(ctx, golden_dir):
    recs = fasta_raw_seqs(open(os.path.join(golden_dir, "28S.fasta"), "rb").read())
    st = _run_records(ctx, recs, 31, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE_IUPAC)
    assert (st["n_total"], st["n_fwd"]) == (718_007, 350_983)
    st = _run_records(ctx, recs, 31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS)
    assert (st["n_total"], st["n_fwd"]) == (718_007, 350_983)
    assert_stats_equal(st, O.reduce_records(recs, 31, O.PATH_BITS_CANONICAL, O.PRE_STRIP_RETURNS), "28S bits")
    # README program (reference src/lib.rs:11-38): k=4 AAAA count
    st = _run_records(ctx, recs, 4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    assert st["hist"][0] == 8_108
    assert_stats_equal(st, O.reduce_records(recs, 4, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), "28S readme")


def test_fastq_head_pins(ctx, golden_dir):
    recs = fastq_raw_seqs(open(os.path.join(golden_dir, "PRJNA271013_head.fq"), "rb").read())
    st = _run_records(ctx, recs, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    assert_stats_equal(st, O.reduce_records(recs, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), "fq bytes")
    assert st["n_total"] == 209_965 and st["sum"] == 0x047AD82A7ED0CABA


def test_scan_file_pipeline(ctx, golden_dir, tmp_path):
    """The README program end to end (reference src/lib.rs:15-35): CPU parser -> pinned batches -> overlapped H2D + scan."""
    import gzip
    fa = os.path.join(golden_dir, "28S.fasta")
    recs = fasta_raw_seqs(open(fa, "rb").read())
    for batch_bytes in (1 << 14, 1 << 20):   # tiny batches force many submit/wait rotations
        st = nt.scan_file(ctx, fa, 4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=batch_bytes, n_batches=3)
        assert st["n_records"] == 570 and st["n_bases"] == 738_580 and st["hist"][0] == 8_108
        assert_stats_equal(st, O.reduce_records(recs, 4, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), "scan_file 28S")
    st = nt.scan_file(ctx, fa, 31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS)
    assert (st["n_total"], st["n_fwd"]) == (718_007, 350_983)
    fq = os.path.join(golden_dir, "PRJNA271013_head.fq")
    gz = tmp_path / "head.fq.gz"
    gz.write_bytes(gzip.compress(open(fq, "rb").read()))
    st = nt.scan_file(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1 << 16)
    assert st["n_records"] == 2000 and st["n_bases"] == 250_000 and st["n_total"] == 209_965 and st["sum"] == 0x047AD82A7ED0CABA
    # BASELINE.json configs[4]: gzip FASTQ stream + minimizers (w=11, k=21) through the same pipeline
    recs_fq = fastq_raw_seqs(open(fq, "rb").read())
    st = nt.scan_file(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1 << 16, w=11)
    assert_stats_equal(st, O.minimizers_reduce(b"".join(r + b"\n" for r in recs_fq), 21, 11, True, True), "gz minimizers")
    # parallel producer (one parser thread per file range) gives the same reduced result
    for threads in (1, 3, 16):
        stp = nt.scan_file_parallel(ctx, fq, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=threads, batch_bytes=1 << 16)
        assert stp["n_records"] == 2000 and stp["n_bases"] == 250_000
        assert_stats_equal(stp, O.reduce_records(recs_fq, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), ("parallel", threads))
    stp = nt.scan_file_parallel(ctx, fa, 31, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS, threads=8, batch_bytes=1 << 18)
    assert (stp["n_records"], stp["n_total"], stp["n_fwd"]) == (570, 718_007, 350_983)
    # gzip through the parallel producer: the whole file (every member) is inflated into memory by libdeflate when that
    # library can be loaded, then parsed in parallel; an in-memory gzip BUFFER is refused (a gzip stream is sequential)
    seq_st = nt.scan_file(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1 << 16)
    multi = tmp_path / "multi.fq.gz"
    fq_lines = open(fq, "rb").read().split(b"\n")
    cut = (len(fq_lines) // 8) * 4
    multi.write_bytes(gzip.compress(b"\n".join(fq_lines[:cut]) + b"\n") + gzip.compress(b"\n".join(fq_lines[cut:])))
    try:
        import ctypes
        ctypes.CDLL("libdeflate.so.0")
        have_libdeflate = True
    except OSError:
        have_libdeflate = False
    if have_libdeflate:
        bgzf = tmp_path / "blocks.fq.gz"   # block gzip: members located by their 'BC' size field and inflated in parallel
        bgzf.write_bytes(bgzf_compress(open(fq, "rb").read(), block=20000))
        assert gzip.decompress(bgzf.read_bytes()) == open(fq, "rb").read()
        for path_ in (gz, multi, bgzf):
            par = nt.scan_file_parallel(ctx, str(path_), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=4, batch_bytes=1 << 16)
            assert_stats_equal(par, seq_st, f"gzip via the parallel producer: {path_.name}")
            assert par["n_records"] == seq_st["n_records"]
        bad = tmp_path / "bad.fq.gz"
        blob = bytearray(gz.read_bytes()); blob[len(blob) // 2] ^= 0x55
        bad.write_bytes(bytes(blob))
        with pytest.raises(nt.NtkError) as e:   # corrupted stream (libdeflate verifies the CRC-32 of every member)
            nt.scan_file_parallel(ctx, str(bad), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
        assert e.value.status == 8
    else:
        with pytest.raises(nt.NtkError) as e:
            nt.scan_file_parallel(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, streaming_fallback=False)
        assert e.value.status == 6
        assert_stats_equal(nt.scan_file_parallel(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), seq_st, "fallback")
    with pytest.raises(nt.NtkError) as e:
        nt.scan_file_parallel(ctx, None, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, data=gz.read_bytes())
    assert e.value.status == 6
    # records longer than a whole batch go through a one-off batch of their own: batch_bytes is a knob, not a limit
    st = nt.scan_file(ctx, fa, 4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1024)
    assert_stats_equal(st, O.reduce_records(recs, 4, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), "oversized records")
    stp = nt.scan_file_parallel(ctx, fa, 4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=4, batch_bytes=1024)
    assert_stats_equal(stp, st, "oversized records, parallel")


# ---- minimizers and quality mask ---------------------------------------------------------

def test_minimizer_and_quality_mask_kats(ctx):
    assert nt.minimizer(b"ATTTCG", 3, ctx) == b"AAA"                       # reference src/sequence.rs:363-367
    assert nt.quality_mask(b"AGCT", b"AAA0", ord("5"), ctx) == b"AGCN"     # reference src/sequence.rs:369-374
    got = nt.bit_minimizers([0b001011, 0b001011, 0b110001], 3, 2, ctx)      # reference src/bitkmer.rs:261-267
    assert list(got) == [0b0010, 0b0010, 1]
    assert list(nt.bit_minimizers([0b001011], 3, 1, ctx)) == [0] and list(nt.bit_minimizers([0b11000011], 4, 2, ctx)) == [0]
    rng = np.random.default_rng(21)
    alphabet = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    for trial in range(25):
        n = int(rng.integers(1, 4000))
        seq = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        m = int(rng.integers(1, min(n, 40) + 1))
        assert nt.minimizer(seq, m, ctx) == O.minimizer(seq, m), (trial, n, m)
        q = bytes(rng.integers(33, 80, n, dtype=np.uint8))
        assert nt.quality_mask(seq, q, 53, ctx) == O.quality_mask(seq, q, 53)
    vals = rng.integers(0, 1 << 62, 5000, dtype=np.uint64)
    for k, m in ((31, 21), (32, 32), (21, 1), (16, 9)):
        v = vals & np.uint64((1 << (2 * k)) - 1 if k < 32 else 0xFFFFFFFFFFFFFFFF)
        want = np.array([O.bit_minimizer(int(x), k, m) for x in v[:300]], dtype=np.uint64)
        assert np.array_equal(nt.bit_minimizers(v, k, m, ctx)[:300], want), (k, m)


def test_sequence_canonical_kats(ctx):
    # reference src/sequence.rs:354-361
    for seq, want in ((b"A", b"A"), (b"T", b"A"), (b"AAGT", b"AAGT"), (b"ACTT", b"AAGT"), (b"GC", b"GC"), (b"", b"")):
        assert nt.canonical(seq, ctx) == want == O.canonical(seq)
    rng = np.random.default_rng(8)
    for _ in range(50):
        s_ = bytes(rng.choice(list(b"ACGTacgtNRY"), size=int(rng.integers(1, 300))).astype(np.uint8))
        assert nt.canonical(s_, ctx) == O.canonical(s_)
    assert nt.mask_header_tabs(b"a\tb\tc") == b"a|b|c" and nt.mask_header_tabs(b"abc") is None
    assert nt.mask_header_utf8(b"ok") is None and nt.mask_header_utf8(b"bad\xff") == "bad\ufffd".encode()


def test_bit_reverse_complement_and_canonical_kats(ctx):
    # reference src/bitkmer.rs:253-259, 270-286
    assert list(nt.bit_reverse_complement([0b000000, 0b111111], 3, ctx)) == [0b111111, 0]
    assert list(nt.bit_reverse_complement([0, 0b00011011], 4, ctx)) == [0b11111111, 0b00011011]
    assert nt.bytes_to_bitmer(b"C") == (1, 1) and nt.bytes_to_bitmer(b"TTA") == (60, 3) and nt.bytes_to_bitmer(b"AAA") == (0, 3)
    assert nt.bitmer_to_bytes(1, 1) == b"C" and nt.bitmer_to_bytes(60, 3) == b"TTA" and nt.bitmer_to_bytes(0, 3) == b"AAA"
    rng = np.random.default_rng(8)
    for k in (1, 4, 21, 31, 32):
        v = rng.integers(0, 1 << 63, 2000, dtype=np.uint64) & np.uint64((1 << (2 * k)) - 1 if k < 32 else 0xFFFFFFFFFFFFFFFF)
        rc = nt.bit_reverse_complement(v, k, ctx)
        can, flg = nt.bit_canonical(v, k, ctx)
        for i in range(200):
            assert int(rc[i]) == O.bit_reverse_complement(int(v[i]), k)
            assert (int(can[i]), bool(flg[i])) == O.bit_canonical(int(v[i]), k)


def _fastq_text(host: np.ndarray, n_reads: int, L: int, seed: int) -> bytes:
    """FASTQ text of synthetic reads (host = the packed batch layout: L bases + a break byte per read) with random qualities, so that a
    zlib-6 stream of it is made of many dynamic blocks and is not tiny."""
    rng = np.random.default_rng(seed)
    seqs = host.reshape(n_reads, L + 1)[:, :L]
    hdr = np.frombuffer(b"@r%08d\n" % 0, dtype=np.uint8)
    rec = np.empty((n_reads, len(hdr) + L + 3 + L + 1), dtype=np.uint8)
    rec[:, : len(hdr)] = hdr
    idx = np.arange(n_reads)
    for d in range(8):
        rec[:, 2 + 7 - d] = 48 + (idx // 10**d) % 10
    o = len(hdr)
    rec[:, o: o + L] = seqs
    rec[:, o + L: o + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, o + L + 3: o + 2 * L + 3] = rng.integers(33, 74, (n_reads, L), dtype=np.uint8)
    rec[:, -1] = 10
    return rec.tobytes()


def test_gzip_file_streamed_through_the_parallel_producer(ctx, tmp_path):
    """BASELINE.json configs[4] on its fast route: a >= 64 MB ONE-member zlib-6 FASTQ through ntk_scan_file_parallel -
    speculative parallel inflate (route 2) consumed WHILE it runs by the parser threads -> pinned batches -> H2D -> fused (21, 11)
    minimizers / the k = 21 scan, against the oracle on the plain text.  Also: a small window (the inflater waits for the parsers all the
    time), several members, block gzip, a truncated and a corrupt stream (reference src/parser/mod.rs:95-108: MultiGzDecoder reads every
    member, truncation is an error), and a record longer than the window."""
    import zlib
    from needletail_amd import _lib as NL2
    def gzip_member(data: bytes, level: int) -> bytes:
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()
    n_reads, L = 410_000, 150
    host = O.synth_reads(0x5EED0002, 0, n_reads, L, 2)
    text = _fastq_text(host, n_reads, L, 64)
    z = gzip_member(text, 6)
    assert len(z) >= 64 << 20, len(z)
    gz = tmp_path / "c2_prefix.fq.gz"
    gz.write_bytes(z)
    def minimizers_oracle(buf: np.ndarray, parts: int = 8) -> dict:   # read-aligned parts on threads (the C oracle runs outside the GIL): windows never span reads
        from concurrent.futures import ThreadPoolExecutor
        per = (n_reads + parts - 1) // parts * (L + 1)
        with ThreadPoolExecutor(parts) as ex:
            rs = list(ex.map(lambda i: O.minimizers_reduce(buf[i * per: (i + 1) * per].tobytes(), 21, 11, True, True), range(parts)))
        tot = {key: sum(r[key] for r in rs) for key in ("n_total", "n_fwd", "n_rc")}
        tot["sum"] = sum(r["sum"] for r in rs) & (2**64 - 1)
        tot["xor"] = 0
        for r in rs: tot["xor"] ^= r["xor"]
        tot["hist"] = sum(r["hist"] for r in rs)
        return tot
    want_min = minimizers_oracle(host)
    want_k = O.reduce_fused(host, 21, True, True, True)
    try:
        for window in (0, 8 << 20):
            ctx.set_option(NL2.OPT_GZ_STREAM_WINDOW_BYTES, window)
            st = nt.scan_file_parallel(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=16, batch_bytes=4 << 20, w=11)
            g = st["gzip"]
            assert (g["route"], g["streamed"], g["members"]) == (2, 1, 1) and g["chunks"] >= 16 and g["text_bytes"] == len(text), g
            assert g["peak_backlog_bytes"] <= (window or (512 << 20)) + (128 << 20), g      # the window + one chunk's text
            assert st["n_records"] == n_reads and st["n_bases"] == n_reads * L
            assert_stats_equal(st, want_min, ("streamed gzip, (21, 11) minimizers", window))
            assert 0 < g["first_batch_s"] < g["total_s"]
        ctx.set_option(NL2.OPT_GZ_STREAM_WINDOW_BYTES, 0)
        st = nt.scan_file_parallel(ctx, str(gz), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=5, batch_bytes=1 << 20)
        assert st["gzip"]["route"] == 2 and st["gzip"]["streamed"] == 1
        assert_stats_equal(st, want_k, "streamed gzip, k = 21")
        # one thread: inflate, then parse (route 3, not streamed)
        small_reads = 30_000
        small_text = _fastq_text(host[: small_reads * (L + 1)], small_reads, L, 65)
        small_want = O.reduce_fused(host[: small_reads * (L + 1)], 21, True, True, True)
        one = tmp_path / "small.fq.gz"
        one.write_bytes(gzip_member(small_text, 6))
        st = nt.scan_file_parallel(ctx, str(one), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=1, batch_bytes=1 << 18)
        assert (st["gzip"]["route"], st["gzip"]["streamed"]) == (3, 0)
        assert_stats_equal(st, small_want, "one thread")
        # several members (cut at record boundaries and in the middle of a record), and block gzip
        rec_len = len(small_text) // small_reads
        cuts = [0, 7000 * rec_len, 7000 * rec_len + 40, 19_000 * rec_len + 200, len(small_text)]
        multi = tmp_path / "multi.fq.gz"
        multi.write_bytes(b"".join(gzip_member(small_text[a:b], lvl) for (a, b), lvl in zip(zip(cuts, cuts[1:]), (6, 1, 9, 4))))
        st = nt.scan_file_parallel(ctx, str(multi), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=6, batch_bytes=1 << 18)
        assert (st["gzip"]["route"], st["gzip"]["streamed"], st["gzip"]["members"]) == (2, 1, 4)
        assert_stats_equal(st, small_want, "four members")
        try:
            import ctypes
            ctypes.CDLL("libdeflate.so.0")
            bg = tmp_path / "blocks.fq.gz"
            bg.write_bytes(bgzf_compress(small_text, block=60000))
            ctx.set_option(NL2.OPT_GZ_STREAM_WINDOW_BYTES, 8 << 20)
            st = nt.scan_file_parallel(ctx, str(bg), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=6, batch_bytes=1 << 18)
            assert (st["gzip"]["route"], st["gzip"]["streamed"]) == (1, 1) and st["gzip"]["text_bytes"] == len(small_text)
            assert_stats_equal(st, small_want, "block gzip, streamed")
        except OSError:
            pass
        ctx.set_option(NL2.OPT_GZ_STREAM_WINDOW_BYTES, 0)
        # truncated / corrupt: NTK_ERR_PARSE, whatever the decoder had delivered before it got there
        zs = one.read_bytes()
        for name, bad in (("cut in the middle", z[: len(z) // 2]), ("trailer cut", z[:-5]), ("small, cut", zs[: len(zs) * 2 // 3]),
                          ("bit flip", z[: len(z) // 3] + bytes([z[len(z) // 3] ^ 0x10]) + z[len(z) // 3 + 1:]),
                          ("last member cut", multi.read_bytes()[:-9])):
            badf = tmp_path / "bad.fq.gz"
            badf.write_bytes(bad)
            with pytest.raises(nt.NtkError) as e:
                nt.scan_file_parallel(ctx, str(badf), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=8, batch_bytes=1 << 20, streaming_fallback=False)
            assert e.value.status == 8, name
        # the TEXT is bad (a FASTQ record without its quality line in the middle of a good stream): the parsers give up, the inflater is
        # told to stop (it must not run on into memory nobody drains) and the call returns the parse error
        cut_at = (n_reads // 2) * (len(text) // n_reads)
        broken = text[:cut_at] + b"@broken\nACGT\n" + text[cut_at:]
        badf = tmp_path / "badtext.fq.gz"
        badf.write_bytes(gzip_member(broken, 1))
        with pytest.raises(nt.NtkError) as e:
            nt.scan_file_parallel(ctx, str(badf), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=8, batch_bytes=1 << 20, streaming_fallback=False)
        assert e.value.status == 8
        st = nt.scan_file_parallel(ctx, str(one), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=4, batch_bytes=1 << 18)   # and the ctx is fine afterwards
        assert_stats_equal(st, small_want, "after a cancelled run")
        # a record longer than the window (a 20 MB contig against an 8 MiB window): the window grows instead of the run stalling
        rng = np.random.default_rng(9)
        contig = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 20 << 20)].tobytes()
        fa_text = b">short\nACGTACGTACGTACGTACGTACGTACGT\n>long\n" + contig + b"\n>tail\nTTTTTTTTTTTTTTTTTTTTTTTTTGGGG\n"
        fa = tmp_path / "contig.fa.gz"
        fa.write_bytes(gzip_member(fa_text, 1))
        ctx.set_option(NL2.OPT_GZ_STREAM_WINDOW_BYTES, 8 << 20)
        st = nt.scan_file_parallel(ctx, str(fa), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=4, batch_bytes=1 << 20)
        recs = [b"ACGTACGTACGTACGTACGTACGTACGT", contig, b"TTTTTTTTTTTTTTTTTTTTTTTTTGGGG"]
        assert st["n_records"] == 3
        assert_stats_equal(st, O.reduce_records(recs, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE), "a record longer than the window")
    finally:
        ctx.set_option(NL2.OPT_GZ_STREAM_WINDOW_BYTES, 0)


@pytest.mark.parametrize("k,w", [(21, 11), (5, 3), (31, 1), (16, 20)])
def test_windowed_minimizers_reduce(ctx, k, w):
    buf = O.synth_reads(0x5EED0005, 3, 300, 150, 6).tobytes()
    t = to_dev(buf)
    ctx.accum_reset()
    ctx.minimizers_reduce_device(t, len(buf), k, w, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    assert_stats_equal(ctx.accum_read(), O.minimizers_reduce(buf, k, w, True, True), (k, w))
    ctx.accum_reset()
    ctx.minimizers_reduce_device(t, len(buf), k, w, nt.PATH_BITS_CANONICAL, nt.PRE_NONE)
    assert_stats_equal(ctx.accum_read(), O.minimizers_reduce(buf, k, w, False, False), (k, w, "bits"))


@pytest.mark.parametrize("w", [1, 2, 15, 16, 17, 31, 32, 33, 64, 100, 255, 256])
def test_windowed_minimizers_any_window_and_tile_boundaries(ctx, w):
    """Window sizes around the 16- and 32-bit word edges of the LDS flag words and up to the API's maximum, over long
    unbroken contigs (windows crossing the 2048-position tiles of the window-min kernel), low-complexity stretches (ties:
    the leftmost minimum decides the flag) and lengths around the tile size."""
    rng = np.random.default_rng(500 + w)
    parts = []
    for L in (2047, 2048, 2049, 5000, w + 20, 12_345):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)].copy()
        seq[L // 3: L // 3 + min(L // 4, 400)] = ord("A") if L % 2 else ord("T")     # homopolymer run: ties everywhere
        per = np.frombuffer(b"ACGTTGCA", dtype=np.uint8)
        m = min(L // 5, 300)
        seq[L // 2: L // 2 + m] = np.resize(per, m)                                   # short period: repeated k-mers
        parts.append(seq.tobytes())
    buf = b"\n".join(parts) + b"\n"
    t = to_dev(buf)
    for k, (path, accept_u, tie_rc) in ((21, (nt.PATH_BYTES_CANONICAL, True, True)), (8, (nt.PATH_BITS_CANONICAL, False, False))):
        ctx.accum_reset()
        ctx.minimizers_reduce_device(t, len(buf), k, w, path, nt.PRE_NORMALIZE if accept_u else nt.PRE_NONE)
        got = ctx.accum_read()
        assert_stats_equal(got, O.minimizers_reduce(buf, k, w, accept_u, tie_rc), (k, w))
        assert got["n_total"] > 0


# ---- materialise mode ---------------------------------------------------------------------------------

@pytest.mark.parametrize("k,path", [(5, nt.PATH_BITS), (16, nt.PATH_BITS_CANONICAL), (21, nt.PATH_BITS_CANONICAL), (32, nt.PATH_BITS_CANONICAL)])
def test_materialize_dense(ctx, k, path):
    buf = O.synth_reads(0x5EED0003, 7, 400, 150, 8).tobytes()
    n = len(buf)
    t = to_dev(buf)
    nt_ = (n + 1023) // 1024 * 1024
    vals = torch.zeros(nt_, dtype=torch.int64, device="cuda")
    v16 = torch.zeros(nt_ // 16, dtype=torch.int16, device="cuda")
    r16 = torch.zeros(nt_ // 16, dtype=torch.int16, device="cuda")
    ctx.materialize_device(t, n, k, path, nt.PRE_NONE, vals, v16, r16)
    ctx.synchronize()
    vals = vals.cpu().numpy().view(np.uint64)
    v16 = v16.cpu().numpy().view(np.uint16)
    r16 = r16.cpu().numpy().view(np.uint16)
    e = np.arange(n)
    valid = (v16[e // 16] >> (15 - e % 16)) & 1
    rcb = (r16[e // 16] >> (15 - e % 16)) & 1
    # oracle per record
    recs = buf.split(b"\n")[:-1]
    start = 0
    for r in recs:
        pos, val, flg = O.bit_kmers_arrays(r, k, path == nt.PATH_BITS_CANONICAL)
        ends = start + pos.astype(np.int64) + (k - 1)
        got_ends = e[start : start + len(r) + 1][valid[start : start + len(r) + 1] == 1]
        assert np.array_equal(got_ends, ends)
        assert np.array_equal(vals[ends], val)
        assert np.array_equal(rcb[ends].astype(np.uint8), flg)
        start += len(r) + 1


def oracle_whole_batch(t, n_records, L, k, path, pre, seed, chunk_records=None):
    """The oracle's reduced result of a whole device-resident synthetic batch.  The bytes come back from the device in
    chunks (test_synth_reads_device_matches_oracle pins the generator against the oracle's; the first chunk is re-checked
    here) and go through the literal per-record chain on every host thread."""
    stride = L + 1
    chunk_records = chunk_records or max(1, (1 << 30) // stride)
    threads = os.cpu_count() or 1
    total = None
    for first in range(0, n_records, chunk_records):
        n = min(chunk_records, n_records - first)
        host = t[first * stride:(first + n) * stride].cpu().numpy()
        if first == 0:
            m = min(n, 2000)
            assert np.array_equal(host[: m * stride], O.synth_reads(seed, 0, m, L, 1))
        offs = np.arange(n + 1, dtype=np.uint64) * stride
        part = O.reduce_batch(host, offs, 1, k, path, pre, threads)
        if total is None:
            total = part
        else:
            for key in ("n_total", "n_fwd", "n_rc"):
                total[key] += part[key]
            total["sum"] = (total["sum"] + part["sum"]) & (2 ** 64 - 1)
            total["xor"] ^= part["xor"]
            total["hist"] = total["hist"] + part["hist"]
    return total


def test_full_size_properties_config3(ctx):
    """BASELINE.json configs[2]: 1M x 10 kb contigs, k=31 bit path (strip_returns -> bit_kmers(31, true)): the whole batch
    bit-exactly against the oracle; plus linearity and reverse-complement invariance."""
    n_contigs, L, k = 1_000_000, 10_000, 31
    stride = L + 1
    nbytes = n_contigs * stride
    t = torch.empty(nbytes + 1024, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0003, 0, n_contigs, L, 1, t)
    path, pre = nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS
    ctx.accum_reset(); ctx.reduce_device(t, nbytes, k, path, pre); whole = ctx.accum_read()
    assert whole["n_total"] == whole["n_fwd"] + whole["n_rc"] == int(whole["hist"].sum())
    assert 0 < whole["n_total"] <= n_contigs * (L - k + 1)
    # the ENTIRE batch against the oracle's literal per-record chain (all host threads), bit-exact: 5 scalars + 4096 bins
    assert_stats_equal(whole, oracle_whole_batch(t, n_contigs, L, k, O.PATH_BITS_CANONICAL, O.PRE_STRIP_RETURNS, 0x5EED0003), "whole batch")
    a, b = 16 * 7_001, 16 * 40_003
    ctx.accum_reset()
    for lo, hi in ((0, a), (a, b), (b, n_contigs)):
        ctx.reduce_device(t.data_ptr() + lo * stride, (hi - lo) * stride, k, path, pre)
    assert_stats_equal(ctx.accum_read(), whole, "linearity")
    t2 = torch.empty_like(t)
    ctx.reverse_complement_records_device(t, t2, n_contigs, L, stride)
    ctx.accum_reset(); ctx.reduce_device(t2, nbytes, k, path, pre); rcst = ctx.accum_read()
    assert rcst["n_total"] == whole["n_total"] and rcst["n_fwd"] == whole["n_rc"] and rcst["sum"] == whole["sum"]
    assert rcst["xor"] == whole["xor"] and np.array_equal(rcst["hist"], whole["hist"])


def test_materialize_agrees_with_reduce_at_scale(ctx):
    """The two sinks must describe the same k-mer stream: summing / xoring / counting the dense materialised plane on the
    device (torch) reproduces the reduce-mode accumulators (2 M reads, k = 21 specialised builds and a generic k)."""
    n_reads, L = 2_000_000, 150
    nbytes = n_reads * (L + 1)
    t = torch.empty(nbytes + 1024, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 77, n_reads, L, 2, t)
    n16 = (nbytes + 15) // 16
    vals = torch.empty(n16 * 16, dtype=torch.int64, device="cuda")
    v16 = torch.empty(n16, dtype=torch.int16, device="cuda")
    r16 = torch.empty(n16, dtype=torch.int16, device="cuda")
    shifts = (15 - torch.arange(16, device="cuda", dtype=torch.int32)).view(1, 16)
    for k, path, pre in ((21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), (31, nt.PATH_BITS_CANONICAL, nt.PRE_NONE),
                         (24, nt.PATH_BITS_CANONICAL, nt.PRE_NONE), (11, nt.PATH_BITS, nt.PRE_NONE)):
        ctx.accum_reset(); ctx.reduce_device(t, nbytes, k, path, pre); red = ctx.accum_read()
        ctx.materialize_device(t, nbytes, k, path, pre, vals, v16, r16)
        valid = ((v16.to(torch.int32).view(-1, 1) & 0xFFFF) >> shifts) & 1
        rcb = ((r16.to(torch.int32).view(-1, 1) & 0xFFFF) >> shifts) & 1
        m = valid.view(-1).bool()
        sel = vals[m]
        assert int(m.sum()) == red["n_total"] and int((rcb.view(-1).bool() & m).sum()) == red["n_rc"], k
        assert int(sel.sum()) & 0xFFFFFFFFFFFFFFFF == red["sum"], k            # int64 wrap-around == sum mod 2^64
        x = sel
        while x.numel() > 1:                                                    # xor-reduce by halving
            if x.numel() % 2:
                x = torch.cat([x, x.new_zeros(1)])
            x = x[: x.numel() // 2] ^ x[x.numel() // 2:]
        assert (int(x[0]) & 0xFFFFFFFFFFFFFFFF) == red["xor"], k
        del sel, x, m, valid, rcb


# ---- synthetic generator and BASELINE-size properties ----------------------------------------------------

def test_device_synth_matches_cpu_generator(ctx):
    n_reads, L = 5000, 150
    t = torch.zeros(n_reads * (L + 1) + 64, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 123, n_reads, L, 1, t)
    ctx.synchronize()
    got = t[: n_reads * (L + 1)].cpu().numpy()
    assert np.array_equal(got, O.synth_reads(0x5EED0002, 123, n_reads, L, 1))


def test_full_size_properties_config2(ctx):
    """BASELINE.json configs[1]: 10M x 150 bp, k=21 canonical:
    (1) the WHOLE batch equals the oracle exactly; (2) linearity: whole == sum of record-aligned parts;
    (3) reverse-complementing every read leaves histogram / sum / xor unchanged and swaps n_fwd <-> n_rc (k odd)."""
    n_reads, L, k = 10_000_000, 150, 21
    stride = L + 1
    nbytes = n_reads * stride
    t = torch.empty(nbytes + 1024, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 0, n_reads, L, 1, t)
    path, pre = nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE
    ctx.accum_reset(); ctx.reduce_device(t, nbytes, k, path, pre); whole = ctx.accum_read()
    assert whole["n_total"] == whole["n_fwd"] + whole["n_rc"] == int(whole["hist"].sum())
    assert 0 < whole["n_total"] <= n_reads * (L - k + 1)
    # (1) the ENTIRE batch against the oracle's literal per-record chain (all host threads), bit-exact: 5 scalars + 4096 bins
    assert_stats_equal(whole, oracle_whole_batch(t, n_reads, L, k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, 0x5EED0002), "whole batch")
    # (2) linearity over three unequal record-aligned parts (16-B aligned cuts: 16 | 151*16)
    a = 16 * 100_003  # reads; a*stride is a multiple of 16
    b = 16 * 400_001
    ctx.accum_reset()
    for lo, hi in ((0, a), (a, b), (b, n_reads)):
        ctx.reduce_device(t.data_ptr() + lo * stride, (hi - lo) * stride, k, path, pre)
    assert_stats_equal(ctx.accum_read(), whole, "linearity")
    # (3) reverse-complement invariance
    t2 = torch.empty_like(t)
    ctx.reverse_complement_records_device(t, t2, n_reads, L, stride)
    ctx.accum_reset(); ctx.reduce_device(t2, nbytes, k, path, pre); rcst = ctx.accum_read()
    assert rcst["n_total"] == whole["n_total"] and rcst["n_fwd"] == whole["n_rc"] and rcst["n_rc"] == whole["n_fwd"]
    assert rcst["sum"] == whole["sum"] and rcst["xor"] == whole["xor"] and np.array_equal(rcst["hist"], whole["hist"])


def test_full_size_properties_k_above_32(ctx):
    """CanonicalKmers with k = 51 / 127 on the whole config-2 batch (10 M x 150 bp) - sizes the literal iterator does not finish in seconds, so
    size-independent properties: (1) the packed-stream kernel and the byte-walking kernel (NTK_ROUTE_NO_SPECULATION) agree on counters and
    histogram, normalised input or not; (2) linearity over unequal record-aligned parts; (3) reverse-complementing every read swaps n_fwd and
    n_rc (k odd: no k-mer is its own reverse complement) and leaves the histogram alone; (4) n_total against the window count the host derives
    from the positions of the breaks (runs of bases of length r hold r - k + 1 windows)."""
    n_reads, L = 10_000_000, 150
    stride = L + 1
    nbytes = n_reads * stride
    t = torch.empty(nbytes + 1024, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 0, n_reads, L, 1, t)
    path = nt.PATH_BYTES_CANONICAL
    # runs of bases: every byte that is not ACGT ends one (the synthetic batch holds ACGT, N and the record separator)
    bases = (t[:nbytes] == 65) | (t[:nbytes] == 67) | (t[:nbytes] == 71) | (t[:nbytes] == 84)
    brk = torch.nonzero(~bases).flatten()
    runs = torch.diff(brk, prepend=torch.tensor([-1], device="cuda")) - 1   # the batch ends with a separator: every run is closed
    try:
        for k in (51, 127):
            want_total = int(torch.clamp(runs - k + 1, min=0).sum())
            got = {}
            for pre in (nt.PRE_NORMALIZE, nt.PRE_NONE):
                for route in (0, NL.ROUTE_NO_SPECULATION):
                    ctx.set_option(NL.OPT_MINIMIZER_ROUTE, route)
                    ctx.reduce_device(t, nbytes, k, path, pre, reset=True)
                    got[(pre, route)] = ctx.accum_read()
            ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0)
            with redone_launches(ctx) as rl:   # the synthetic batch (upper case, random) leaves the packed-stream kernel's result standing
                ctx.reduce_device(t, nbytes, k, path, nt.PRE_NORMALIZE, reset=True)
                ctx.reduce_device(t, nbytes, k, path, nt.PRE_NONE)
                ctx.reduce_device(t, nbytes, 21, path, nt.PRE_NONE)
                assert rl.count() == 0, k
            whole = got[(nt.PRE_NORMALIZE, 0)]
            assert whole["n_total"] == want_total == whole["n_undigested"] and whole["n_total"] == whole["n_fwd"] + whole["n_rc"] == int(whole["hist"].sum())
            assert whole["sum"] == 0 and whole["xor"] == 0
            for key, st in got.items():   # (1)
                assert_stats_equal(st, whole, ("k > 32, routes", k, key))
            a, b = 16 * 100_003, 16 * 400_001   # (2): a * stride is a multiple of 16
            ctx.accum_reset()
            for lo, hi in ((0, a), (a, b), (b, n_reads)):
                ctx.reduce_device(t.data_ptr() + lo * stride, (hi - lo) * stride, k, path, nt.PRE_NORMALIZE)
            assert_stats_equal(ctx.accum_read(), whole, ("k > 32, linearity", k))
            t2 = torch.empty_like(t)   # (3)
            ctx.reverse_complement_records_device(t, t2, n_reads, L, stride)
            ctx.reduce_device(t2, nbytes, k, path, nt.PRE_NORMALIZE, reset=True)
            rcst = ctx.accum_read()
            assert rcst["n_total"] == whole["n_total"] and rcst["n_fwd"] == whole["n_rc"] and rcst["n_rc"] == whole["n_fwd"]
            assert np.array_equal(rcst["hist"], whole["hist"])
            del t2
    finally:
        ctx.set_option(NL.OPT_MINIMIZER_ROUTE, 0)


# ---- property-based differential test on the GPU (hypothesis) ---------------------------------------------------------
from hypothesis import given, settings, strategies as st_  # noqa: E402

_ALPHABET = b"ACGT" * 6 + b"acgt" + b"NnUuRYKM-.* \t\r\n\x00\x7f\x80\xff0@>"
_HCTX = []


@settings(max_examples=60, deadline=None)
@given(data=st_.lists(st_.sampled_from(list(_ALPHABET)), min_size=0, max_size=5000).map(bytes), k=st_.integers(1, 32),
       mode=st_.sampled_from(MODES))
def test_reduce_matches_oracle_property(data, k, mode):
    if not _HCTX:
        _HCTX.append(nt.Context(0, stream=torch.cuda.current_stream().cuda_stream))
    path, pre, canon, tie, u = mode
    assert_stats_equal(gpu_reduce(_HCTX[0], data, k, path, pre), O.reduce_fused(data, k, canon, tie, u), (k, path, pre, len(data)))


_ALPHABET_W = b"ACGT" * 80 + b"acgt" + b"NnUu-. \t\r\n\x00\xff"   # mostly bases: windows of 33+ bases have to exist


@settings(max_examples=40, deadline=None)
@given(data=st_.lists(st_.sampled_from(list(_ALPHABET_W)), min_size=0, max_size=6000).map(bytes), k=st_.integers(33, 255), normalized=st_.booleans())
def test_wide_k_matches_oracle_property(data, k, normalized):
    """CanonicalKmers with k = 33..255 on the reduce face (the packed-stream kernel, or the byte-walking one behind its flag) against the
    literal iterator.  On the device face every byte that is not a base breaks the windows - the whitespace class too, which the packer would
    have deleted - so the reference sees the pieces between them."""
    import re
    if not _HCTX:
        _HCTX.append(nt.Context(0, stream=torch.cuda.current_stream().cuda_stream))
    recs = re.split(rb"[\n\r\t ]", data) if normalized else data.split(b"\n")
    want = _wide_reference(recs, k, normalized)
    ctx = _HCTX[0]
    ctx.reduce_device(to_dev(data), len(data), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE if normalized else nt.PRE_NONE, reset=True)
    got = ctx.accum_read()
    assert_stats_equal(got, want, (k, normalized, len(data)))
    assert got["n_undigested"] == got["n_total"]


# ---- lifecycle / concurrency smoke ----------------------------------------------------------------------------------------

def test_context_lifecycle_and_independent_contexts():
    """ctx create/destroy in a loop (no leaks that break later work), two independent ctxs interleaved on their own streams,
    a ctx on a non-default torch stream, and one pinned batch reused many times."""
    buf = O.synth_reads(0x5EED0009, 0, 500, 150, 3).tobytes()
    want = O.reduce_fused(buf, 21, True, True, True)
    t = to_dev(buf)
    torch.cuda.synchronize()
    for _ in range(20):
        with nt.Context(0) as c:
            c.accum_reset(); c.reduce_device(t, len(buf), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
            assert_stats_equal(c.accum_read(), want, "lifecycle")
    a, b = nt.Context(0), nt.Context(0)
    a.accum_reset(); b.accum_reset()
    for _ in range(5):
        a.reduce_device(t, len(buf), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
        b.reduce_device(t, len(buf), 31, nt.PATH_BITS_CANONICAL, nt.PRE_NONE)
    ra, rb = a.accum_read(), b.accum_read()
    w31 = O.reduce_fused(buf, 31, True, False, False)
    assert ra["n_total"] == 5 * want["n_total"] and ra["sum"] == (5 * want["sum"]) & 0xFFFFFFFFFFFFFFFF and ra["xor"] == want["xor"]
    assert rb["n_total"] == 5 * w31["n_total"] and rb["xor"] == w31["xor"]
    a.close(); b.close()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with nt.Context(0, stream=s.cuda_stream) as c:
            c.accum_reset(); c.reduce_device(t, len(buf), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
            assert_stats_equal(c.accum_read(), want, "side stream")
    recs = buf.split(b"\n")[:-1]
    with nt.Context(0) as c:
        c.accum_reset()
        bt = c.batch(1 << 16, 1024)
        for rep in range(50):
            for r in recs[:100]:
                assert bt.append(r, nt.PRE_NORMALIZE)
            bt.submit(21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
            bt.wait()
        bt.release()
        st = c.accum_read()
        w100 = O.reduce_fused(b"".join(r + b"\n" for r in recs[:100]), 21, True, True, True)
        assert st["n_total"] == 50 * w100["n_total"] and st["sum"] == (50 * w100["sum"]) & 0xFFFFFFFFFFFFFFFF


def _specimen_files():
    import tomli
    spec = os.path.join(os.path.dirname(__file__), "golden", "specimen")
    out = []
    for kind, skip in (("FASTA", set()), ("FASTQ", {"wrapping_original_sanger.fastq", "longreads_original_sanger.fastq", "tricky.fastq"})):
        with open(os.path.join(spec, kind, "index.toml"), "rb") as f:
            idx = tomli.load(f)
        for t in idx["valid"]:
            if "comments" in (t.get("tags") or []) or t["filename"] in skip:
                continue
            out.append(os.path.join(spec, kind, t["filename"]))
    return out


def test_specimen_corpus_through_the_pipeline(ctx):
    """Every valid file of the reference's conformance corpus (reference tests/format_specimens.rs; protein, RNA, IUPAC,
    gapped, soft-masked, DOS line ends, zero-length records ...) through parser -> pinned batches -> scan, against the
    literal per-record oracle chain on the parser's raw sequences, for the README mode and both bit modes."""
    files = _specimen_files()
    assert len(files) > 80
    n_kmers = 0
    for path in files:
        recs = [r.raw_seq for r in nt.parse_fastx_file(path)]
        for k, p, pre in ((4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), (21, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS),
                          (7, nt.PATH_BITS, nt.PRE_NORMALIZE_IUPAC)):
            st = nt.scan_file(ctx, path, k, p, pre, batch_bytes=1 << 16)
            assert st["n_records"] == len(recs), path
            assert_stats_equal(st, O.reduce_records(recs, k, p, pre), f"{os.path.basename(path)} k={k}")
            n_kmers += st["n_total"]
    assert n_kmers > 10_000


def test_specimen_corpus_gzipped_through_the_streamed_route(ctx, tmp_path):
    """Round 6: the same corpus, every valid file gzip-compressed (one member; every third file as two members, every fifth as block gzip),
    through ntk_scan_file_parallel's streamed route (inflater + parser threads side by side) - multi-line FASTA, DOS line ends, zero-length
    records, files of a few bytes - against the literal per-record chain; then the shapes that must be errors: an empty member (EmptyFile,
    reference src/parser/mod.rs:88-91), text that is neither FASTA nor FASTQ, a truncated record at the end of the text, garbage after the
    last member."""
    import gzip
    import zlib
    files = _specimen_files()
    n_kmers = 0
    for i, path in enumerate(files):
        data = open(path, "rb").read()
        recs = [r.raw_seq for r in nt.parse_fastx_file(path)]
        if i % 5 == 4:
            try:
                import ctypes
                ctypes.CDLL("libdeflate.so.0")
                z = bgzf_compress(data, block=997)
            except OSError:
                z = gzip.compress(data)
        elif i % 3 == 2 and len(data) > 40:
            cut = data.rfind(b"\n", 0, len(data) // 2) + 1 or len(data) // 2     # (members need not end at record boundaries)
            z = gzip.compress(data[:cut], 6) + gzip.compress(data[cut:], 1)
        else:
            z = gzip.compress(data, 9 if i % 2 else 1)
        gz = tmp_path / "s.gz"
        gz.write_bytes(z)
        for k, p_, pre, threads in ((4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, 3), (21, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS, 2)):
            if len(data) < 2:
                continue
            st = nt.scan_file_parallel(ctx, str(gz), k, p_, pre, threads=threads, batch_bytes=1 << 14, streaming_fallback=False)
            assert st["gzip"]["streamed"] == 1 and st["gzip"]["text_bytes"] == len(data), path
            assert st["n_records"] == len(recs), path
            assert_stats_equal(st, O.reduce_records(recs, k, p_, pre), f"{os.path.basename(path)} k={k} gz")
            n_kmers += st["n_total"]
    assert n_kmers > 10_000
    good = b"@r1\nACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIII\n"
    for name, z in (("empty member", gzip.compress(b"")), ("one byte", gzip.compress(b"A")), ("not FASTA / FASTQ", gzip.compress(b"hello world\nACGT\n" * 50)),
                    ("truncated last record", gzip.compress(good * 40 + b"@r2\nACGT\n+\n")),
                    ("garbage after the last member", gzip.compress(good * 40) + b"\x00\x01garbage garbage garbage")):
        bad = tmp_path / "bad.gz"
        bad.write_bytes(z)
        with pytest.raises(nt.NtkError) as e:
            nt.scan_file_parallel(ctx, str(bad), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=3, batch_bytes=1 << 14, streaming_fallback=False)
        assert e.value.status == 8, name
    # trailing zero padding after the last member is tolerated (as zlib-based readers do), and the ctx works after the errors
    ok = tmp_path / "padded.gz"
    ok.write_bytes(gzip.compress(good * 40) + b"\0" * 512)
    st = nt.scan_file_parallel(ctx, str(ok), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=3, batch_bytes=1 << 14, streaming_fallback=False)
    assert st["n_records"] == 40 and st["n_total"] == 40 * 6


# ---- quality masking fused into the scan (reference src/sequence.rs:285-296) -----------------------

def _qual_dev(qual: bytes):
    n = len(qual)
    t = torch.zeros(((n + 1023) // 1024 * 1024 + 1024,), dtype=torch.uint8, device="cuda")  # quality 0 in the padding
    if n:
        t[:n] = torch.frombuffer(bytearray(qual), dtype=torch.uint8).cuda()
    return t


@pytest.mark.parametrize("k", [1, 4, 11, 16, 17, 21, 27, 31, 32])
def test_quality_masked_reduce_device(ctx, k):
    """`(seq, qual).quality_mask(cutoff)` then the chain, in one pass: against the oracle's quality_mask followed by its
    reduce, for every mode, realistic Phred+33 qualities and arbitrary bytes/cutoffs."""
    rng = np.random.default_rng(1000 + k)
    n = 200_000
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)]
    other = np.frombuffer(b"NnacgtUu-\n", dtype=np.uint8)[rng.integers(0, 10, size=n)]
    buf = np.where(rng.random(n) < 0.01, other, base).astype(np.uint8).tobytes()
    for trial, (path, pre, canon, tie_rc, accept_u) in enumerate(MODES):
        if trial % 2 == 0:
            qual = rng.integers(33, 75, size=n, dtype=np.uint8).tobytes()
            cutoff = [36, 53, 74, 34][trial % 4]
        else:
            qual = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            cutoff = [3, 128, 129, 250][(trial + k) % 4]
        masked = O.quality_mask(buf, qual, cutoff)
        want = O.reduce_fused(masked, k, canon, tie_rc, accept_u)
        ctx.accum_reset()
        ctx.reduce_device(to_dev(buf), n, k, path, pre, d_qual=_qual_dev(qual), quality_cutoff=cutoff)
        assert_stats_equal(ctx.accum_read(), want, f"quality k={k} mode={trial} cutoff={cutoff}")
    # cutoff 0 and "no quality stream" are the plain scan
    want = O.reduce_fused(buf, k, True, True, True)
    ctx.accum_reset()
    ctx.reduce_device(to_dev(buf), n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, d_qual=_qual_dev(qual), quality_cutoff=0)
    assert_stats_equal(ctx.accum_read(), want, "cutoff 0")


def test_quality_masked_materialize_and_minimizers(ctx):
    rng = np.random.default_rng(4242)
    n, k, w, cutoff = 50_000, 21, 11, 40
    buf = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()
    qual = rng.integers(33, 75, size=n, dtype=np.uint8).tobytes()
    masked = O.quality_mask(buf, qual, cutoff)
    # minimizers over the masked reads
    want = O.minimizers_reduce(masked, k, w, True, True)
    ctx.accum_reset()
    ctx.reduce_device(to_dev(buf), n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, d_qual=_qual_dev(qual), quality_cutoff=cutoff)
    assert_stats_equal(ctx.accum_read(), want, "minimizers + quality")
    # materialise: identical planes to materialising the masked buffer without a quality stream
    nt16 = (n + 15) // 16 * 16
    outs = []
    for b, q, c in ((buf, _qual_dev(qual), cutoff), (masked, None, 0)):
        vals = torch.zeros(nt16 + 1024, dtype=torch.int64, device="cuda")
        v16 = torch.zeros(nt16 // 16 + 64, dtype=torch.int16, device="cuda")
        r16 = torch.zeros(nt16 // 16 + 64, dtype=torch.int16, device="cuda")
        ctx.materialize_device(to_dev(b), n, k, nt.PATH_BITS_CANONICAL, nt.PRE_NONE, vals, v16, r16, d_qual=q, quality_cutoff=c)
        torch.cuda.synchronize()
        valid = v16[: nt16 // 16].cpu().numpy().view(np.uint16)
        bits = np.unpackbits(valid.byteswap().view(np.uint8))[:n].astype(bool)
        outs.append((valid.copy(), r16[: nt16 // 16].cpu().numpy().copy(), vals[:n].cpu().numpy()[bits]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert outs[0][2].size > 1000


def test_quality_masked_pipeline(ctx, golden_dir, tmp_path):
    """FASTQ file -> parser -> pinned batches carrying the quality lines -> masked scan, against the literal per-record chain
    quality_mask -> normalize -> canonical_kmers on the reference's own FASTQ sample; sequential and parallel producers."""
    fq = os.path.join(golden_dir, "PRJNA271013_head.fq")
    recs = [(r.raw_seq, r.qual.encode()) for r in nt.parse_fastx_file(fq)]
    assert len(recs) > 100
    for cutoff in (35, 53, 64):
        masked = [O.quality_mask(s, q, cutoff) for s, q in recs]
        want = O.reduce_records(masked, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
        plain = O.reduce_records([s for s, _ in recs], 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
        st = nt.scan_file(ctx, fq, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, batch_bytes=1 << 16, quality_cutoff=cutoff)
        assert_stats_equal(st, want, f"pipeline cutoff {cutoff}")
        assert st["n_total"] < plain["n_total"] or cutoff == 35
        stp = nt.scan_file_parallel(ctx, fq, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, threads=4, batch_bytes=1 << 16,
                                    quality_cutoff=cutoff)
        assert_stats_equal(stp, want, f"parallel pipeline cutoff {cutoff}")
    # a FASTA file has no qualities: a cutoff changes nothing
    fa = os.path.join(golden_dir, "28S.fasta")
    a = nt.scan_file(ctx, fa, 4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    b = nt.scan_file(ctx, fa, 4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, quality_cutoff=60)
    assert_stats_equal(a, b, "fasta + cutoff")
    # whitespace inside a sequence line with a low quality is an N by the time normalize runs (it is not deleted)
    odd = tmp_path / "odd.fq"
    odd.write_bytes(b"@r\nACGTAC GTACGTACGTACGTAC\tGTACGT\n+\nIIIIII!IIIIIIIIIIIIIIIIIIIIIII\n")
    seq, qual = b"ACGTAC GTACGTACGTACGTAC\tGTACGT", b"IIIIII!IIIIIIIIIIIIIIIIIIIIIII"
    for cutoff in (34, 80):
        want = O.reduce_records([O.quality_mask(seq, qual, cutoff)], 5, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
        st = nt.scan_file(ctx, str(odd), 5, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, quality_cutoff=cutoff)
        assert_stats_equal(st, want, f"whitespace + quality, cutoff {cutoff}")
    # a batch filled for one cutoff cannot be submitted with another
    b = ctx.batch(1 << 16, 16)
    assert b.append(b"ACGTACGTACGT", nt.PRE_NORMALIZE, qual=b"IIIIIIIIIIII", quality_cutoff=40)
    with pytest.raises(nt.NtkError):
        b.submit(4, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, quality_cutoff=41)
    b.release()


def test_quality_stream_at_any_address(ctx):
    """Both streams at device addresses whose low 32 bits have the top bit set / clear (a sign-extended buffer descriptor
    base would fault or read elsewhere), several chunks of tiles long."""
    rng = np.random.default_rng(99)
    n, k, cutoff = 3_000_000, 21, 36
    buf = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()
    qual = rng.integers(33, 75, size=n, dtype=np.uint8).tobytes()
    want = O.reduce_fused(O.quality_mask(buf, qual, cutoff), k, True, True, True)
    big = torch.zeros(5 << 30, dtype=torch.uint8, device="cuda")   # spans at least one 4 GiB boundary
    base = big.data_ptr()
    hb, hq = torch.frombuffer(bytearray(buf), dtype=torch.uint8), torch.frombuffer(bytearray(qual), dtype=torch.uint8)
    seen = set()
    for want_bit in (1, 0):
        off = 0
        while ((base + off) >> 31) & 1 != want_bit:
            off += 1 << 30
        s_off, q_off = off + 4096, off + 4096 + ((n + 4096 + 15) // 16 * 16)
        assert q_off + n + 2048 < big.numel() and (base + s_off) % 16 == 0 and (base + q_off) % 16 == 0
        seen.add(((base + q_off) >> 31) & 1)
        big[s_off: s_off + n] = hb.cuda()
        big[q_off: q_off + n] = hq.cuda()
        ctx.accum_reset()
        ctx.reduce_device(big[s_off:], n, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, d_qual=big[q_off:], quality_cutoff=cutoff)
        assert_stats_equal(ctx.accum_read(), want, f"address bit31={want_bit}")
    assert seen == {0, 1}
    del big
    torch.cuda.empty_cache()


def test_batch_packer_against_a_model(ctx):
    """ntk_batch_append / ntk_batch_append_quality (the line-wise fast paths and the per-byte path) against a Python model of
    the packer: every pre-step, records with and without bytes of the deleted class, with and without qualities."""
    rng = np.random.default_rng(31337)
    alphabet = np.frombuffer(b"ACGTacgtNn-. \t\r\n*U", dtype=np.uint8)
    for pre in (nt.PRE_NONE, nt.PRE_STRIP_RETURNS, nt.PRE_NORMALIZE, nt.PRE_NORMALIZE_IUPAC):
        deleted = {nt.PRE_NONE: b"", nt.PRE_STRIP_RETURNS: b"\r\n"}.get(pre, b" \t\r\n")
        for cutoff in (0, 40):
            b = ctx.batch(1 << 20, 4096)
            want_seq, want_off = bytearray(), [0]
            for r in range(300):
                L = int(rng.integers(0, 400))
                p_ws = [0.0, 0.02, 0.3][r % 3]
                seq = np.where(rng.random(L) < p_ws, alphabet[rng.integers(12, 16, size=L)], alphabet[rng.integers(0, 12, size=L)]).astype(np.uint8).tobytes()
                qual = rng.integers(33, 75, size=L, dtype=np.uint8).tobytes() if (cutoff and r % 4) else None
                assert b.append(seq, pre, qual=qual, quality_cutoff=cutoff)
                for i, ch in enumerate(seq):
                    if ch in deleted and not (qual is not None and qual[i] < cutoff):
                        continue
                    want_seq.append(ch)
                want_seq.append(ord("\n"))
                want_off.append(len(want_seq))
            got_seq, got_off = b.buffers()
            assert bytes(got_seq) == bytes(want_seq), (pre, cutoff)
            assert list(got_off) == want_off
            b.release()


def test_bench_two_ranks_on_one_gpu_match_single_rank():
    """bench.py's N > 1 path end to end (records sharded by rank, the overlapped all-reduce of the accumulator words, the
    xor digest rebuilt from its summable form) with two ranks sharing cuda:0 over gloo, against one rank scanning the same
    2 x reads: identical reduced results.  (RCCL itself needs two GPUs; the driver's scaling run covers that.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, READS="2500000")
    r = subprocess.run(["bash", os.path.join(root, "tools", "n2_on_one_gpu.sh")], cwd=root, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "n2_on_one_gpu ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_minimizers_in_chunks():
    """Long inputs are scanned in chunks with w+k-2 bytes of left context (bounded scratch); with a 4 KiB chunk every
    boundary case shows up in a small buffer: records and windows straddling chunk edges, all window sizes."""
    rng = np.random.default_rng(2718)
    parts = []
    for L in rng.integers(1, 3000, size=60):
        parts.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(L))].tobytes())
    buf = b"\n".join(parts) + b"\n"
    with nt.Context(0, stream=torch.cuda.current_stream().cuda_stream) as c:
        c.set_option(NL.OPT_MINIMIZER_CHUNK_BYTES, 4096)
        t = to_dev(buf)
        for k, w in ((21, 11), (31, 2), (17, 16), (5, 64), (12, 256), (32, 1)):
            for path, accept_u, tie_rc in ((nt.PATH_BYTES_CANONICAL, True, True), (nt.PATH_BITS_CANONICAL, False, False)):
                want = O.minimizers_reduce(buf, k, w, accept_u, tie_rc)
                for two_pass in (False, True):   # the default route (a fused kernel where one serves the pair), then the chunked two-pass path
                    c.set_option(NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_TWO_PASS if two_pass else 0)
                    c.accum_reset()
                    c.minimizers_reduce_device(t, len(buf), k, w, path, nt.PRE_NORMALIZE if accept_u else nt.PRE_NONE)
                    assert_stats_equal(c.accum_read(), want, (k, w, "chunked two-pass" if two_pass else "default route"))


def test_compressed_inputs_through_the_pipeline(ctx, golden_dir, tmp_path):
    """bzip2 / xz / zstd / gzip inputs (reference tests/test_compressed.rs data files and larger streams written here) through
    parser -> pinned batches -> scan give the plain file's result."""
    import bz2
    import lzma
    plain = nt.scan_file(ctx, os.path.join(golden_dir, "test.fa"), 3, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
    for ext in ("gz", "bz2", "xz", "zst"):
        st = nt.scan_file(ctx, os.path.join(golden_dir, "test.fa." + ext), 3, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
        assert_stats_equal(st, plain, ext)
        assert st["n_records"] == 2 and st["n_total"] > 0
    fa = os.path.join(golden_dir, "28S.fasta")
    data = open(fa, "rb").read()
    want = nt.scan_file(ctx, fa, 21, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS)
    for name, blob in (("x.bz2", bz2.compress(data)), ("x.xz", lzma.compress(data))):
        p = tmp_path / name
        p.write_bytes(blob)
        assert_stats_equal(nt.scan_file(ctx, str(p), 21, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS, batch_bytes=1 << 16), want, name)
        # the parallel entry point cannot split these streams: it falls back to the streaming reader
        assert_stats_equal(nt.scan_file_parallel(ctx, str(p), 21, nt.PATH_BITS_CANONICAL, nt.PRE_STRIP_RETURNS), want, name + " parallel")


def test_scan_larger_than_one_launch(ctx):
    """A 35 GB resident batch: beyond 2^25 tiles (33 GB) the scan is split into several launches with 32-bit launch-relative
    tile indices.  The result over the whole buffer must equal the sum of the results over two single-launch halves cut at a
    record boundary, and a prefix sample must match the oracle."""
    free, _ = torch.cuda.mem_get_info()
    reads, L = 232_000_000, 150
    n = reads * (L + 1)
    if free < n + (4 << 30):
        pytest.skip("not enough free device memory for the 35 GB case")
    assert n > (8 << 22) * 992
    seq = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0004, 0, reads, L, 1, seq)
    k, path, pre = 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE
    ctx.accum_reset(); ctx.reduce_device(seq, n, k, path, pre); whole = ctx.accum_read()
    half = (reads // 2) * (L + 1)                      # a record boundary, a multiple of 16 bytes? make it one
    half -= half % (16 * (L + 1))
    ctx.accum_reset(); ctx.reduce_device(seq, half, k, path, pre); a = ctx.accum_read()
    ctx.accum_reset(); ctx.reduce_device(seq[half:], n - half, k, path, pre); b = ctx.accum_read()
    for key in ("n_total", "n_fwd", "n_rc"):
        assert whole[key] == a[key] + b[key], key
    assert whole["sum"] == (a["sum"] + b["sum"]) % (1 << 64) and whole["xor"] == a["xor"] ^ b["xor"]
    assert np.array_equal(whole["hist"], a["hist"] + b["hist"])
    assert whole["n_total"] > reads * 100
    sample = 20_000
    ctx.accum_reset(); ctx.reduce_device(seq, sample * (L + 1), k, path, pre)
    assert_stats_equal(ctx.accum_read(), O.reduce_fused(O.synth_reads(0x5EED0004, 0, sample, L, 1), k, True, True, True), "prefix")
    # windowed minimizers over 6 GB (24 chunks of 256 MiB with left context, byte offsets beyond 2^32): linear over a cut at
    # a record boundary too, and equal to the oracle on the prefix sample
    n6 = 40_000_000 * (L + 1)
    h6 = 17_000_000 * (L + 1)
    h6 -= h6 % (16 * (L + 1))
    ctx.accum_reset(); ctx.reduce_device(seq, n6, k, path, pre, w=11); mw = ctx.accum_read()
    ctx.accum_reset(); ctx.reduce_device(seq, h6, k, path, pre, w=11); ma = ctx.accum_read()
    ctx.accum_reset(); ctx.reduce_device(seq[h6:], n6 - h6, k, path, pre, w=11); mb = ctx.accum_read()
    for key in ("n_total", "n_fwd", "n_rc"):
        assert mw[key] == ma[key] + mb[key], ("minimizers", key)
    assert mw["sum"] == (ma["sum"] + mb["sum"]) % (1 << 64) and mw["xor"] == ma["xor"] ^ mb["xor"]
    assert np.array_equal(mw["hist"], ma["hist"] + mb["hist"]) and mw["n_total"] > 40_000_000 * 100
    ctx.accum_reset(); ctx.reduce_device(seq, sample * (L + 1), k, path, pre, w=11)
    assert_stats_equal(ctx.accum_read(), O.minimizers_reduce(O.synth_reads(0x5EED0004, 0, sample, L, 1), k, 11, True, True), "minimizer prefix")
    del seq
    torch.cuda.empty_cache()


def test_rccl_allreduce_through_the_c_abi_single_rank(ctx):
    """The RCCL path of the C ABI (ntk_comm_* / ntk_allreduce_accumulators: ncclAllReduce(ncclUint64, ncclSum) on the scan
    stream + the xor rebuild) with a one-rank communicator: the sum over one rank is the rank's own result."""
    from needletail_amd import distributed as D
    buf = O.synth_reads(0x5EED0004, 0, 5000, 150, 2).tobytes()
    want = O.reduce_fused(buf, 21, True, True, True)
    t = to_dev(buf)
    for make in (lambda: D.Communicator.for_rank(ctx, 1, 0, D.Communicator.unique_id()), lambda: D.Communicator.all_local([ctx])):
        with make() as comm:
            assert comm.size == 1
            for _ in range(2):
                ctx.accum_reset()
                ctx.reduce_device(t, len(buf), 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
                comm.allreduce_accumulators()
                assert_stats_equal(ctx.accum_read(), want, "rccl single rank")
    with pytest.raises(nt.NtkError):
        D.Communicator.for_rank(ctx, 2, 5, b"\0" * 128)   # rank out of range: an argument error, not a hang


def test_rccl_all_visible_devices():
    """configs[3] over EVERY device this process can see (1 on the builder's box, 8 on a node - the test scales with
    torch.cuda.device_count()): ONE process, one ctx and one host thread per device, one ntk_comm_init_all communicator; each
    device generates and scans its round-robin share (record batches of 2^20 reads) of a 4 M-read
    seed-0x5EED0004 set, ONE ntk_allreduce_accumulators, and every device must then hold the oracle's result of the WHOLE set."""
    import threading
    from needletail_amd import _lib as NL
    from needletail_amd import distributed as D
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1 and NL.device_count() == n_dev
    total_reads, L, k = 4_000_000, 150, 21
    stride = L + 1
    want = O.reduce_fused_parallel(O.synth_reads(0x5EED0004, 0, total_reads, L, 1), stride, k, True, True, True,
                                   max(1, os.cpu_count() or 1))
    ctxs = [nt.Context(d) for d in range(n_dev)]      # own streams: nothing here goes through torch's current stream
    shards, seqs, errors = [], [], []
    try:
        for d in range(n_dev):
            batches = D.round_robin_batches(total_reads, d, n_dev)
            shards.append(batches)
            seqs.append(torch.empty(sum(n for _, n in batches) * stride + 2048, dtype=torch.uint8, device=f"cuda:{d}"))
        assert sum(n for b in shards for _, n in b) == total_reads
        torch.cuda.synchronize()

        def work(d):
            try:
                c, pos = ctxs[d], 0
                for first, n in shards[d]:
                    c.synth_reads_device(0x5EED0004, first, n, L, 1, seqs[d][pos:])
                    pos += n * stride
                c.accum_reset()
                c.reduce_device(seqs[d], pos, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE)
                c.synchronize()
            except Exception as e:   # noqa: BLE001 - reported below, from the main thread
                errors.append((d, repr(e)))

        threads = [threading.Thread(target=work, args=(d,)) for d in range(n_dev)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors and not any(t.is_alive() for t in threads), errors
        mine = [c.accum_read() for c in ctxs]
        assert sum(m["n_total"] for m in mine) == want["n_total"]
        with D.Communicator.all_local(ctxs) as comm:
            assert comm.size == n_dev
            comm.allreduce_accumulators()
            for d, c in enumerate(ctxs):
                assert_stats_equal(c.accum_read(), want, f"device {d} of {n_dev} after the all-reduce")
    finally:
        for c in ctxs:
            c.close()
        del seqs
        torch.cuda.empty_cache()


def test_bench_self_launch_all_devices():
    """`python bench.py --gpus <all visible devices>` end to end as the driver runs it: with more than one device it
    re-launches itself under torch.distributed.run (one process per GPU, the library's own RCCL communicator); with one
    device the same configs[3] path runs as a one-rank RCCL job.  Either way: the whole read set verified against the oracle,
    the library's communicator in use (rccl_ranks == devices, no fallback note)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_dev = torch.cuda.device_count()
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_dev), "--reads", "8000000", "--steps", "3", "--warmup", "1",
           "--preheat-ms", "20", "--no-cpu-baseline", "--no-secondary", "--init-timeout-s", "120"]
    if n_dev == 1:   # one rank: the RCCL path needs the rank environment (--gpus 1 alone is the single-GPU configs[1] run)
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        cmd += ["--workload", "c4"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert line["n_gpus"] == n_dev and cfg["rccl_ranks"] == n_dev and cfg["devices_visible"] == n_dev
    assert "collective" not in cfg and "test_mode" not in cfg, cfg
    assert cfg["reads_total"] == 8_000_000 and cfg["seed"] == "0x5eed0004"
    assert line["result"]["verified"] and "bit-exact" in line["result"]["verified"]
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


def test_bench_collective_fallback_is_opt_in():
    """A communicator that cannot come up is fatal unless --allow-collective-fallback is given (then the line says so)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", NTK_BENCH_FORCE_COLLECTIVE_FALLBACK="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "c4", "--reads", "1000000", "--steps", "2", "--warmup", "1",
           "--preheat-ms", "10", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "did not come up" in r.stderr and r.stdout.strip() == "", r.stderr[-2000:]
    r = subprocess.run(cmd + ["--allow-collective-fallback"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cfg = json.loads(r.stdout.strip().splitlines()[-1])["config"]
    assert cfg["rccl_ranks"] == 0 and cfg["collective"].startswith("FALLBACK")


@pytest.mark.parametrize("chunk_bytes", [None, 97, 4096])
def test_batched_compat_face_matches_the_iterators_per_record(ctx, chunk_bytes, restore_options):
    """ntk_bit_kmers_batch / ntk_canonical_kmers_batch: one call for a whole batch of records, element-wise against the
    oracle's literal iterators (reference src/sequence.rs:237-252) record by record; ragged, empty and all-N records,
    mixed case, k up to 255 on the byte path, and the capacity protocol.  The call pipelines chunks of the batch (two in
    flight, 16 MiB of packed bytes each); NTK_OPT_COMPAT_CHUNK_BYTES = 97 / 4096 forces hundreds of chunks out of this small batch:
    chunks of one record, records larger than a chunk, the capacity running out in the middle of a chunk."""
    ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, chunk_bytes or 0)   # (restored at the end of the test)
    rng = np.random.default_rng(21)
    alphabet = np.frombuffer(b"ACGTACGTACGTacgtNn-", dtype=np.uint8)
    records = [b"", b"A", b"N" * 40, b"ACGT" * 10, b"acgtACGTnACGTTGCA" * 3]
    for _ in range(400):
        records.append(bytes(alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 400)))]))
    records += [b"", bytes(alphabet[rng.integers(0, 4, 3000)])]
    for k, canonical in ((1, True), (4, False), (21, True), (31, True), (32, False)):
        counts, pos, val, flg = nt.bit_kmers_batch(records, k, canonical, ctx)
        assert len(counts) == len(records)
        o = 0
        for r, n in zip(records, counts.tolist()):
            p_, v_, f_ = O.bit_kmers_arrays(r, k, canonical)
            assert n == len(p_), (k, len(r))
            assert np.array_equal(pos[o:o + n], p_) and np.array_equal(val[o:o + n], v_) and np.array_equal(flg[o:o + n], f_)
            o += n
        assert o == len(pos)
    for k in (1, 4, 21, 33, 70, 255):
        counts, pos, flg = nt.canonical_kmers_batch(records, k, ctx)
        o = 0
        for r, n in zip(records, counts.tolist()):
            p_, f_ = O.canonical_kmers_arrays(r, O.reverse_complement(r), k)
            assert n == len(p_), (k, len(r))
            assert np.array_equal(pos[o:o + n], p_) and np.array_equal(flg[o:o + n], f_)
            o += n
        assert o == len(pos)
    # the same items as bit planes (ntk_canonical_kmers_batch_planes): the records are uploaded as they lie (no break bytes: windows
    # must not reach across a record start), a chunk begins on a word boundary of the planes, empty records share a position
    for k in (1, 2, 4, 21, 33, 70, 255):
        pl = nt.canonical_kmers_planes(records, k, ctx)
        tot = 0
        for i, r in enumerate(records):
            p_, f_ = O.canonical_kmers_arrays(r, O.reverse_complement(r), k)
            gp, gf = pl.arrays(i)
            assert np.array_equal(gp, p_) and np.array_equal(gf, f_), (k, i, len(r))
            tot += len(p_)
        assert pl.total == tot and int(pl.rec_bit[-1]) == 16 * len(pl.valid16)
        # nothing is set outside the records' own windows (padding bits, the last k - 1 starts of a record)
        allbits = int(np.unpackbits(pl.valid16.astype(">u2").view(np.uint8)).sum())
        assert allbits == tot and int(np.unpackbits((pl.rc16 & ~pl.valid16).astype(">u2").view(np.uint8)).sum()) == 0
    it = list(nt.canonical_kmers_planes(records[:8], 4, ctx).iter(4, records[4], O.reverse_complement(records[4])))
    assert it == O.canonical_kmers(records[4], O.reverse_complement(records[4]), 4)
    # offsets need not start at 0 (a slice of a reader's buffer), and an empty batch is fine
    import ctypes as C_
    from needletail_amd import _lib as L_
    flat = b"ACGTN" + b"".join(records[:40])
    offs2 = np.zeros(41, dtype=np.uint64); np.cumsum([len(r) for r in records[:40]], out=offs2[1:]); offs2 += 5
    capw = int(offs2[-1] - offs2[0]) // 16 + 41
    rb2 = np.zeros(41, dtype=np.uint64); v2 = np.zeros(capw, dtype=np.uint16); r2 = np.zeros(capw, dtype=np.uint16)
    nw2, tt2 = C_.c_uint64(0), C_.c_uint64(0)
    L_.check(L_.lib().ntk_canonical_kmers_batch_planes(ctx._h, flat, offs2.ctypes.data, 40, 21, rb2.ctypes.data, v2.ctypes.data, r2.ctypes.data, capw,
                                                      C_.byref(nw2), C_.byref(tt2)), "planes with offsets[0] = 5")
    ref = nt.canonical_kmers_planes(records[:40], 21, ctx)
    assert tt2.value == ref.total and np.array_equal(v2[: nw2.value], ref.valid16) and np.array_equal(r2[: nw2.value], ref.rc16) and np.array_equal(rb2, ref.rec_bit)
    e = nt.canonical_kmers_planes([], 21, ctx)
    assert e.total == 0 and len(e.valid16) == 0
    # capacity protocol: too small a buffer reports the needed count and fills what fits
    import ctypes as C
    from needletail_amd import _lib as L
    seq = b"".join(records)
    offs = np.zeros(len(records) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in records], out=offs[1:])
    rb = np.zeros(len(records) + 1, dtype=np.uint64); nw = C.c_uint64(0); tt = C.c_uint64(0)
    v16 = np.zeros(4, dtype=np.uint16); r16 = np.zeros(4, dtype=np.uint16)
    rc = L.lib().ntk_canonical_kmers_batch_planes(ctx._h, seq, offs.ctypes.data, len(records), 21, rb.ctypes.data, v16.ctypes.data, r16.ctypes.data,
                                                  4, C.byref(nw), C.byref(tt))
    assert rc == 5 and nw.value >= len(seq) // 16 and not v16.any()
    want_counts, want_pos, want_val, want_flg = nt.bit_kmers_batch(records, 21, True, ctx)
    cap = 100
    cnt = np.zeros(len(records), dtype=np.uint64); p2 = np.zeros(cap, dtype=np.uint64); v2 = np.zeros(cap, dtype=np.uint64)
    f2 = np.zeros(cap, dtype=np.uint8); tot = C.c_uint64(0)
    rc = L.lib().ntk_bit_kmers_batch(ctx._h, seq, offs.ctypes.data, len(records), 21, 1, cnt.ctypes.data, p2.ctypes.data,
                                     v2.ctypes.data, f2.ctypes.data, cap, C.byref(tot))
    assert rc == 5 and tot.value == len(want_pos) and np.array_equal(cnt, want_counts)
    assert np.array_equal(p2, want_pos[:cap]) and np.array_equal(v2, want_val[:cap]) and np.array_equal(f2, want_flg[:cap])
    # a larger batch: 20 000 reads of 150 bp in one call
    big = [bytes(r) for r in O.synth_reads(0x5EED0002, 0, 20000, 150, 4).reshape(20000, 151)[:, :150]]
    counts, pos, val, flg = nt.bit_kmers_batch(big, 21, True, ctx)
    assert int(counts.sum()) == len(pos)
    st = O.reduce_fused(b"".join(r + b"\n" for r in big), 21, True, False, False)
    assert len(pos) == st["n_total"] and int(flg.sum()) == st["n_rc"] and int(val.sum(dtype=np.uint64)) == st["sum"]
    for i in (0, 7, 19999):
        o = int(counts[:i].sum())
        p_, v_, f_ = O.bit_kmers_arrays(big[i], 21, True)
        assert np.array_equal(pos[o:o + len(p_)], p_) and np.array_equal(val[o:o + len(p_)], v_)


def test_batched_compat_face_with_page_locked_arrays(ctx):
    """ntk_pinned_alloc / ntk_pinned_free: the caller's arrays page-locked by the library (the copies of the batched compat face then
    run at the PCIe rate instead of through a bounce buffer) - same results as with pageable arrays, element-wise against the oracle."""
    import ctypes as C
    from needletail_amd import _lib as L
    lib = L.lib()
    assert lib.ntk_pinned_alloc(64, None) != 0                       # no out pointer: an argument error, not a crash
    lib.ntk_pinned_free(None)                                        # freeing nothing is allowed
    held = []

    def pinned(n_items, dtype):
        n_bytes = max(int(n_items) * np.dtype(dtype).itemsize, 8)
        ptr = C.c_void_p()
        L.check(lib.ntk_pinned_alloc(n_bytes, C.byref(ptr)), "ntk_pinned_alloc")
        assert ptr.value
        held.append(ptr)
        return np.frombuffer((C.c_uint8 * n_bytes).from_address(ptr.value), dtype=dtype)[:int(n_items)]

    reads = [bytes(r) for r in O.synth_reads(0x5EED0002, 3, 5000, 150, 16).reshape(5000, 151)[:, :150]] + [b"", b"ACGTN" * 7]
    want_counts, want_pos, want_flg = nt.canonical_kmers_batch(reads, 21, ctx)   # pageable arrays (checked against the oracle above)
    flat = pinned(sum(len(r) for r in reads), np.uint8); flat[:] = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = pinned(len(reads) + 1, np.uint64); offs[0] = 0; np.cumsum([len(r) for r in reads], out=offs[1:])
    cap = len(want_pos)
    counts, pos, flg = pinned(len(reads), np.uint64), pinned(cap, np.uint64), pinned(cap, np.uint8)
    tot = C.c_uint64(0)
    try:
        for _ in range(3):   # the banks of the pipeline are re-used from call to call
            counts[:] = 0; pos[:] = 0; flg[:] = 0
            L.check(lib.ntk_canonical_kmers_batch(ctx._h, C.cast(flat.ctypes.data, C.c_char_p), offs.ctypes.data, len(reads), 21, counts.ctypes.data,
                                                  pos.ctypes.data, flg.ctypes.data, cap, C.byref(tot)), "ntk_canonical_kmers_batch")
            assert tot.value == cap and np.array_equal(counts, want_counts) and np.array_equal(pos, want_pos) and np.array_equal(flg, want_flg)
        o = 0
        for r, n in list(zip(reads, counts.tolist()))[:50] + [(reads[-1], int(counts[-1]))]:
            p_, f_ = O.canonical_kmers_arrays(r, O.reverse_complement(r), 21)
            if r is reads[-1]:
                o = cap - n
            assert n == len(p_) and np.array_equal(pos[o:o + n], p_) and np.array_equal(flg[o:o + n], f_)
            o += n
    finally:
        del flat, offs, counts, pos, flg
        for ptr in held:
            lib.ntk_pinned_free(ptr)


def test_fused_minimizers_match_the_oracle_and_the_two_pass_path(ctx, restore_options):
    """configs[4] kernel side: the fused minimizer builds (one pass, no scratch planes) against the literal minimizer of every
    window (oracle) and against the two-pass path (materialise + window-min) on the same buffer, for every fused (k, w),
    both tie rules; a (k, w) without a fused build still works."""
    rng = np.random.default_rng(17)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNU\n", dtype=np.uint8)
    h = bytes(rng.choice(list(b"ACGT"), size=41).astype(np.uint8))
    buf = bytes(alphabet[rng.integers(0, len(alphabet), 60_000)]) + h + O.reverse_complement(h) + b"A" * 90 + b"T" * 90 + \
        O.synth_reads(0x5EED0002, 9, 3000, 150, 8).tobytes()
    t = to_dev(buf)
    # (22, 12), (23, 11), (23, 12): windows of 33 / 34 bytes - the fused builds with three halo lanes; w = 5: the short-window fused builds;
    # (24, 11), (21, 6): the generic kernel
    for k, w in ((21, 11), (17, 11), (18, 11), (19, 11), (20, 11), (22, 11), (21, 9), (21, 10), (21, 12), (23, 9), (23, 10), (23, 11), (23, 12), (22, 12), (24, 11), (21, 5), (15, 5), (19, 5), (23, 5), (21, 6)):
        for path, pre, tie, u in ((nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, True, True), (nt.PATH_BITS_CANONICAL, nt.PRE_NONE, False, False)):
            want = O.minimizers_reduce(buf, k, w, accept_u=u, tie_rc=tie)
            ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, path, pre, w=w)
            assert_stats_equal(ctx.accum_read(), want, ("fused", k, w, tie))
            with ctx_option(ctx, NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_TWO_PASS):
                ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, path, pre, w=w)
                assert_stats_equal(ctx.accum_read(), want, ("two-pass", k, w, tie))
    # a 2 M-read batch: fused == two-pass (the two-pass path is pinned against the oracle above and in the chunk test)
    n_reads = 2_000_000
    big = torch.empty(n_reads * 151 + 1024, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 0, n_reads, 150, 1, big)
    ctx.accum_reset(); ctx.reduce_device(big, n_reads * 151, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11); fused = ctx.accum_read()
    with ctx_option(ctx, NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_TWO_PASS):
        ctx.accum_reset(); ctx.reduce_device(big, n_reads * 151, 21, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11); two = ctx.accum_read()
    assert_stats_equal(fused, two, "2 M reads")
    assert fused["n_total"] > 0


def test_generic_fused_minimizers_any_k_w(ctx, restore_options):
    """The generic fused minimizer kernel (run-time k <= 31 and w <= 49; every (k, w) without a register-fused build) against the literal
    minimizer of every window (oracle: sequence::minimizer, reference src/sequence.rs:139-152, on each window of w + k - 1 good bases):
    window lengths around the lane (16 / 32 / 48 positions) and the power-of-two boundaries of the sliding minimum, k on both sides of the
    one-word / two-word values, both tie rules, ragged records, repeats (leftmost rule), with and without a quality stream; and the same
    buffer through the two-pass path."""
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNU\n", dtype=np.uint8)
    h = bytes(rng.choice(list(b"ACGT"), size=70).astype(np.uint8))
    buf = bytes(alphabet[rng.integers(0, len(alphabet), 30_000)]) + h + O.reverse_complement(h) + h + b"A" * 200 + b"AC" * 100 + b"T" * 90 + \
        O.synth_reads(0x5EED0002, 9, 1500, 150, 8).tobytes() + O.synth_reads(0x5EED0003, 0, 40, 3000, 4).tobytes()
    t = to_dev(buf)
    qual = bytes(rng.integers(33, 75, len(buf)).astype(np.uint8))
    tq = to_dev(qual)
    pairs = [(k, w) for k in (1, 4, 11, 16, 17, 23, 27, 31) for w in (1, 2, 3, 8, 13, 16, 17, 19, 31, 32, 33, 49)] + [(21, 19), (25, 11), (31, 15), (15, 25)]
    for i, (k, w) in enumerate(pairs):
        path, pre, tie, u = ((nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, True, True), (nt.PATH_BITS_CANONICAL, nt.PRE_NONE, False, False))[i & 1]
        want = O.minimizers_reduce(buf, k, w, accept_u=u, tie_rc=tie)
        ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, path, pre, w=w)
        assert_stats_equal(ctx.accum_read(), want, ("generic fused", k, w, tie))
    # k = 25 / 26: the last k of the v_min_f64 keys (value << 11 | position | strand) and the first of the general keys; the general keys
    # below 26 as well (NTK_ROUTE_NO_F64), and the generic kernel on pairs that have a register-fused build (NTK_ROUTE_NO_REGFUSED)
    for k, w, off in ((24, 7, 0), (25, 49, 0), (26, 49, 0), (25, 12, 0), (26, 12, 0), (25, 33, NL.ROUTE_NO_F64), (16, 20, NL.ROUTE_NO_F64),
                      (9, 5, NL.ROUTE_NO_F64), (21, 11, NL.ROUTE_NO_REGFUSED), (17, 16, NL.ROUTE_NO_REGFUSED)):
        with ctx_option(ctx, NL.OPT_MINIMIZER_ROUTE, off):
            for path, pre, tie, u in ((nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, True, True), (nt.PATH_BITS_CANONICAL, nt.PRE_NONE, False, False)):
                want = O.minimizers_reduce(buf, k, w, accept_u=u, tie_rc=tie)
                ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, path, pre, w=w)
                assert_stats_equal(ctx.accum_read(), want, ("generic fused", k, w, tie, off))
    for k, w, cutoff in ((23, 11, 50), (31, 19, 60), (12, 33, 40)):
        masked = O.quality_mask(buf, qual, cutoff)
        want = O.minimizers_reduce(masked, k, w, accept_u=True, tie_rc=True)
        ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w, d_qual=tq, quality_cutoff=cutoff)
        assert_stats_equal(ctx.accum_read(), want, ("generic fused, quality", k, w, cutoff))
    # beyond the kernel's range (k = 32, w = 50) and with the kernel switched off: the two-pass path, same results
    for k, w in ((32, 11), (21, 50)):
        want = O.minimizers_reduce(buf, k, w, accept_u=False, tie_rc=False)
        ctx.accum_reset(); ctx.reduce_device(t, len(buf), k, nt.PATH_BITS_CANONICAL, nt.PRE_NONE, w=w)
        assert_stats_equal(ctx.accum_read(), want, ("two-pass fallback", k, w))
    want = O.minimizers_reduce(buf, 23, 11, accept_u=True, tie_rc=True)
    with ctx_option(ctx, NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_NO_GENERIC):
        ctx.accum_reset(); ctx.reduce_device(t, len(buf), 23, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=11)
        assert_stats_equal(ctx.accum_read(), want, "two-pass with the generic kernel off")
    # a 2 M-read batch: several launches' worth of tiles per wave, generic fused == two-pass
    n_reads = 2_000_000
    big = torch.empty(n_reads * 151 + 1024, dtype=torch.uint8, device="cuda")
    ctx.synth_reads_device(0x5EED0002, 0, n_reads, 150, 1, big)
    for k, w in ((23, 11), (31, 19)):
        ctx.accum_reset(); ctx.reduce_device(big, n_reads * 151, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w); fused = ctx.accum_read()
        with ctx_option(ctx, NL.OPT_MINIMIZER_ROUTE, NL.ROUTE_NO_GENERIC):
            ctx.accum_reset(); ctx.reduce_device(big, n_reads * 151, k, nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, w=w); two = ctx.accum_read()
        assert_stats_equal(fused, two, ("2 M reads", k, w))
        assert fused["n_total"] > 0


def test_reset_flag_starts_a_new_result(ctx, monkeypatch):
    """NTK_FLAG_RESET (ntk_params.flags bit 16): the reduce call zeroes the accumulators inside its own launch - same result as
    ntk_accum_reset + the call, on every route that takes it (plain, quality-masked, fused and two-pass minimizers, empty input,
    several launches, a pinned batch, a whole-reader scan)."""
    a = O.synth_reads(0x5EED0011, 0, 3000, 150, 4).tobytes()
    b = O.synth_reads(0x5EED0012, 5, 2000, 150, 2).tobytes()
    ta, tb = to_dev(a), to_dev(b)
    path, pre = nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE
    for k in (4, 16, 21, 31):
        ctx.accum_reset(); ctx.reduce_device(ta, len(a), k, path, pre)          # leave something in the accumulators
        ctx.reduce_device(tb, len(b), k, path, pre, reset=True)
        assert_stats_equal(ctx.accum_read(), O.reduce_fused(b, k, True, True, True), ("reset", k))
        ctx.reduce_device(ta, len(a), k, path, pre)                             # and without the flag it accumulates
        both = ctx.accum_read()
        assert both["n_total"] == O.reduce_fused(a, k, True, True, True)["n_total"] + O.reduce_fused(b, k, True, True, True)["n_total"]
    # forward-only bit path
    ctx.reduce_device(tb, len(b), 21, nt.PATH_BITS, nt.PRE_NONE, reset=True)
    assert_stats_equal(ctx.accum_read(), O.reduce_fused(b, 21, False, False, False), "reset fwd")
    # quality-masked
    q = np.full(len(b), 73, dtype=np.uint8); q[::5] = 34
    tq = _qual_dev(q.tobytes())
    ctx.reduce_device(tb, len(b), 21, path, pre, d_qual=tq, quality_cutoff=40, reset=True)
    assert_stats_equal(ctx.accum_read(), O.reduce_fused(O.quality_mask(b, q.tobytes(), 40), 21, True, True, True), "reset quality")
    # minimizers: fused build, then the two-pass path
    want = O.minimizers_reduce(b, 21, 11, True, True)
    ctx.reduce_device(tb, len(b), 21, path, pre, w=11, reset=True)
    assert_stats_equal(ctx.accum_read(), want, "reset fused minimizers")
    want2 = O.minimizers_reduce(b, 21, 33, True, True)
    ctx.reduce_device(tb, len(b), 21, path, pre, w=33, reset=True)
    assert_stats_equal(ctx.accum_read(), want2, "reset two-pass minimizers")
    # empty input: only the reset happens
    ctx.reduce_device(tb, 0, 21, path, pre, reset=True)
    z = ctx.accum_read()
    assert z["n_total"] == 0 and z["sum"] == 0 and z["xor"] == 0 and int(np.asarray(z["hist"]).sum()) == 0
    # a pinned batch
    ctx.reduce_device(ta, len(a), 21, path, pre)
    bt = ctx.batch(1 << 20, 1 << 14)
    recs = b.split(b"\n")[:-1]
    for r in recs:
        assert bt.append(r, pre)
    bt.submit(21, path, pre, reset=True); bt.wait(); bt.release()
    assert_stats_equal(ctx.accum_read(), O.reduce_fused(b, 21, True, True, True), "reset batch")


from _refs import minimizer_with_position as _ref_minimizer_with_position  # noqa: E402


@pytest.mark.parametrize("chunk_bytes", [None, 97, 4096])
def test_minimizer_batch_matches_the_reference_function_per_record(ctx, restore_options, chunk_bytes):
    """ntk_minimizer_batch = sequence::minimizer (reference src/sequence.rs:139-152) applied to every record of a reader batch in one call:
    raw-byte comparison (mixed case, N, IUPAC, U - complement() maps what it maps), homopolymers and repeats (ties: the reference's loop order
    decides which window is reported), records of exactly m bytes, a record beyond 64 KiB (the one-block kernel), the reference's own literal;
    a record shorter than m fails the call and names itself; the empty batch."""
    import ctypes as C
    from needletail_amd import _lib as L
    # the upload / kernel / download pipeline over hundreds of chunks (the default is 16 MiB: one chunk here); restored at the end of the test
    ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, chunk_bytes or 0)
    rng = np.random.default_rng(41)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNnURYKMSWBDHV", dtype=np.uint8)
    assert nt.minimizer_batch([b"ATTTCG"], 3, ctx) == [b"AAA"]                      # reference src/sequence.rs:363-367
    assert nt.minimizer_batch([], 5, ctx) == []
    for m in (1, 2, 3, 8, 15, 21, 31, 40):
        recs = [bytes(alphabet[rng.integers(0, len(alphabet), int(n))]) for n in rng.integers(m, 300, 120)]
        recs += [b"A" * m, b"T" * (m + 5), b"AC" * (m + 3), bytes(rng.choice(list(b"ACGT"), size=m).astype(np.uint8))]
        h = bytes(rng.choice(list(b"ACGT"), size=m + 20).astype(np.uint8))
        recs += [h + O.reverse_complement(h), h + b"N" + h]
        # around the LDS staging limit of the wave kernel (1024 bytes) and well beyond it (candidates read from global memory)
        recs += [bytes(alphabet[rng.integers(0, len(alphabet), n)]) for n in (1023, 1024, 1025, 3000)] if m in (3, 21) else []
        mins, pos, flg = nt.minimizer_batch(recs, m, ctx, with_positions=True)
        for r, rec in enumerate(recs):
            want = _ref_minimizer_with_position(rec, m)
            assert want[0] == O.minimizer(rec, m)
            assert (mins[r], int(pos[r]), int(flg[r])) == want, (m, r, rec)
    # a long record goes through the one-block kernel; its neighbours through the wave kernel
    long_rec = O.synth_reads(0x5EED0003, 0, 1, 70_000, 4).tobytes()[:70_000]
    recs = [b"GATTACAGATTACA", long_rec, b"TTTTTTTTTTTTTTTTTTTTTTTTT"]
    mins, pos, flg = nt.minimizer_batch(recs, 12, ctx, with_positions=True)
    assert mins[0] == O.minimizer(recs[0], 12) and mins[2] == O.minimizer(recs[2], 12)
    assert mins[1] == O.minimizer(long_rec, 12)
    rc_long = O.reverse_complement(long_rec)
    assert (rc_long if flg[1] else long_rec)[int(pos[1]): int(pos[1]) + 12] == mins[1]
    # offsets that do not start at 0
    seq = b"NNNNN" + b"".join(recs[:1] + recs[2:])
    offs = np.array([5, 5 + len(recs[0]), 5 + len(recs[0]) + len(recs[2])], dtype=np.uint64)
    out = np.zeros(2 * 12, dtype=np.uint8)
    bad = C.c_uint64(0)
    L.check(L.lib().ntk_minimizer_batch(ctx._h, seq, offs.ctypes.data, 2, 12, out.ctypes.data, None, None, C.byref(bad)), "ntk_minimizer_batch")
    assert out[:12].tobytes() == mins[0] and out[12:].tobytes() == mins[2] and bad.value == 2 ** 64 - 1
    # a record shorter than m: the call fails before computing anything and names the record
    with pytest.raises(ValueError, match="record 1 "):
        nt.minimizer_batch([b"ACGTACGT", b"ACG", b"ACGTACGT"], 5, ctx)


@pytest.mark.parametrize("chunk_bytes", [None, 97, 4096])
def test_bit_kmers_planes_face_matches_the_iterator_per_record(ctx, chunk_bytes, restore_options):
    """ntk_bit_kmers_batch_planes = Sequence::bit_kmers(k, canonical) (reference src/sequence.rs:250-252, src/bitkmer.rs:39-143) for every record
    of a batch as "emitted" / "was_rc" planes per window start + dense packed values: element-wise against the oracle's literal iterator,
    record by record - every k 1..32, canonical and forward-only, ragged / empty / all-N records, mixed case, U (a break on this path), palindromes
    (ties keep the forward k-mer), records touching chunk boundaries (the three-bank pipeline over hundreds of chunks) - with the values
    downloaded and with the values packed on the host from the planes."""
    ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, chunk_bytes or 0)
    rng = np.random.default_rng(57)
    alphabet = np.frombuffer(b"ACGTACGTACGTacgtNnU-", dtype=np.uint8)
    records = [b"", b"A", b"N" * 40, b"ACGT" * 10, b"acgtACGTnACGTTGCA" * 3, b"AATT", b"GAATTC" * 6]
    for _ in range(300):
        records.append(bytes(alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 400)))]))
    records.append(bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8)))
    for k in (1, 2, 4, 11, 16, 17, 21, 31, 32):
        for canonical in (True, False):
            pl = nt.bit_kmers_planes(records, k, canonical, ctx)
            total = 0
            for i, r in enumerate(records):
                want = O.bit_kmers(r, k, canonical)
                assert list(pl.iter(i)) == want, (k, canonical, i)
                total += len(want)
            assert pl.total == total
    # values packed on the host from the planes alone (values=False: a quarter byte per base crosses PCIe)
    for k, canonical in ((21, True), (4, True), (32, False)):
        pl = nt.bit_kmers_planes(records, k, canonical, ctx, values=False)
        for i, r in enumerate(records):
            assert list(pl.iter(i)) == O.bit_kmers(r, k, canonical), (k, canonical, i, "host-packed")
    assert nt.bit_kmers_planes([], 5, True, ctx).total == 0
    with pytest.raises(ValueError):
        nt.bit_kmers_planes(records, 33, True, ctx)
