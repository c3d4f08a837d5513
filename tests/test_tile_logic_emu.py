"""CPU check of the scan kernel's per-lane bit manipulation: tests/emu/emu_scan.cpp compiles the same
source the HIP kernels use (needletail_amd/csrc/ntk_tile.hpp) for the host and emulates one wave64 in
lock-step; results are compared with the oracle.  (The GPU parity tests proper are in test_gpu_parity.py.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")


_EMU = None


def _emu_lib():
    global _EMU
    if _EMU is None:
        _EMU = _load_emu()
    return _EMU


@pytest.fixture(scope="module")
def emu():
    return _emu_lib()


def _load_emu():
    so = os.path.join(EMU_DIR, "libntk_emu.so")
    src = os.path.join(EMU_DIR, "emu_scan.cpp")
    hdr = os.path.join(HERE, "..", "needletail_amd", "csrc", "ntk_tile.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
    L = C.CDLL(so)
    L.emu_scan.restype = C.c_int
    L.emu_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.emu_encode16.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    return L


def emu_scan(L, buf: bytes, k, canon, tie_rc, accept_u, tpw=3, materialize=False):
    n = len(buf)
    npad = (n + 15) // 16 * 16
    arr = np.frombuffer(buf + b"\xAA" * (npad - n), dtype=np.uint8).copy()  # garbage in the 16-B padding
    out = np.zeros(4 + 4096, dtype=np.uint64)
    nt = (n + 1023) // 1024 * 1024
    vals = np.zeros(max(nt, 1), dtype=np.uint64) if materialize else None
    v16 = np.zeros(max(nt // 16, 1), dtype=np.uint16) if materialize else None
    r16 = np.zeros(max(nt // 16, 1), dtype=np.uint16) if materialize else None
    rc = L.emu_scan(arr.ctypes.data, n, npad, k, int(canon), int(tie_rc), int(accept_u), tpw, out.ctypes.data,
                    vals.ctypes.data if materialize else None, v16.ctypes.data if materialize else None,
                    r16.ctypes.data if materialize else None)
    assert rc == 0
    st = {"n_total": int(out[0]), "n_fwd": int(out[1]), "n_rc": int(out[0] - out[1]), "sum": int(out[2]),
          "xor": int(out[3]), "hist": out[4:].copy()}
    return (st, vals, v16, r16) if materialize else st


def assert_stats_equal(a, b, ctx=""):
    for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
        assert a[key] == b[key], (ctx, key, a[key], b[key])
    assert np.array_equal(a["hist"], b["hist"]), ctx


def test_encode16_all_bytes(emu):
    out = np.zeros(3, dtype=np.uint32)
    for accept_u in (0, 1):
        for b0 in range(256):
            for slot in (0, 5, 15):
                raw = bytearray(b"ACGTACGTACGTACGT")
                raw[slot] = b0
                emu.emu_encode16(bytes(raw), accept_u, out.ctypes.data)
                code, rcode, bad = (int(x) for x in out)
                good = bytes([b0]) in (b"A", b"C", b"G", b"T", b"a", b"c", b"g", b"t") or (accept_u and b0 in b"Uu")
                assert bool((bad >> (15 - slot)) & 1) == (not good), (accept_u, b0, slot)
                assert bad & ~(1 << (15 - slot)) == 0
                if good:
                    exp = 3 if b0 in b"Uu" else O.lib().ntko_nuc2bit(b0)
                    assert (code >> (30 - 2 * slot)) & 3 == exp
                    assert (rcode >> (2 * slot)) & 3 == 3 - exp


def test_window_masks_both_forms_against_the_definition(emu):
    """The scalar mask algebra by itself, every window length 1..32: OK[j] (window ending at byte j of the lane is emitted) from
    G[i] (byte i of the lane is a base), as 64-bit lane masks - in the finished form (window_masks / window_masks1, used by the
    round-1 kernels) and in the form whose last AND is left to the masked region (window_masks*_ab: exec = A & B) - against
    the definition: lanes 0 / 1 are halo lanes and emit nothing; lane l emits at byte j iff the k bytes ending there, reaching
    back into lanes l - 1 and l - 2, are all bases."""
    emu.emu_window_masks.restype = C.c_int
    emu.emu_window_masks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(99)
    for trial in range(60):
        p_bad = [0.0, 0.01, 0.05, 0.3][trial % 4]
        good = rng.random((64, 16)) >= p_bad                      # good[lane, byte]
        if trial == 5:
            good[:] = True
        G = np.zeros(16, dtype=np.uint64)
        for i in range(16):
            G[i] = sum(1 << l for l in range(64) if good[l, i])
        flat = good.reshape(-1)                                    # byte stream of the tile: lane * 16 + byte
        for k in range(1, 33):
            ok, ab = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
            assert emu.emu_window_masks(G.ctypes.data, k, ok.ctypes.data, ab.ctypes.data) == 0
            want = np.zeros(16, dtype=np.uint64)
            for l in range(2, 64):
                for j in range(16):
                    e = l * 16 + j
                    if flat[e - k + 1: e + 1].all():
                        want[j] |= np.uint64(1 << l)
            assert np.array_equal(ok, want), (trial, k, "finished form")
            assert np.array_equal(ab, want), (trial, k, "A & B form")
        # windows of 33 .. 48 bytes (fused minimizers such as (23, 11)): three halo lanes, the span form of the algebra at compile time
        emu.emu_window_masks_wide.restype = C.c_int
        emu.emu_window_masks_wide.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        if trial < 24:
            for km in range(33, 49):
                ab = np.zeros(16, dtype=np.uint64)
                assert emu.emu_window_masks_wide(G.ctypes.data, km, ab.ctypes.data) == 0
                want = np.zeros(16, dtype=np.uint64)
                for l in range(3, 64):
                    for j in range(16):
                        e = l * 16 + j
                        if e - km + 1 >= 0 and flat[e - km + 1: e + 1].all():
                            want[j] |= np.uint64(1 << l)
                assert np.array_equal(ab, want), (trial, km, "three halo lanes")


@pytest.mark.parametrize("k", list(range(1, 33)))
def test_emu_vs_oracle_synthetic(emu, k):
    buf = O.synth_reads(0x5EED0002, 0, 40, 150, 8).tobytes()  # ~0.8% N, '\n' separators
    for canon, tie_rc, accept_u in ((1, 1, 1), (1, 0, 0), (0, 0, 0)):
        want = O.reduce_fused(buf, k, bool(canon), bool(tie_rc), bool(accept_u))
        # bit 0: k-specialised build (k = 21, 31), bit 1: scalar-validity variant, bit 2: sv2 (k >= 17), bit 3: its 14-bit histogram
        for tpw in (1, 2, 3, 7, 4, 12):
            got = emu_scan(emu, buf, k, canon, tie_rc, accept_u, tpw)
            assert_stats_equal(got, want, (k, canon, tie_rc, accept_u, tpw))


def test_emu_vs_literal_reference_chain(emu, golden_dir):
    from _fastx import fastq_raw_seqs
    recs = fastq_raw_seqs(open(os.path.join(golden_dir, "PRJNA271013_head.fq"), "rb").read())[:300]
    buf = b"".join(r + b"\n" for r in recs)
    for k in (4, 21, 31):
        want = O.reduce_records(recs, k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
        assert_stats_equal(emu_scan(emu, buf, k, 1, 1, 1), want, ("bytes", k))
        want = O.reduce_records(recs, k, O.PATH_BITS_CANONICAL, O.PRE_NONE)
        assert_stats_equal(emu_scan(emu, buf, k, 1, 0, 0), want, ("bits", k))


def test_emu_random_alphabet_and_lengths(emu):
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtACGTACGTNnUuRY-\n\x00\xff", dtype=np.uint8)
    for trial in range(60):
        n = int(rng.integers(0, 2500))
        buf = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        k = int(rng.integers(1, 33))
        for canon, tie_rc, accept_u in ((1, 1, 1), (1, 0, 0), (0, 0, 1)):
            want = O.reduce_fused(buf, k, bool(canon), bool(tie_rc), bool(accept_u))
            got = emu_scan(emu, buf, k, canon, tie_rc, accept_u, int(rng.integers(1, 16)))
            assert_stats_equal(got, want, (trial, n, k, canon, tie_rc, accept_u))


@pytest.mark.parametrize("k", list(range(17, 33)))
def test_emu_sv2_forward_only(emu, k):
    """lane_tile_sv2_fwd (BitNuclKmer with canonical = false, reference src/bitkmer.rs:80-108): several tiles, breaks at lane and
    tile edges, both histogram sizes, with and without U."""
    rng = np.random.default_rng(100 + k)
    alphabet = np.frombuffer(b"ACGT" * 8 + b"acgtNUu\n", dtype=np.uint8)
    n = int(rng.integers(3000, 7000))
    a = alphabet[rng.integers(0, len(alphabet), n)].copy()
    for at in (15, 16, 17, 991, 992, 993, 1007, 1008, 1984, 2 * 992 - k, 3 * 992 + k):
        a[at] = ord("N")
    a[2500:2500 + k - 1] = ord("A"); a[2499] = a[2500 + k - 1] = ord("N")   # k-1 good bases between two breaks: no window
    a[2600:2600 + k] = ord("C"); a[2599] = a[2600 + k] = ord("N")           # exactly one window
    buf = a.tobytes()
    for accept_u in (0, 1):
        want = O.reduce_fused(buf, k, False, False, bool(accept_u))
        assert want["n_fwd"] == want["n_total"] and want["n_rc"] == 0
        for tpw in (4, 12):
            assert_stats_equal(emu_scan(emu, buf, k, 0, 0, accept_u, tpw), want, (k, accept_u, tpw))


@pytest.mark.parametrize("k", list(range(1, 17)))
def test_emu_sv2_word_builds(emu, k):
    """lane_tile_sv2w (k <= 16: one-word values kept left-aligned; canonical with both tie rules, and forward-only): several
    tiles, breaks at lane and tile edges, palindromic stretches (strand ties for even k), both histogram sizes."""
    rng = np.random.default_rng(200 + k)
    alphabet = np.frombuffer(b"ACGT" * 8 + b"acgtNUu\n", dtype=np.uint8)
    n = int(rng.integers(3000, 7000))
    a = alphabet[rng.integers(0, len(alphabet), n)].copy()
    for at in (15, 16, 17, 991, 992, 993, 1007, 1008, 1984, 2 * 992 - k, 3 * 992 + k):
        a[at] = ord("N")
    a[1200:1400] = np.resize(np.frombuffer(b"ACGT", dtype=np.uint8), 200)      # ACGT repeats: every even-k window is a palindrome
    a[1500:1600] = np.resize(np.frombuffer(b"AT", dtype=np.uint8), 100)
    a[2500:2500 + k - 1] = ord("A"); a[2499] = a[2500 + k - 1] = ord("N")      # k-1 good bases between two breaks: no window
    a[2600:2600 + k] = ord("C"); a[2599] = a[2600 + k] = ord("N")              # exactly one window
    buf = a.tobytes()
    for canon, tie_rc, accept_u in ((1, 1, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0), (0, 0, 0), (0, 0, 1)):
        want = O.reduce_fused(buf, k, bool(canon), bool(tie_rc), bool(accept_u))
        for tpw in (4, 12):
            assert_stats_equal(emu_scan(emu, buf, k, canon, tie_rc, accept_u, tpw), want, (k, canon, tie_rc, accept_u, tpw))


def test_emu_materialize_matches_bit_kmers(emu):
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTACGTACGTacgtN", dtype=np.uint8)
    for k, canonical in ((3, True), (16, True), (17, False), (21, True), (32, True)):
        seq = bytes(alphabet[rng.integers(0, len(alphabet), 1500)])
        st, vals, v16, r16 = emu_scan(emu, seq, k, canonical, 0, 0, 2, materialize=True)
        pos, val, flg = O.bit_kmers_arrays(seq, k, canonical)
        e = np.arange(len(seq))
        valid = (v16[e // 16] >> (15 - e % 16)) & 1
        rcb = (r16[e // 16] >> (15 - e % 16)) & 1
        ends = e[valid == 1]
        assert np.array_equal(ends - (k - 1), pos.astype(np.int64))
        assert np.array_equal(vals[ends], val)
        assert np.array_equal(rcb[ends].astype(np.uint8), flg)


# ---- property-based differential test (the reference has no randomised k-mer tests) -------------------
from hypothesis import given, settings, strategies as st_  # noqa: E402

_ALPHABET = b"ACGT" * 6 + b"acgt" + b"NnUuRYKM-.* \t\r\n\x00\x7f\x80\xff0@>"


@settings(max_examples=150, deadline=None)
@given(data=st_.lists(st_.sampled_from(list(_ALPHABET)), min_size=0, max_size=1400).map(bytes),
       k=st_.integers(1, 32), mode=st_.sampled_from([(1, 1, 1), (1, 0, 0), (0, 0, 0), (1, 0, 1), (1, 1, 0)]),
       variant=st_.integers(0, 15))
def test_emu_matches_oracle_property(data, k, mode, variant):
    L = _emu_lib()
    canon, tie_rc, accept_u = mode
    want = O.reduce_fused(data, k, bool(canon), bool(tie_rc), bool(accept_u))
    got = emu_scan(L, data, k, canon, tie_rc, accept_u, variant)
    assert_stats_equal(got, want, (k, mode, variant, len(data)))


# ---- quality masking fused into the load (reference src/sequence.rs:285-296) --------------------

def emu_scan_quality(L, buf: bytes, qual: bytes, cutoff, k, canon, tie_rc, accept_u, tpw=3):
    n = len(buf)
    assert len(qual) == n
    npad = (n + 15) // 16 * 16
    arr = np.frombuffer(buf + b"\xAA" * (npad - n), dtype=np.uint8).copy()
    qarr = np.frombuffer(qual + b"\x00" * (npad - n), dtype=np.uint8).copy()
    out = np.zeros(4 + 4096, dtype=np.uint64)
    L.emu_scan_quality.restype = C.c_int
    L.emu_scan_quality.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                   C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.emu_scan_quality(arr.ctypes.data, qarr.ctypes.data, cutoff, n, npad, k, int(canon), int(tie_rc), int(accept_u),
                            tpw, out.ctypes.data, None, None, None)
    assert rc == 0
    return {"n_total": int(out[0]), "n_fwd": int(out[1]), "n_rc": int(out[0] - out[1]), "sum": int(out[2]),
            "xor": int(out[3]), "hist": out[4:].copy()}


def test_quality_compare_is_exact_for_every_byte_and_cutoff(emu):
    """The SWAR `q < cutoff` of ntk_tile.hpp quality_break against the plain comparison, all 256 quality bytes x all 255
    cutoffs, observed through the scan: one valid 1-mer per unmasked base."""
    seq = b"ACGT" * 64                       # 256 bases, one per quality value
    qual = bytes(range(256))
    for cutoff in range(1, 256):
        got = emu_scan_quality(emu, seq, qual, cutoff, 1, 0, 0, 0)
        assert got["n_total"] == 256 - cutoff, cutoff
        masked = O.quality_mask(seq, qual, cutoff)
        assert masked == b"".join(b"N" if q < cutoff else bytes([s]) for s, q in zip(seq, qual))
        assert_stats_equal(got, O.reduce_fused(masked, 1, False, False, False), cutoff)


def test_emu_quality_scan_matches_masked_oracle(emu):
    rng = np.random.default_rng(77)
    for trial in range(40):
        n = int(rng.integers(0, 3000))
        buf = bytes(rng.choice(list(_ALPHABET), size=n).astype(np.uint8))
        qual = bytes(rng.integers(33, 75, size=n, dtype=np.uint8)) if trial % 3 else bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        cutoff = int(rng.integers(1, 256)) if trial % 3 == 0 else int(rng.integers(34, 75))
        k = int(rng.integers(1, 33))
        canon, tie_rc, accept_u = [(1, 1, 1), (1, 0, 0), (0, 0, 0), (1, 0, 1)][trial % 4]
        # the reference masks (base, quality) pairs first, then runs the chain (src/sequence.rs:285-296)
        masked = O.quality_mask(buf, qual, cutoff)
        want = O.reduce_fused(masked, k, bool(canon), bool(tie_rc), bool(accept_u))
        for tpw in (0, 3, 4, 12):
            got = emu_scan_quality(emu, buf, qual, cutoff, k, canon, tie_rc, accept_u, tpw)
            assert_stats_equal(got, want, (trial, k, cutoff, tpw))


# ---- fused windowed minimizers (ntk_tile.hpp lane_tile_sv2_min; BASELINE.json configs[4]) ---------------------------

def emu_minimizers(L, buf: bytes, k, w, tie_rc, accept_u, hb14):
    n = len(buf)
    npad = (n + 15) // 16 * 16
    arr = np.frombuffer(buf + b"\xAA" * (npad - n), dtype=np.uint8).copy()
    out = np.zeros(4 + 4096, dtype=np.uint64)
    L.emu_minimizers.restype = C.c_int
    L.emu_minimizers.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.emu_minimizers(arr.ctypes.data, n, npad, k, w, int(tie_rc), int(accept_u), int(hb14), out.ctypes.data)
    if rc == -2:
        return None
    assert rc == 0
    return {"n_total": int(out[0]), "n_fwd": int(out[1]), "n_rc": int(out[0] - out[1]), "sum": int(out[2]),
            "xor": int(out[3]), "hist": out[4:].copy()}


# (k <= 16: the value is one word and the key is built from it; the product ships k = 15..22 x w = 9..12 with k + w - 1 <= 32)
FUSED_KW = ((21, 11), (17, 11), (18, 11), (19, 11), (20, 11), (22, 11), (21, 9), (21, 10), (21, 12),
            (15, 10), (15, 9), (16, 12), (16, 16), (15, 16), (19, 10), (22, 9), (20, 13),
            (15, 5), (19, 5), (21, 5), (23, 5), (16, 2), (17, 3), (18, 4), (20, 6), (22, 7), (19, 8),   # short windows (w <= 8): the doubling path
            (23, 9), (23, 10), (23, 11), (23, 12), (22, 12), (21, 16), (23, 16))   # k = 23; windows of 33 .. 38 bytes: three halo lanes (Sv2Geom)


def test_emu_fused_minimizers_match_the_literal_minimizer(emu):
    """Keys, strand bit, leftmost tie rule, lane-boundary import and the histogram-derived digests of the fused minimizer
    build against sequence::minimizer applied window by window (oracle, reference src/sequence.rs:139-152): random text
    with breaks, reverse-complement palindromes, homopolymer runs (every window ties), both tie rules."""
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtACGTACGTACGTACGTNU\n", dtype=np.uint8)
    assert emu_minimizers(emu, b"ACGT", 24, 11, 1, 1, 0) is None     # no register-fused build: the generic fused kernel serves it (below)
    for trial in range(40):
        n = int(rng.integers(0, 2600))
        b = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        if trial % 4 == 0:
            h = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(10, 60))).astype(np.uint8))
            b = b + h + O.reverse_complement(h) + h + b"A" * 70 + b"T" * 70 + b[:80]
        for k, w in FUSED_KW:
            for tie, u in ((1, 1), (0, 0)):
                want = O.minimizers_reduce(b, k, w, accept_u=bool(u), tie_rc=bool(tie))
                hb14 = 1 if k == 23 else trial % 2   # k = 23: the digests' high parts follow from the histogram only with 14-bit cells (2k - 14 <= 32), the product's size
                assert_stats_equal(emu_minimizers(emu, b, k, w, tie, u, hb14), want, (trial, k, w, tie, u))


# ---- the generic fused minimizer kernel (ntk_tile.hpp minimizer_windows; ntk_kernels.hpp minimizer_scan_kernel) ---------------------

def emu_minimizers_generic(L, buf: bytes, k, w, tie_rc, accept_u, f64):
    n = len(buf)
    npad = (n + 15) // 16 * 16
    arr = np.frombuffer(buf + b"\xAA" * (npad - n), dtype=np.uint8).copy()
    out = np.zeros(4 + 4096, dtype=np.uint64)
    L.emu_minimizers_generic.restype = C.c_int
    L.emu_minimizers_generic.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.emu_minimizers_generic(arr.ctypes.data, n, npad, k, w, int(tie_rc), int(accept_u), int(f64), out.ctypes.data)
    if rc == -2:
        return None
    assert rc == 0
    return {"n_total": int(out[0]), "n_fwd": int(out[1]), "n_rc": int(out[0] - out[1]), "sum": int(out[2]),
            "xor": int(out[3]), "hist": out[4:].copy()}


def test_emu_generic_minimizers_match_the_literal_minimizer(emu):
    """The per-lane source of the generic fused minimizer kernel (any k <= 31, w <= 49 at run time), lock-step over 64 lanes with the kernel's
    run-time tile geometry, against sequence::minimizer applied window by window (oracle, reference src/sequence.rs:139-152): both key forms
    (k <= 25: value, tile position and strand in one ordered double under a plain minimum; the general keys under the left-preferring
    minimum), every doubling round and every overlap shift (w = 1..49), windows reaching one / two / three lanes back, k on both sides of
    the one-word / two-word values and of the key forms, both tie rules and alphabets, inverted repeats within a window (equal values on
    opposite strands: the leftmost rule), homopolymers, ragged records."""
    rng = np.random.default_rng(29)
    alphabet = np.frombuffer(b"ACGTacgtACGTACGTACGTACGTNU\n", dtype=np.uint8)
    assert emu_minimizers_generic(emu, b"ACGT", 32, 11, 1, 1, 0) is None and emu_minimizers_generic(emu, b"ACGT", 21, 50, 1, 1, 0) is None
    assert emu_minimizers_generic(emu, b"ACGT", 26, 11, 1, 1, 1) is None   # f64 keys hold k <= 25
    ws = list(range(1, 50))
    for trial in range(14):
        n = int(rng.integers(0, 3200))
        b = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        if trial % 3 == 0:
            h = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(10, 70))).astype(np.uint8))
            b = b + h + O.reverse_complement(h) + h + b"A" * 90 + b"AC" * 40 + b"T" * 70 + b[:80]
        if trial % 5 == 1:
            b = bytes(rng.choice(list(b"ACGT"), size=2500).astype(np.uint8))   # no breaks: every window of a long run
        picks = [(int(rng.choice([1, 3, 8, 11, 15, 16, 17, 21, 24, 25, 26, 28, 31])), w) for w in rng.choice(ws, size=7, replace=False)]
        picks += [(25, 49), (26, 49), (16, 17), (17, 16), (31, 33), (23, 11), (12, 32)][trial % 7: trial % 7 + 2]
        for k, w in picks:
            for tie, u in ((1, 1), (0, 0)):
                want = O.minimizers_reduce(b, k, w, accept_u=bool(u), tie_rc=bool(tie))
                assert_stats_equal(emu_minimizers_generic(emu, b, k, w, tie, u, 0), want, (trial, k, w, tie, u, "general keys"))
                if k <= 25:
                    assert_stats_equal(emu_minimizers_generic(emu, b, k, w, tie, u, 1), want, (trial, k, w, tie, u, "f64 keys"))


def test_emu_generic_minimizers_every_window_length(emu):
    """w = 1..49 one by one (every overlap shift 0..17 behind every doubling depth) on one buffer, k = 19 (two words, f64 keys) and k = 27."""
    rng = np.random.default_rng(31)
    h = bytes(rng.choice(list(b"ACGT"), size=50).astype(np.uint8))
    b = bytes(rng.choice(list(b"ACGT"), size=1500).astype(np.uint8)) + b"N" + h + O.reverse_complement(h) + b"G" * 120 + b"\n" + h[:30] * 4
    for w in range(1, 50):
        for k, f64 in ((19, 1), (27, 0)):
            want = O.minimizers_reduce(b, k, w, accept_u=True, tie_rc=True)
            assert_stats_equal(emu_minimizers_generic(emu, b, k, w, 1, 1, f64), want, (k, w, f64))


def test_emu_generic_minimizers_with_a_quality_stream(emu):
    """The QM builds of the generic fused minimizer kernel: the quality tile folded into the sequence bytes (quality_break16) ahead of the same
    per-lane source; against the oracle's quality_mask followed by the literal minimizer of every window."""
    rng = np.random.default_rng(37)
    L = emu
    L.emu_minimizers_generic_quality.restype = C.c_int
    L.emu_minimizers_generic_quality.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                 C.c_int, C.c_void_p]
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNU\n", dtype=np.uint8)
    for trial in range(6):
        n = int(rng.integers(200, 2600))
        b = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        q = rng.integers(33, 75, n, dtype=np.uint8)
        if trial % 2:
            q[:] = 73; q[rng.integers(0, n, max(1, n // 40))] = 34
        npad = (n + 15) // 16 * 16
        arr = np.frombuffer(b + b"\xAA" * (npad - n), dtype=np.uint8).copy()
        qarr = np.frombuffer(q.tobytes() + b"\x49" * (npad - n), dtype=np.uint8).copy()
        for k, w, cutoff, f64 in ((23, 11, 50, 1), (31, 19, 60, 0), (12, 33, 40, 1), (17, 16, 35, 0), (25, 49, 45, 1)):
            out = np.zeros(4 + 4096, dtype=np.uint64)
            assert L.emu_minimizers_generic_quality(arr.ctypes.data, qarr.ctypes.data, cutoff, n, npad, k, w, 1, 1, f64, out.ctypes.data) == 0
            got = {"n_total": int(out[0]), "n_fwd": int(out[1]), "n_rc": int(out[0] - out[1]), "sum": int(out[2]), "xor": int(out[3]), "hist": out[4:].copy()}
            want = O.minimizers_reduce(O.quality_mask(b, q.tobytes(), cutoff), k, w, accept_u=True, tie_rc=True)
            assert_stats_equal(got, want, (trial, k, w, cutoff, f64))


def test_emu_window_masks_at_run_time(emu):
    """window_masks_runtime (the generic fused minimizer kernel's validity: window length L = k + w - 1 known at run time, 1 .. 79; a 16-way
    switch on (L - 2) mod 16 around the compile-time-indexed prefix / suffix algebra) against the definition: the window of L bytes ending at
    byte j of lane l is emitted iff lane l >= halo lanes and bytes 16 l + j - L + 1 .. 16 l + j are all bases (positions before the tile are not)."""
    emu.emu_window_masks_runtime.restype = C.c_int
    emu.emu_window_masks_runtime.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(77)
    for trial in range(12):
        p_break = [0.0, 0.01, 0.03, 0.1, 0.3, 0.6][trial % 6]
        good = rng.random(1024) >= p_break                      # byte p = 16 * lane + i
        if trial == 7:
            good[:] = True
        g = np.zeros(16, dtype=np.uint64)
        for i in range(16):
            g[i] = sum(1 << l for l in range(64) if good[16 * l + i])
        run = np.zeros(1024, dtype=np.int64)                    # length of the run of bases ending at p
        for p in range(1024):
            run[p] = (run[p - 1] + 1 if p else 1) if good[p] else 0
        for L in range(1, 80):
            halo = 2 + (max(L - 31, 0) + 15) // 16              # the kernel's: 2 lanes of k-mer halo + ceil((w - 1) / 16) with k = 31 (any split of L does)
            halo = min(halo, 5)
            ok = np.zeros(16, dtype=np.uint64)
            assert emu.emu_window_masks_runtime(g.ctypes.data, L, halo, ok.ctypes.data) == 0
            for j in range(16):
                want = sum(1 << l for l in range(halo, 64) if run[16 * l + j] >= L)
                assert int(ok[j]) == want, (trial, L, j, hex(int(ok[j])), hex(want))


def _wide_reference(recs, k, normalized):
    """CanonicalKmers with 33 <= k <= 255 per record through the oracle's literal iterator: counters + the histogram of the leading six bases
    of every emitted slice (as tests/test_gpu_parity.py holds it for the GPU)."""
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i; code[ch | 0x20] = i
    st = {"n_total": 0, "n_fwd": 0, "hist": np.zeros(4096, dtype=np.uint64)}
    for r in recs:
        if normalized:
            r = O.normalize(r)[0]
        rc = O.reverse_complement(r)
        pos, flg = O.canonical_kmers_arrays(r, rc, k)
        for p, f in zip(pos.tolist(), flg.tolist()):
            sl = rc[len(rc) - p - k: len(rc) - p] if f else r[p: p + k]
            b = 0
            for ch in sl[:6]:
                b = b * 4 + int(code[ch])
            st["hist"][b] += 1
        st["n_total"] += len(pos); st["n_fwd"] += len(pos) - int(flg.sum())
    return st


def test_emu_wide_k_reduce(emu):
    """Round 6: CanonicalKmers with k = 33..255 from the packed 2-bit streams (wide_canonical_reduce_kernel): the per-slot functions of
    ntk_tile.hpp - break mask and last break per slot, the 16-bit window mask from the last break before the slot, the word / bit offset of the
    window's start, the strand on the first 32 bases - run tile by tile on the host against the literal iterator: upper-case records with
    lengths around k, the 16-byte slots and the 4096-byte tiles, breaks anywhere, every k mod 16.  Inverted repeats and lower case must raise
    the flags on which the kernel's launch is redone by the byte-walking kernel."""
    emu.emu_wide_reduce.restype = C.c_int
    emu.emu_wide_reduce.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]

    def run(buf, k, accept_u):
        n = len(buf)
        npad = (n + 15) // 16 * 16
        arr = np.frombuffer(buf + b"\xAA" * (npad - n), dtype=np.uint8).copy()   # garbage in the 16-byte padding: beyond n everything is a break
        out = np.zeros(4 + 4096, dtype=np.uint64)
        assert emu.emu_wide_reduce(arr.ctypes.data, n, npad, k, int(accept_u), out.ctypes.data) == 0
        return {"n_total": int(out[0]), "n_fwd": int(out[1]), "ties": int(out[2]), "bit5": int(out[3]), "hist": out[4:].copy()}

    rng = np.random.default_rng(3356)
    recs = []
    for n in [0, 1, 32, 33, 34, 47, 48, 49, 64, 65, 254, 255, 256, 257, 271, 272, 4095, 4096, 4097, 4351, 4352, 9000] + [int(x) for x in rng.integers(0, 1200, 30)]:
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        a[rng.random(n) > 0.998] = ord("N")
        recs.append(a.tobytes())
    buf = b"\n".join(recs) + b"\n"
    for k in list(range(33, 50)) + [63, 64, 65, 97, 128, 200, 254, 255]:
        want = _wide_reference(recs, k, False)
        for accept_u in (False, True):
            got = run(buf, k, accept_u)
            assert got["ties"] == 0
            assert (got["n_total"], got["n_fwd"]) == (want["n_total"], want["n_fwd"]) and np.array_equal(got["hist"], want["hist"]), (k, accept_u)
    # U is a base only on input read as normalised (reference src/sequence.rs:30); lower case reads the same there
    recs_u = [np.frombuffer(b"ACGUacgutT", dtype=np.uint8)[rng.integers(0, 10, n)].tobytes() for n in (20, 33, 120, 300, 700)]
    buf_u = b"\n".join(recs_u) + b"\n"
    for k in (33, 50, 77):
        want = _wide_reference(recs_u, k, True)
        got = run(buf_u, k, True)
        assert got["ties"] == 0 and (got["n_total"], got["n_fwd"]) == (want["n_total"], want["n_fwd"]) and np.array_equal(got["hist"], want["hist"]), k
    assert run(buf_u, 33, False)["bit5"] == 1 and run(buf, 33, False)["bit5"] == 0   # (un-normalised input with bit 5 anywhere: the launch is redone)
    # windows equal to their reverse complement over 32 bases: counted, so that the kernel can hand the launch to the byte-walking kernel
    for rec in (b"ACGT" * 80, b"AT" * 100, b"A" * 300 + b"T" * 300):
        assert run(rec + b"\n", 34, True)["ties"] > 0
    assert run(b"A" * 300 + b"T" * 300 + b"\n", 255, True)["ties"] > 0
