"""ntk_gunzip (the gzip front-end of the parallel producer) against Python's zlib on the CPU: every route (block gzip,
speculative parallel inflate of an ordinary stream, sequential), every deflate block type, many members, and the reference reader's
error semantics - MultiGzDecoder (reference src/parser/mod.rs:95-108) reads EVERY member, and a truncated or corrupt stream is an error,
not a short read."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from _fastx import bgzf_compress

from needletail_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NTK_OK, NTK_ERR_PARSE = 0, 8


def gunzip(data: bytes, threads: int):
    out, n, info = C.c_void_p(), C.c_uint64(0), L.GunzipInfo()
    rc = L.lib().ntk_gunzip(data, len(data), threads, C.byref(out), C.byref(n), C.byref(info))
    if rc != NTK_OK:
        assert not out.value
        return rc, None, info
    got = C.string_at(out.value, n.value) if n.value else b""
    L.lib().ntk_gunzip_free(out, n.value)
    return rc, got, info


def fastq(n_reads: int, seed: int, noisy: bool) -> bytes:
    rng = np.random.default_rng(seed)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n_reads, 100))]
    quals = rng.integers(33, 74, (n_reads, 100)).astype(np.uint8) if noisy else np.full((n_reads, 100), 73, dtype=np.uint8)
    return b"".join(b"@r%07d\n%s\n+\n%s\n" % (i, seqs[i].tobytes(), quals[i].tobytes()) for i in range(n_reads))


def deflate(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = bytearray()
    for i in range(0, len(data), flush_every):
        out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)
    return bytes(out + c.flush())


TEXT = fastq(30_000, 7, True)      # ~6.3 MB of text: several chunks per thread
EASY = fastq(30_000, 8, False)


@pytest.mark.parametrize("threads", [1, 2, 5, 8])
def test_ordinary_stream_every_thread_count(threads):
    for text in (TEXT, EASY):
        z = deflate(text)
        rc, got, info = gunzip(z, threads)
        assert rc == NTK_OK and got == text
        assert info.route == (3 if threads == 1 else 2) and info.members == 1 and info.threads == threads
        if threads > 1:
            assert info.chunks > 1 and info.marker_symbols > 0   # chunks really entered the stream in the middle


@pytest.mark.parametrize("how", ["stored", "level1", "level9", "fixed", "huffman", "rle", "flush100", "flush5000"])
def test_every_block_type_and_strategy(how):
    text = TEXT[:2_500_000]
    z = {"stored": lambda: deflate(text, 0), "level1": lambda: deflate(text, 1), "level9": lambda: deflate(text, 9),
         "fixed": lambda: deflate(text, 6, zlib.Z_FIXED), "huffman": lambda: deflate(text, 6, zlib.Z_HUFFMAN_ONLY),
         "rle": lambda: deflate(EASY[:2_500_000], 6, zlib.Z_RLE), "flush100": lambda: deflate(text[:400_000], 6, flush_every=100),
         "flush5000": lambda: deflate(text, 6, flush_every=5000)}[how]()
    want = gzip.decompress(z)
    for threads in (1, 4):
        rc, got, _ = gunzip(z, threads)
        assert rc == NTK_OK and got == want


def test_many_members_padding_and_garbage():
    parts = [TEXT[:900_000], b"", EASY[:1_200_000], b"x", TEXT[900_000:2_000_000]]
    z = b"".join(deflate(p) for p in parts)
    for threads in (1, 3, 8):
        rc, got, info = gunzip(z, threads)
        assert rc == NTK_OK and got == b"".join(parts) and info.members == len(parts)
    rc, got, _ = gunzip(z + b"\0\0\0", 4)                 # trailing zero padding is tolerated (as zlib-based readers do)
    assert rc == NTK_OK and got == b"".join(parts)
    assert gunzip(z + b"garbage", 4)[0] == NTK_ERR_PARSE   # anything else after a member must be a member
    # header fields: FEXTRA, FNAME, FCOMMENT, FHCRC
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = body.compress(parts[0]) + body.flush()
    hdr = b"\x1f\x8b\x08\x1e" + b"\0" * 6 + struct.pack("<H", 5) + b"hello" + b"name\0" + b"comment\0" + b"\x12\x34"
    z2 = hdr + raw + struct.pack("<II", zlib.crc32(parts[0]), len(parts[0]))
    assert gzip.decompress(z2) == parts[0]
    for threads in (1, 4):
        assert gunzip(z2, threads)[1] == parts[0]


def test_block_gzip_route():
    z = bgzf_compress(TEXT)
    rc, got, info = gunzip(z, 4)
    assert rc == NTK_OK and got == TEXT
    assert info.route in (1, 2)      # 1 when libdeflate.so.0 is loadable (it is in the image), else the ordinary route
    bad = bytearray(z); bad[len(z) // 2] ^= 0x40
    assert gunzip(bytes(bad), 4)[0] == NTK_ERR_PARSE


def test_the_reference_data_file():
    z = open(os.path.join(ROOT, "tests", "golden", "test.fa.gz"), "rb").read()
    for threads in (1, 4):
        rc, got, _ = gunzip(z, threads)
        assert rc == NTK_OK and got == gzip.decompress(z) == open(os.path.join(ROOT, "tests", "golden", "test.fa"), "rb").read()


def test_truncation_and_corruption_are_errors():
    z = deflate(TEXT)
    for cut in (len(z) - 1, len(z) - 8, len(z) - 9, len(z) // 2, len(z) // 3, 30, 12):
        for threads in (1, 6):
            assert gunzip(z[:cut], threads)[0] == NTK_ERR_PARSE, cut
    rng = np.random.default_rng(3)
    for _ in range(25):   # one flipped bit anywhere in the deflate data: either the decoder or the CRC catches it
        bad = bytearray(z)
        bad[int(rng.integers(12, len(z) - 8))] ^= 1 << int(rng.integers(0, 8))
        assert gunzip(bytes(bad), 6)[0] == NTK_ERR_PARSE
    for at in (len(z) - 7, len(z) - 2):   # the trailer: CRC-32, ISIZE
        bad = bytearray(z); bad[at] ^= 0x10
        assert gunzip(bytes(bad), 6)[0] == NTK_ERR_PARSE
    assert gunzip(b"\x1f\x8b" + b"\0" * 30, 2)[0] == NTK_ERR_PARSE
    assert gunzip(b"not gzip at all, not at all", 2)[0] == NTK_ERR_PARSE
    out, n = C.c_void_p(), C.c_uint64(0)
    assert L.lib().ntk_gunzip(None, 0, 2, C.byref(out), C.byref(n), None) == 2   # NTK_ERR_BAD_ARG


def test_truncated_literal_heavy_streams_end_at_once():
    """a stream cut inside a block whose all-zero Huffman code is a LITERAL kept the decoder busy on phantom zero bits
    until the output limit (seconds, gigabytes, NTK_ERR_UNSUPPORTED instead of NTK_ERR_PARSE): the literal path never looked at the input
    position.  Z_HUFFMAN_ONLY (no matches at all) and incompressible data at level 1 (hardly any) are the cases the match path's check
    never saw; every cut must be a parse error within a fraction of a second."""
    import time
    rng = np.random.default_rng(5)
    low_match = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3_000_000)])
    for name, z in (("huffman only, FASTQ", deflate(TEXT, 6, zlib.Z_HUFFMAN_ONLY)), ("huffman only, bases", deflate(low_match, 6, zlib.Z_HUFFMAN_ONLY)),
                    ("level 1, random bytes", deflate(bytes(rng.integers(0, 256, 2_000_000).astype(np.uint8)), 1)),
                    ("filtered", deflate(low_match, 6, zlib.Z_FILTERED))):
        for cut in sorted(set(int(x) for x in rng.integers(20, len(z) - 9, 6)) | {len(z) - 9, len(z) // 2}):
            for threads in (1, 4):
                t0 = time.perf_counter()
                rc = gunzip(z[:cut], threads)[0]
                dt = time.perf_counter() - t0
                assert rc == NTK_ERR_PARSE and dt < 2.0, (name, cut, threads, rc, dt)


def test_high_ratio_chunks_are_deferred_not_grown():
    """a chunk that is not at the head of the chain decodes against a cap (96 M symbols) instead of the global limit; one
    that outgrows it gives its buffers back and is decoded again when it is the head.  0.7 GB of one byte value at level 9 is ~0.7 MB of
    deflate data - two chunks of 350 MB each: every speculative one is deferred, the result is still the input."""
    n = 700_000_000
    c = zlib.compressobj(9, zlib.DEFLATED, 31)
    z = b"".join(c.compress(b"\x07" * (1 << 24)) for _ in range(n >> 24)) + c.compress(b"\x07" * (n & ((1 << 24) - 1))) + c.flush()
    out, n_out, info = C.c_void_p(), C.c_uint64(0), L.GunzipInfo()
    assert L.lib().ntk_gunzip(z, len(z), 4, C.byref(out), C.byref(n_out), C.byref(info)) == NTK_OK
    try:
        assert n_out.value == n and info.route == 2
        got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n,))
        assert int(got.min()) == 7 and int(got.max()) == 7
        if info.chunks > 1:
            assert info.chunks_deferred >= 1 or info.chunks_dropped >= info.chunks - 1, (info.chunks, info.chunks_deferred, info.chunks_dropped)
    finally:
        del got
        L.lib().ntk_gunzip_free(out, n_out.value)


def test_random_structured_inputs():
    """Data with long-range repeats, runs and incompressible stretches; random flush points make blocks of every size."""
    rng = np.random.default_rng(11)
    for trial in range(6):
        pieces = []
        for _ in range(int(rng.integers(20, 60))):
            kind = int(rng.integers(0, 4))
            if kind == 0: pieces.append(bytes(rng.integers(0, 256, int(rng.integers(1, 60_000))).astype(np.uint8)))
            elif kind == 1: pieces.append(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 80_000)))
            elif kind == 2 and pieces: pieces.append(pieces[int(rng.integers(0, len(pieces)))])
            else: pieces.append(TEXT[int(rng.integers(0, 10**6)):][: int(rng.integers(1, 200_000))])
        data = b"".join(pieces)
        z = deflate(data, int(rng.integers(1, 10)), flush_every=int(rng.choice([0, 0, 777, 40_000])))
        for threads in (2, 7):
            rc, got, _ = gunzip(z, threads)
            assert rc == NTK_OK and got == data, trial


def test_inflater_under_sanitizers(tmp_path):
    """ntk_pgzip.cpp under AddressSanitizer + UBSan (tools/fuzz_gunzip.cpp): 250 streams - FASTQ-like text, runs, random bytes, long-range
    repeats, every zlib level / strategy, flush points, several members - valid, truncated, bit-flipped or with a piece cut out / doubled,
    at 1..8 threads.  A valid stream comes back byte for byte; nothing crashes, reads out of bounds or hangs."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_gunzip")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                        os.path.join(ROOT, "tools", "fuzz_gunzip.cpp"), "-lz", "-ldl", "-lpthread"], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, "250"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz_gunzip ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
