"""The CPU producer (needletail_amd/csrc/ntk_fastx.cpp) against the reference reader's own unit tests, restated
(reference src/parser/fasta.rs:378-483, src/parser/fastq.rs:460-629, src/parser/mod.rs:169-254, tests/test_compressed.rs)
and against the reference's data files.  No GPU needed: parsing stays on the CPU."""
import gzip
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import needletail_amd as nt
from _fastx import fasta_raw_seqs, fastq_raw_seqs


def recs(data: bytes):
    rd = nt.parse_fastx_string(data)
    out = []
    while True:
        r = rd.next_raw()
        if r is None:
            return out
        out.append(r)


def kinds(data: bytes):
    """(n_ok_records, error kind or None)"""
    n = 0
    try:
        rd = nt.parse_fastx_string(data)
        while rd.next_raw() is not None:
            n += 1
    except nt.NeedletailError as e:
        return n, e.kind
    return n, None


# ---- FASTA: reference src/parser/fasta.rs:388-482 ----------------------------------------------

def test_fasta_basic():
    r = recs(b">test\nACGT\n>test2\nTGCA\n")
    assert [(x[0], x[1]) for x in r] == [(b"test", b"ACGT"), (b"test2", b"TGCA")]


def test_wrapped_fasta():
    r = recs(b">test\nACGT\nACGT\n>test2\nTGCA\nTG")
    assert [(x[0], x[1], x[4]) for x in r] == [(b"test", b"ACGT\nACGT", 8), (b"test2", b"TGCA\nTG", 6)]


def test_wrapped_fasta_windows_newlines():
    r = recs(b">test\r\nACGT\r\nACGT\r\n>test2\r\nTGCA\r\nTG")
    assert [(x[0], x[1], x[4], x[3]) for x in r] == [(b"test", b"ACGT\r\nACGT", 8, 1), (b"test2", b"TGCA\r\nTG", 6, 4)]


def test_fasta_premature_ending():
    assert kinds(b">test\nAGCT\n>test2") == (1, "UnexpectedEnd")
    assert kinds(b">test\r\nAGCT\r\n>test2\r\n") == (1, "UnexpectedEnd")


def test_fasta_empty_records():
    for data in (b">\n\n>shine\nAGGAGGU", b">\r\n\r\n>shine\r\nAGGAGGU"):
        r = recs(data)
        assert [(x[0], x[1]) for x in r] == [(b"", b""), (b"shine", b"AGGAGGU")]


# ---- FASTQ: reference src/parser/fastq.rs:473-628 ------------------------------------------------

def test_simple_fastq_both_line_endings():
    for data in (b"@test\nAGCT\n+test\n~~a!\n@test2\nTGCA\n+test\nWUI9",
                 b"@test\r\nAGCT\r\n+test\r\n~~a!\r\n@test2\r\nTGCA\r\n+test\r\nWUI9"):
        r = recs(data)
        assert [(x[0], x[1], x[2]) for x in r] == [(b"test", b"AGCT", b"~~a!"), (b"test2", b"TGCA", b"WUI9")]


def test_fastq_eof_cases():
    assert kinds(b"@test\nACGT\n+\nIII") == (0, "UnequalLengths")             # test_eof_in_qual
    assert kinds(b"@test\nAGCT\n+test\n~~a!\n@test2\nTGCA") == (1, "UnexpectedEnd")   # test_eof_in_seq
    assert kinds(b"@test\nAGCT\n+test\n~~a!\n\n") == (1, None)                 # extra empty newlines are ok
    assert kinds(b"@test\nAGCT\n+test\n~~a!\n\n@TEST\nA\n+TEST\n~") == (1, "InvalidStart")
    assert kinds(b"@test\nAGCT\n+\nIII\n@TEST\nA\n+\nI") == (0, "UnequalLengths")   # test_mismatched_lengths


def test_fastq_empty_records_and_line_numbers():
    r = recs(b"@\n\n+\n\n@test2\nTGCA\n+test2\n~~~~\n")
    assert [(x[0], x[1], x[2]) for x in r] == [(b"", b"", b""), (b"test2", b"TGCA", b"~~~~")]
    s = b"ACGTACGATCGTACGTAGCTGCTAGCTAGCATGCATGACACACACGTACGATCGTACGTAGCTGCTAGCTAGCATGCATGACACAC"
    q = b"0" * len(s)
    data = b"@NCBI actually has files like this\n" + s + b"\n+\n" + q + b"\n@NCBI actually has files like this\n\n+\n\n" \
           b"@NCBI actually has files like this\n" + s + b"\n+\n" + q
    assert [x[3] for x in recs(data)] == [1, 5, 9]   # start_line_number, test_weird_ncbi_file


def test_reference_bad_files(golden_dir):
    def file_kinds(name):
        n = 0
        try:
            for _ in nt.parse_fastx_file(os.path.join(golden_dir, name)):
                n += 1
        except nt.NeedletailError as e:
            return n, e.kind
        return n, None
    assert file_kinds("bad_header.fastq") == (1, "UnexpectedEnd")       # fastq.rs:604-614
    assert file_kinds("random_tsv.fq") == (1, "InvalidSeparator")        # fastq.rs:617-627


# ---- entry point: reference src/parser/mod.rs:185-210, tests/test_compressed.rs --------------------

def test_sniffing_errors():
    assert kinds(b"") == (0, "EmptyFile")
    assert kinds(b"@") == (0, "EmptyFile")
    assert kinds(gzip.compress(b"")) == (0, "EmptyFile")
    assert kinds(b"hello") == (0, "UnknownFormat")
    with pytest.raises(nt.NeedletailError):
        nt.parse_fastx_file("/nonexistent/file.fa")


def test_gzip_roundtrip_and_multi_member(golden_dir):
    plain = open(os.path.join(golden_dir, "test.fa"), "rb").read()
    want = [(b"test", b"AGCTGATCGA"), (b"test2", b"TAGC")]
    assert [(x[0], x[1]) for x in recs(plain)] == want
    assert [(r.id.encode(), r.seq.encode()) for r in nt.parse_fastx_file(os.path.join(golden_dir, "test.fa.gz"))] == want
    # concatenated members decode as one stream (MultiGzDecoder)
    two = gzip.compress(b">a\nACGT\n") + gzip.compress(b">b\nTTTT\nGG\n")
    assert [(x[0], x[1]) for x in recs(two)] == [(b"a", b"ACGT"), (b"b", b"TTTT\nGG")]


def test_python_record_shape():
    r = list(nt.parse_fastx_string(">test description here\nAGCT\nGA\n@bad"))  # '@' inside FASTA is sequence text
    assert r[0].id == "test description here" and r[0].name == "test" and r[0].description == "description here"
    assert r[0].seq == "AGCTGA@bad" and r[0].qual is None and r[0].is_fasta()
    q = list(nt.parse_fastx_string("@r1\nACGT\n+\nIIII\n"))[0]
    assert q.is_fastq() and q.qual == "IIII" and q.seq == "ACGT"


# ---- whole files + buffer-boundary stress against an independent splitter -------------------------

def test_golden_files_match_independent_splitter(golden_dir):
    data = open(os.path.join(golden_dir, "28S.fasta"), "rb").read()
    got = [x[1] for x in recs(data)]
    assert got == fasta_raw_seqs(data) and len(got) == 570
    assert sum(x[4] for x in recs(data)) == 738_580       # benches/benchmark.rs:151
    data = open(os.path.join(golden_dir, "PRJNA271013_head.fq"), "rb").read()
    got = [x[1] for x in recs(data)]
    assert got == fastq_raw_seqs(data) and sum(len(s) for s in got) == 250_000   # benches/benchmark.rs:97
    assert [x[1] for x in recs(gzip.compress(data))] == got


def test_records_across_buffer_boundaries():
    rng = np.random.default_rng(1)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    # FASTA: records from tiny to far beyond the 64 KiB initial buffer, wrapped at 70 columns, mixed line endings
    parts, want = [], []
    for i, n in enumerate([0, 1, 69, 70, 71, 5000, 65_000, 66_000, 200_000, 3, 1_300_000, 10]):
        seq = bytes(letters[rng.integers(0, 5, n)])
        nl = b"\r\n" if i % 3 == 1 else b"\n"
        body = nl.join(seq[j:j + 70] for j in range(0, len(seq), 70)) if n else b""
        parts.append(b">rec%d some description" % i + nl + body + nl)
        want.append(seq)
    data = b"".join(parts)
    got = recs(data)
    assert [x[1].replace(b"\n", b"").replace(b"\r", b"") for x in got] == want
    assert [x[4] for x in got] == [len(s) for s in want]
    # FASTQ: 20k records whose sizes make record boundaries hit every buffer offset
    parts, want = [], []
    for i in range(20_000):
        n = int(rng.integers(0, 300))
        seq = bytes(letters[rng.integers(0, 5, n)])
        parts.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * n + b"\n")
        want.append(seq)
    got = recs(b"".join(parts))
    assert [x[1] for x in got] == want and [x[3] for x in got] == [1 + 4 * i for i in range(20_000)]


def test_parallel_split_points_are_record_starts():
    """The parallel producer cuts a plain file at record starts; FASTQ quality lines that start with '@' must not fool it."""
    import ctypes as C
    from needletail_amd import _lib as L
    rng = np.random.default_rng(4)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    parts, starts, off = [], set(), 0
    for i in range(3000):
        n = int(rng.integers(1, 120))
        seq = bytes(letters[rng.integers(0, 5, n)])
        qual = bytes(rng.choice(np.frombuffer(b"@+I#5@@", dtype=np.uint8), n))  # many '@' and '+' at line starts
        rec = b"@r%d some text\n" % i + seq + b"\n+\n" + qual + b"\n"
        starts.add(off); off += len(rec); parts.append(rec)
    data = b"".join(parts)
    for pieces in (2, 7, 64, 500):
        cuts = (C.c_uint64 * (pieces + 1))()
        assert L.lib().ntk_fastx_split_points(data, len(data), pieces, cuts) == 0
        cl = list(cuts)
        assert cl[0] == 0 and cl[-1] == len(data) and cl == sorted(cl)
        assert all(c in starts or c == len(data) for c in cl)
    fa = b"".join(b">c%d\n" % i + b"ACGT>ACGT\nAC\n" for i in range(500))   # '>' inside sequence lines is not a start
    cuts = (C.c_uint64 * 9)()
    assert L.lib().ntk_fastx_split_points(fa, len(fa), 8, cuts) == 0
    assert all(fa[c:c + 2] == b">c" or c == len(fa) for c in cuts)


def test_parallel_split_points_skip_bare_headers():
    """A FASTA record without a sequence line must not end a piece: the piece would be a truncated record to
    its reader although the whole file parses."""
    import ctypes as C
    from needletail_amd import _lib as L
    data = b">a\nACGTACGT\n>empty\n>b\nACGT\n>c\nAC\n"
    whole = [(i, s) for i, s, *_ in recs(data)]
    assert whole == [(b"a", b"ACGTACGT"), (b"empty", b""), (b"b", b"ACGT"), (b"c", b"AC")]
    for pieces in (2, 3, 4, 5, 8):
        cuts = (C.c_uint64 * (pieces + 1))()
        assert L.lib().ntk_fastx_split_points(data, len(data), pieces, cuts) == 0
        cl = list(cuts)
        got = []
        for lo, hi in zip(cl[:-1], cl[1:]):
            if hi > lo:
                got += [(i, s) for i, s, *_ in recs(data[lo:hi])]   # every piece parses on its own
        assert got == whole, (pieces, cl)
    rng = np.random.default_rng(9)
    parts = []
    for i in range(400):
        parts.append(b">r%d\n" % i + (b"" if rng.integers(0, 3) == 0 else b"ACGT" * int(rng.integers(1, 9)) + b"\n"))
    big = b"".join(parts) + b">last\nAC\n"
    whole = [(i, s) for i, s, *_ in recs(big)]
    for pieces in (2, 7, 33):
        cuts = (C.c_uint64 * (pieces + 1))()
        assert L.lib().ntk_fastx_split_points(big, len(big), pieces, cuts) == 0
        cl = list(cuts)
        got = []
        for lo, hi in zip(cl[:-1], cl[1:]):
            if hi > lo:
                got += [(i, s) for i, s, *_ in recs(big[lo:hi])]
        assert got == whole


def test_truncated_compressed_streams_are_errors(golden_dir):
    """A compressed stream that ends before its end marker is an Io error, not a shorter file (flate2's
    MultiGzDecoder and the other decoders of the reference return UnexpectedEof, src/parser/mod.rs:95-108)."""
    data = b"".join(b">r%d\n" % i + b"ACGT" * 750 + b"\n" for i in range(20))
    gz = gzip.compress(data)
    assert len(recs(gz)) == 20
    for frac in (0.25, 0.5, 0.9):
        with pytest.raises(nt.NeedletailError) as e:
            recs(gz[: int(len(gz) * frac)])
        assert e.value.kind == "Io"
    with pytest.raises(nt.NeedletailError):
        recs(gz[:-4])          # the length word of the trailer is missing
    two = gzip.compress(b">a\nACGT\n") + gzip.compress(b">b\nTTTT\nGG\n")
    assert len(recs(two)) == 2
    with pytest.raises(nt.NeedletailError):
        recs(two[:-6])
    for ext in ("bz2", "xz", "zst"):
        raw = open(os.path.join(golden_dir, "test.fa." + ext), "rb").read()
        try:
            ok = recs(raw)
        except nt.NtkError:
            continue           # codec library not installed on this host
        assert len(ok) == 2
        with pytest.raises(nt.NeedletailError) as e:
            recs(raw[: len(raw) * 2 // 3])
        assert e.value.kind == "Io", ext


def test_record_position_and_line_number():
    """reference src/parser/record.rs:259-285 (test_start_line_number, test_position)."""
    r = nt.parse_fastx_string("@test\nACGT\n+\nIIII\n@test2\nACGT\n+\nIIII")
    assert [rec.line for rec in r] == [1, 5]
    r = nt.parse_fastx_string("@test1\nACGT\n+\nIIII\n@test222\nACGT\n+\nIIII\n@test3\nACGT\n+\nIIII")
    got = []
    for rec in r:
        got.append(rec.byte)
        assert r.position() == (rec.line, rec.byte)
    assert got == [0, 19, 40]
    # FASTA: byte offsets advance by whole records, lines by the record's line count
    r = nt.parse_fastx_string(">a\nAC\nGT\n>b\nA\n>c\nACGT")
    assert [(rec.line, rec.byte) for rec in r] == [(1, 0), (4, 9), (6, 14)]


def test_reader_line_ending():
    """reference src/parser/fasta.rs:390-441, fastq.rs tests: None before the first next(), then Unix / Windows."""
    r = nt.parse_fastx_string(">test\nACGT\n>test2\nTGCA\n")
    assert r.line_ending() is None
    rec = next(r)
    assert r.line_ending() == "\n" and rec.line_ending == "\n"
    r = nt.parse_fastx_string(">test\r\nACGT\r\nACGT\r\n>test2\r\nTGCA\r\nTG")
    rec = next(r)
    assert rec.raw_seq == b"ACGT\r\nACGT" and rec.num_bases == 8 and rec.line == 1
    assert r.line_ending() == "\r\n" and rec.line_ending == "\r\n"
    r = nt.parse_fastx_string("@test\r\nAGCT\r\n+test\r\n~~a!\r\n")
    rec = next(r)
    assert r.line_ending() == "\r\n" and rec.qual == "~~a!"
    r = nt.parse_fastx_string("@test\nAGCT\n+test\n~~a!")
    next(r)
    assert r.line_ending() == "\n"


def test_stdin_reader(golden_dir):
    """parse_fastx_stdin (reference src/parser/mod.rs:154-159; tests/test_stdin.rs: '>id1\\nAGTCGTCA' piped in -> 8 bases),
    plain and gzip-compressed, in a child process."""
    import gzip
    import subprocess
    import sys
    root = os.path.dirname(golden_dir.rstrip("/")).rsplit("/tests", 1)[0]
    code = ("import sys; sys.path.insert(0, %r); import needletail_amd as nt; "
            "recs = list(nt.parse_fastx_stdin()); print(len(recs), sum(r.num_bases for r in recs), recs[0].id)" % root)
    for payload in (b">id1\nAGTCGTCA", gzip.compress(b">id1\nAGTCGTCA")):
        r = subprocess.run([sys.executable, "-c", code], input=payload, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        assert r.stdout.decode().split() == ["1", "8", "id1"]
    data = open(os.path.join(golden_dir, "28S.fasta"), "rb").read()
    r = subprocess.run([sys.executable, "-c", code], input=data, capture_output=True, timeout=120)
    assert r.stdout.decode().split()[:2] == ["570", "738580"]


def test_record_write_round_trip(golden_dir):
    """SequenceRecord::write / write_fasta / write_fastq (reference src/parser/record.rs:156-247): parse -> write gives the
    input back (own line ending kept, or forced), a record without qualities is written with 'I' per base."""
    import io
    for text in (b">a desc\nACGT\nAC\n>b\nTTTT\n", b">a\r\nACGT\r\nAC\r\n>b\r\nT\r\n", b"@r1\nACGT\n+\nIIII\n@r2 x\nAC\n+\n!!\n",
                 open(os.path.join(golden_dir, "test.fa"), "rb").read()):
        out = io.BytesIO()
        for rec in nt.parse_fastx_string(text):
            rec.write(out)
        assert out.getvalue() == text
    out = io.BytesIO()
    for rec in nt.parse_fastx_string(b">a\nAC\nGT\n"):
        rec.write(out, "\r\n")
    assert out.getvalue() == b">a\r\nAC\nGT\r\n"
    out = io.BytesIO()
    nt.write_fastq(b"id", b"ACGT", None, out)
    assert out.getvalue() == b"@id\nACGT\n+\nIIII\n"


def test_bgzf_through_the_streaming_reader(golden_dir, tmp_path):
    """Block gzip is just concatenated members to the streaming reader (MultiGzDecoder, reference src/parser/mod.rs:95-108)."""
    from _fastx import bgzf_compress
    data = open(os.path.join(golden_dir, "28S.fasta"), "rb").read()
    p = tmp_path / "28S.fasta.gz"
    p.write_bytes(bgzf_compress(data, block=30000))
    recs = list(nt.parse_fastx_file(str(p)))
    assert len(recs) == 570 and sum(r.num_bases for r in recs) == 738_580


def test_compressed_files_are_read_automatically(golden_dir, tmp_path):
    """reference tests/test_compressed.rs:12-38: the same two records from test.fa.{gz,bz2,xz,zst} (the reference's own data
    files), plus larger bzip2 / xz streams written here (several reader-buffer refills, concatenated records)."""
    import bz2
    import lzma
    for ext in ("gz", "bz2", "xz", "zst"):
        recs = list(nt.parse_fastx_file(os.path.join(golden_dir, "test.fa." + ext)))
        assert [(r.id, r.raw_seq, r.qual) for r in recs] == [("test", b"AGCTGATCGA", None), ("test2", b"TAGC", None)], ext
        assert all(r.is_fasta() for r in recs)
    data = open(os.path.join(golden_dir, "28S.fasta"), "rb").read()
    for name, blob in (("x.bz2", bz2.compress(data)), ("x.xz", lzma.compress(data)), ("x9.bz2", bz2.compress(data * 3, 1))):
        p = tmp_path / name
        p.write_bytes(blob)
        recs = list(nt.parse_fastx_file(str(p)))
        mult = 3 if name == "x9.bz2" else 1
        assert len(recs) == 570 * mult and sum(r.num_bases for r in recs) == 738_580 * mult, name
    bad = tmp_path / "bad.bz2"
    blob = bytearray(bz2.compress(data)); blob[len(blob) // 2] ^= 0xFF
    bad.write_bytes(bytes(blob))
    with pytest.raises(nt.NeedletailError):
        list(nt.parse_fastx_file(str(bad)))


# ---- randomised differential test against a pure-Python model of the reference's record rules -------------------

def _model_fasta(data: bytes):
    """reference src/parser/fasta.rs:196-243: a record runs to the last line feed before the next line starting with '>'."""
    out, starts = [], [0] + [i + 1 for i in range(len(data) - 1) if data[i] == 10 and data[i + 1] == ord(">")]
    line = 1
    for a, b in zip(starts, starts[1:] + [len(data)]):
        rec = data[a:b]
        body = rec[:-1] if rec.endswith(b"\n") else rec          # the record's last line end is not part of it
        first = body.find(b"\n")
        if first < 0:
            hdr, seq = body, b""
        else:
            hdr, seq = body[:first], body[first + 1:]
        if hdr.endswith(b"\r"):
            hdr = hdr[:-1]
        if seq.endswith(b"\r"):
            seq = seq[:-1]
        out.append((hdr[1:], seq, None, line, len(seq) - seq.count(b"\n") - seq.count(b"\r")))
        line += rec.count(b"\n") + (0 if rec.endswith(b"\n") else 1)
    return out


def _model_fastq(data: bytes):
    lines = data.split(b"\n")
    out, i = [], 0
    while i < len(lines) and any(x.rstrip(b"\r") for x in lines[i:]):   # only blank lines may follow the last record
        h, s, _, q = [x[:-1] if x.endswith(b"\r") else x for x in (lines[i:i + 4] + [b""] * 4)[:4]]
        out.append((h[1:], s, q, i + 1, len(s)))
        i += 4
    return out


@settings(max_examples=120, deadline=None)
@given(st.data())
def test_reader_matches_model_on_random_files(data):
    rng = np.random.default_rng(data.draw(st.integers(0, 2**32 - 1)))
    fastq = data.draw(st.booleans())
    nl = b"\r\n" if data.draw(st.booleans()) else b"\n"
    n_rec = data.draw(st.integers(1, 40))
    big = data.draw(st.booleans())
    parts = []
    for r in range(n_rec):
        L = int(rng.integers(0, 90_000 if (big and r % 7 == 0) else 300))
        if r == n_rec - 1 and L == 0:
            L = 1   # a FASTA header with nothing after it at the end of the input is a truncated record (UnexpectedEnd), and a
                    # final FASTQ record of empty lines is indistinguishable from trailing blank lines
        seq = bytes(np.frombuffer(b"ACGTNacgt>@+", dtype=np.uint8)[rng.integers(0, 9 if not fastq else 9, L)])
        hdr = b"id%d >odd @hdr + tab\t" % r
        if fastq:
            qual = bytes(np.frombuffer(b"IJ@>+!~", dtype=np.uint8)[rng.integers(0, 7, L)])   # '@', '>' and '+' are legal here
            parts.append(b"@" + hdr + nl + seq + nl + b"+" + (hdr if r % 2 else b"") + nl + qual + nl)
        else:
            w = int(rng.integers(1, 120))
            body = nl.join(seq[j:j + w] for j in range(0, L, w))
            parts.append(b">" + hdr + nl + body + (nl if L else b""))
    text = b"".join(parts)
    tail = data.draw(st.sampled_from(["keep", "strip", "blank"]))
    if tail == "strip" and text.endswith(nl) and not (not fastq and text.endswith(nl) and parts[-1].count(b"\n") == 1):
        text = text[: -len(nl)]
    elif tail == "blank" and fastq:
        text += nl + nl
    model = _model_fastq(text) if fastq else _model_fasta(text)
    got = [(rid, seq, qual, line, nb) for rid, seq, qual, line, nb in recs(text)]
    assert got == model
    # gzip on top must not change anything
    if len(text) < 200_000:
        assert [(a, b, c, d, e) for a, b, c, d, e in recs(gzip.compress(text))] == model


def test_reader_under_sanitizers(tmp_path):
    """The C++ reader (tools/fuzz_reader.cpp includes ntk_fastx.cpp) under AddressSanitizer + UBSan over 6000 random valid,
    truncated, byte-mutated and gzip-wrapped inputs: no memory error, no undefined behaviour, every input either parses or
    is rejected."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_reader")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    if "avx2" in open("/proc/cpuinfo").read():
        flags.append("-mavx2")
    b = subprocess.run(["g++", *flags, "-o", exe, os.path.join(root, "tools", "fuzz_reader.cpp"), "-lz", "-ldl"],
                       capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
