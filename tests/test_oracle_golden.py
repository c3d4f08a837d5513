"""Pins the CPU oracle (oracle/ntk_oracle.c) against every known-answer vector the reference holds for
the hot path.  Each test cites the reference file:line the literals come from.
The reference crate itself cannot be built here (no rustc); these literals are the pin."""
import os

import numpy as np
import pytest

import oracle as O
from _fastx import fasta_raw_seqs, fastq_raw_seqs


# ---- src/sequence.rs:311-375 ---------------------------------------------------------------

def test_normalize_kats():
    # src/sequence.rs:316-344
    assert O.normalize(b"ACGTU", False) == (b"ACGTT", True)
    assert O.normalize(b"acgtu", False) == (b"ACGTT", True)
    assert O.normalize(b"N.N-N~N N", False) == (b"N-N-N-NN", True)
    assert O.normalize(b"BDHVRYSWKM", True) == (b"BDHVRYSWKM", False)  # None
    assert O.normalize(b"bdhvryswkm", True) == (b"BDHVRYSWKM", True)
    assert O.normalize(b"BDHVRYSWKM", False) == (b"NNNNNNNNNN", True)
    assert O.normalize(b"bdhvryswkm", False) == (b"NNNNNNNNNN", True)
    # doc-test src/sequence.rs:219-224
    assert O.normalize(b"ADGH", False)[0] == b"ANGN"
    assert O.normalize(b"ADGH", True)[0] == b"ADGH"
    assert O.normalize(b"ACGU", True)[0] == b"ACGT"
    # unchanged input reports None
    assert O.normalize(b"ACGTN-", False) == (b"ACGTN-", False)


def test_normalize_python_literals():
    # test_python.py:101-139 (normalize_seq) and :36-41 (Record.normalize)
    n = lambda s, iupac=False: O.normalize(s.encode(), iupac)[0].decode()
    assert n("ACGTU") == "ACGTT"
    assert n("acgtu") == "ACGTT"
    assert n("BDHVRYSWKM") == "NNNNNNNNNN"
    assert n("BDHVRYSWKM", True) == "BDHVRYSWKM"
    assert n("bdhvryswkm", True) == "BDHVRYSWKM"
    assert n("N-N-N-N") == "N-N-N-N"
    assert n("N.N.N.N") == "N-N-N-N"
    assert n("N~N~N~N") == "N-N-N-N"
    for ws in " \t\n\r":
        assert n(ws.join("NNNN")) == "NNNN"
    for junk in "!@#$%^&*|":
        assert n(junk.join("NNNN")) == "NNNNNNN"
    assert n("N9N5N1N") == "NNNNNNN"
    assert n("AGCTGYrtcga", True) == "AGCTGYRTCGA"
    assert n("AGCTGYRTCGA") == "AGCTGNNTCGA"


def test_complement_and_revcomp_kats():
    # src/sequence.rs:347-352, :200; test_python.py:143-149
    assert O.complement(ord("a")) == ord("t")
    assert O.complement(ord("c")) == ord("g")
    assert O.complement(ord("g")) == ord("c")
    assert O.complement(ord("n")) == ord("n")
    assert O.reverse_complement(b"AACC") == b"GGTT"
    assert O.reverse_complement(b"atcg") == b"cgat"
    assert O.reverse_complement(b"ATCG") == b"CGAT"
    for c in b"acgn":
        assert O.reverse_complement(bytes([c])) == bytes([O.complement(c)])


def test_canonical_single_kats():
    # src/sequence.rs:354-361
    assert O.canonical(b"A") == b"A"
    assert O.canonical(b"T") == b"A"
    assert O.canonical(b"AAGT") == b"AAGT"
    assert O.canonical(b"ACTT") == b"AAGT"
    assert O.canonical(b"GC") == b"GC"


def test_minimizer_bytes_kat():
    assert O.minimizer(b"ATTTCG", 3) == b"AAA"  # src/sequence.rs:363-367


def test_quality_mask_kat():
    assert O.quality_mask(b"AGCT", b"AAA0", ord("5")) == b"AGCN"  # src/sequence.rs:369-374


def test_strip_returns():
    # src/sequence.rs:165-191 (no literal KAT in the reference; behaviour per the doc comment)
    assert O.strip_returns(b"ACGT") == (b"ACGT", True)
    assert O.strip_returns(b"AC\nGT\r\nAA\r") == (b"ACGTAA", False)
    assert O.strip_returns(b"") == (b"", True)


# ---- src/kmer.rs:132-227 -------------------------------------------------------------------

def test_kmers_kats():
    assert O.kmers(b"AGCT", 1) == [b"A", b"G", b"C", b"T"]
    assert O.kmers(b"AGNCT", 2) == [b"AG", b"GN", b"NC", b"CT"]
    assert O.kmers(b"AC", 2) == [b"AC"]


def test_canonical_kmers_kats():
    seq = b"AGCT"
    got = O.canonical_kmers(seq, O.reverse_complement(seq), 1)
    assert [(k, f) for _, k, f in got] == [(b"A", False), (b"C", True), (b"C", False), (b"A", True)]
    seq = b"AGCTA"
    got = O.canonical_kmers(seq, O.reverse_complement(seq), 2)
    assert [k for _, k, _ in got] == [b"AG", b"GC", b"AG", b"TA"]
    seq = b"AGNTA"
    got = O.canonical_kmers(seq, O.reverse_complement(seq), 2)
    assert [(p, k) for p, k, _ in got] == [(0, b"AG"), (3, b"TA")]


# ---- src/bitkmer.rs:188-297 ----------------------------------------------------------------

def test_bit_kmers_kats():
    assert [v for _, (v, _), _ in O.bit_kmers(b"AGCT", 1, False)] == [0b00, 0b10, 0b01, 0b11]
    assert [v for _, (v, _), _ in O.bit_kmers(b"ACNGT", 2, False)] == [0b0001, 0b1011]
    assert [v for _, (v, _), _ in O.bit_kmers(b"ACNG", 2, False)] == [1]
    assert [v for _, (v, _), _ in O.bit_kmers(b"AC", 2, False)] == [1]
    assert O.bit_kmers(b"ACGTA", 3, False) == [(0, (6, 3), False), (1, (27, 3), False), (2, (44, 3), False)]
    assert O.bit_kmers(b"TA", 3, False) == []


def test_bit_reverse_complement_kats():
    assert O.bit_reverse_complement(0b000000, 3) == 0b111111
    assert O.bit_reverse_complement(0b111111, 3) == 0
    assert O.bit_reverse_complement(0, 4) == 0b11111111
    assert O.bit_reverse_complement(0b00011011, 4) == 0b00011011


def test_bit_minimizer_kats():
    assert O.bit_minimizer(0b001011, 3, 2) == 0b0010
    assert O.bit_minimizer(0b001011, 3, 1) == 0
    assert O.bit_minimizer(0b11000011, 4, 2) == 0
    assert O.bit_minimizer(0b110001, 3, 2) == 1


def test_bytes_bits_roundtrip_kats():
    assert O.bytes_to_bitmer(b"C") == 1
    assert O.bytes_to_bitmer(b"TTA") == 60
    assert O.bytes_to_bitmer(b"AAA") == 0
    assert O.bitmer_to_bytes(1, 1) == b"C"
    assert O.bitmer_to_bytes(60, 3) == b"TTA"
    assert O.bitmer_to_bytes(0, 3) == b"AAA"


# ---- whole-file pins: benches/benchmark.rs:43-44,66-67,151,97; tests/test_stdin.rs:30-31 -------

@pytest.fixture(scope="module")
def recs_28s(golden_dir):
    return fasta_raw_seqs(open(os.path.join(golden_dir, "28S.fasta"), "rb").read())


@pytest.fixture(scope="module")
def recs_fq(golden_dir):
    return fastq_raw_seqs(open(os.path.join(golden_dir, "PRJNA271013_head.fq"), "rb").read())


def test_28s_base_count(recs_28s):
    # benches/benchmark.rs:151,166,180: 738 580 bases
    assert len(recs_28s) == 570
    assert sum(len(O.strip_returns(r)[0]) for r in recs_28s) == 738_580


def test_28s_k31_byte_path(recs_28s):
    # benches/benchmark.rs:32-44: normalize(true) -> reverse_complement -> canonical_kmers(31)
    st = O.reduce_records(recs_28s, 31, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE_IUPAC)
    assert st["n_total"] == 718_007
    assert st["n_fwd"] == 350_983


def test_28s_k31_bit_path(recs_28s):
    # benches/benchmark.rs:55-67: strip_returns -> bit_kmers(31, true)
    st = O.reduce_records(recs_28s, 31, O.PATH_BITS_CANONICAL, O.PRE_STRIP_RETURNS)
    assert st["n_total"] == 718_007
    assert st["n_fwd"] == 350_983
    # DERIVED digests - regenerated here, must match the survey's throw-away restatement
    assert st["sum"] == 0xD59BC15E9CEBAE61
    assert st["xor"] == 0x38734AE440B6263F


def test_28s_readme_program(recs_28s):
    # README.md:17-46 / src/lib.rs:11-38 / examples/stdin_pipe.rs: normalize(false), k=4, count AAAA.
    st = O.reduce_records(recs_28s, 4, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
    assert st["n_total"] == 736_277 and st["n_rc"] == 385_646   # DERIVED, SURVEY.md B.3
    assert st["hist"][0] == 8_108                                 # AAAA count (DERIVED)
    assert int(st["hist"][:256].sum()) == st["n_total"]
    stb = O.reduce_records(recs_28s, 4, O.PATH_BITS_CANONICAL, O.PRE_STRIP_RETURNS)
    assert int((stb["hist"] > 0).sum()) == 136                    # (4^4 + 4^2) / 2 canonical 4-mers
    assert stb["hist"][0] == 8_108
    assert [int(stb["hist"][b]) for b in (128, 224, 2, 64, 3)] == [11_085, 11_012, 10_726, 9_380, 9_143]


def test_stdin_example_pin():
    # tests/test_stdin.rs:30-31: ">id1\nAGTCGTCA" -> 8 bases, 0 AAAAs
    st = O.reduce_records([b"AGTCGTCA"], 4, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
    assert st["hist"][0] == 0 and st["n_total"] == 5


def test_fastq_head_pins(recs_fq):
    # benches/benchmark.rs:97,111,125: 250 000 bases; k=21 values DERIVED
    assert len(recs_fq) == 2000 and sum(len(r) for r in recs_fq) == 250_000
    a = O.reduce_records(recs_fq, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
    b = O.reduce_records(recs_fq, 21, O.PATH_BITS_CANONICAL, O.PRE_NONE)
    for st in (a, b):
        assert st["n_total"] == 209_965 and st["n_rc"] == 103_784
        assert st["sum"] == 0x047AD82A7ED0CABA and st["xor"] == 0x00000368E0AFC3BC
    assert np.array_equal(a["hist"], b["hist"])


# ---- semantics the survey derived (Appendix A.5) ------------------------------------------------

def test_tie_rules_differ_on_palindromes():
    seq = b"AGCT"
    assert O.canonical_kmers(seq, O.reverse_complement(seq), 4) == [(0, b"AGCT", True)]
    assert O.bit_kmers(seq, 4, True) == [(0, (39, 4), False)]


def test_mixed_case_byte_compare():
    seq = b"acgTT"
    got = O.canonical_kmers(seq, O.reverse_complement(seq), 3)
    assert got == [(0, b"acg", False), (1, b"Acg", True), (2, b"AAc", True)]


# ---- literal iterators vs the independent run-length formulation ---------------------------------

def _concat(records):
    buf = bytearray()
    for r in records:
        buf += r + b"\n"
    return bytes(buf)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 15, 16, 17, 21, 31, 32])
def test_fused_matches_literal_on_28s(recs_28s, k):
    norm = [O.normalize(r, False)[0] for r in recs_28s[:60]]
    lit = O.reduce_records(norm, k, O.PATH_BYTES_CANONICAL, O.PRE_NONE)
    fus = O.reduce_fused(_concat(norm), k, True, True, True)
    for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
        assert lit[key] == fus[key], key
    assert np.array_equal(lit["hist"], fus["hist"])
    stripped = [O.strip_returns(r)[0] for r in recs_28s[:60]]
    lit = O.reduce_records(stripped, k, O.PATH_BITS_CANONICAL, O.PRE_NONE)
    fus = O.reduce_fused(_concat(stripped), k, True, False, False)
    for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
        assert lit[key] == fus[key], key
    lit = O.reduce_records(stripped, k, O.PATH_BITS, O.PRE_NONE)
    fus = O.reduce_fused(_concat(stripped), k, False, False, False)
    for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
        assert lit[key] == fus[key], key


def test_fused_matches_literal_random_alphabet():
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"ACGTacgtNnUuRYKMSWBDHV-.*", dtype=np.uint8)
    for trial in range(200):
        n = int(rng.integers(0, 80))
        seq = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        k = int(rng.integers(1, 9))
        lit = O.reduce_records([seq], k, O.PATH_BITS_CANONICAL, O.PRE_NONE)
        fus = O.reduce_fused(seq, k, True, False, False)
        nrm = O.normalize(seq, False)[0]
        lit2 = O.reduce_records([seq], k, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
        fus2 = O.reduce_fused(nrm, k, True, True, True)
        for key in ("n_total", "n_fwd", "n_rc", "sum", "xor"):
            assert lit[key] == fus[key], (trial, key)
            assert lit2[key] == fus2[key], (trial, key)


def test_batch_and_mt_agree(recs_fq):
    buf = np.frombuffer(_concat(recs_fq), dtype=np.uint8)
    lens = np.array([len(r) + 1 for r in recs_fq], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    a = O.reduce_batch(buf, offs, 1, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE)
    b = O.reduce_batch(buf, offs, 1, 21, O.PATH_BYTES_CANONICAL, O.PRE_NORMALIZE, threads=3)
    assert a["n_total"] == b["n_total"] == 209_965
    assert a["sum"] == b["sum"] and a["xor"] == b["xor"] and np.array_equal(a["hist"], b["hist"])


def test_synth_reads_deterministic():
    a = O.synth_reads(0x5EED0002, 0, 64, 150, 1)
    b = O.synth_reads(0x5EED0002, 32, 32, 150, 1)
    assert a.size == 64 * 151 and np.array_equal(a[32 * 151 :], b)
    assert set(np.unique(a)) <= set(b"ACGTN\n")
    assert (a.reshape(64, 151)[:, 150] == 10).all()
    big = O.synth_reads(0x5EED0002, 0, 20000, 150, 1)
    frac_n = (big == ord("N")).mean()
    assert 0.0005 < frac_n < 0.0015
    # SplitMix64 reference value (public test vector: seed 0 first output)
    assert O.splitmix64_at(0, 0) == 0xE220A8397B1DCDAF


def test_minimizer_with_position_restatement_agrees_with_the_oracle():
    """tests/_refs.py minimizer_with_position (what the ntk_minimizer_batch tests check window start and strand against) returns the
    oracle's sequence::minimizer bytes, and its start / strand point at them (reference src/sequence.rs:139-152; literal :363-367)."""
    import numpy as np
    from _refs import minimizer_with_position
    assert minimizer_with_position(b"ATTTCG", 3) == (b"AAA", 2, 1)   # reverse complement CGAAAT, window 2
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGTACGTacgtNURYKM", dtype=np.uint8)
    for _ in range(300):
        rec = bytes(alphabet[rng.integers(0, len(alphabet), int(rng.integers(1, 80)))])
        m = int(rng.integers(1, len(rec) + 1))
        got, start, is_rc = minimizer_with_position(rec, m)
        assert got == O.minimizer(rec, m)
        assert (O.reverse_complement(rec) if is_rc else rec)[start:start + m] == got
