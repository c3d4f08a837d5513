"""The reference's Python test-suite cases for the parts of its module that need no device (reference test_python.py:
RecordClassTestCase :19-97, DecodePhredTestCase :150-168, parsing/erroring cases :170-226), restated against this repo's
Python mirror.  normalize_seq / reverse_complement / Record.normalize run on the GPU: tests/test_gpu_parity.py."""
import os
import pathlib

import pytest

import needletail_amd as nt
from needletail_amd import Record, decode_phred


def test_record_fields_and_properties():
    r = Record("test description", "AGCTGATCGA")
    assert (r.id, r.seq, r.qual) == ("test description", "AGCTGATCGA", None)
    assert (r.name, r.description) == ("test", "description")
    # reference src/python.rs:148-163: split at the FIRST whitespace of any kind, description left-trimmed, None without one
    assert (Record("id\tdesc here", "A").name, Record("id\tdesc here", "A").description) == ("id", "desc here")
    assert (Record("x  y", "A").name, Record("x  y", "A").description) == ("x", "y")
    assert (Record("solo", "A").name, Record("solo", "A").description) == ("solo", None)
    assert (Record("trail ", "A").name, Record("trail ", "A").description) == ("trail", "")
    assert (Record("", "A").name, Record("", "A").description) == ("", None)
    # char::is_whitespace is the Unicode White_Space property: U+001C..U+001F (Python's \s / str.strip() treat them as
    # whitespace) are NOT in it, U+0085 / U+00A0 / U+3000 are
    assert (Record("a\x1cb c", "A").name, Record("a\x1cb c", "A").description) == ("a\x1cb", "c")
    assert (Record("a \x1fb", "A").name, Record("a \x1fb", "A").description) == ("a", "\x1fb")
    assert (Record("a\u3000\x85 b", "A").name, Record("a\u3000\x85 b", "A").description) == ("a", "b")
    assert r.is_fasta() and not r.is_fastq()
    q = Record("test description", "AGCTGATCGA", ";**9;;????")
    assert q.qual == ";**9;;????" and q.is_fastq() and not q.is_fasta()
    with pytest.raises(ValueError):
        Record("x", "ACGT", "II")


def test_record_eq_hash_len_str_repr():
    a, b = Record("test", "AGCTGATCGA", ";**9;;????"), Record("test", "AGCTGATCGA", ";**9;;????")
    others = [Record("test2", "AGCTGATCGA", ";**9;;????"), Record("test", "TCGATCAGCT", ";**9;;????"),
              Record("test", "AGCTGATCGA", "????;**9;;"), Record("test", "AGCTGATCGA")]
    assert a == b and hash(a) == hash(b)
    for o in others:
        assert a != o and hash(a) != hash(o)
    assert hash(Record("test", "AGCTGATCGA")) == hash(Record("test", "AGCTGATCGA"))
    assert len(Record("test", "AGCTGATCGA")) == 10
    assert str(Record("test", "AGCTGATCGA")) == ">test\nAGCTGATCGA\n"
    assert str(a) == "@test\nAGCTGATCGA\n+\n;**9;;????\n"
    assert repr(Record("test", "AGCTGATCGAAGCTGATCGAA")) == "Record(id=test, seq=AGCTGATCGAAGCTGA…GAA, qual=None)"
    assert repr(Record("test", "AGCTGATCGAAGCTGATCGAA", ";**9;;????;**9;;????;")) == \
        "Record(id=test, seq=AGCTGATCGAAGCTGA…GAA, qual=;**9;;????;**9;;…??;)"


def test_decode_phred():
    want = (2, 27, 14, 27, 14, 33, 33, 37, 37, 37, 33, 37, 27)
    assert decode_phred("#</</BBFFFBF<") == want
    assert decode_phred("B[N[Naaeeeae[", base_64=True) == want
    assert decode_phred("") == ()
    with pytest.raises(ValueError):
        decode_phred("#</</BBFFFBF ")
    with pytest.raises(ValueError):
        decode_phred("B[N[Naaeeeae?", base_64=True)


def test_parse_string_file_and_pathlib(golden_dir, tmp_path):
    fa = tmp_path / "test.fa"
    fa.write_text(">test\nAGCT\nGATCGA\n>test2\nTAGC\n")
    fq = tmp_path / "test.fq"
    fq.write_text("@EAS54_6_R1_2_1_413_324\nCCCTTCTTGTCTTCAGCGTTTCTCC\n+\n;;3;;;;;;;;;;;;7;;;;;;;88\n"
                  "@EAS54_6_R1_2_1_540_792\nTTGGCAGGCCAAGGCCGATGGATCA\n+\n;;;;;;;;;;;7;;;;;-;;;3;83\n")
    for reader in (nt.parse_fastx_string(fa.read_text()), nt.parse_fastx_file(str(fa)), nt.parse_fastx_file(pathlib.Path(fa))):
        recs = list(reader)
        assert [(r.id, r.seq, r.qual) for r in recs] == [("test", "AGCTGATCGA", None), ("test2", "TAGC", None)]
    for reader in (nt.parse_fastx_string(fq.read_text()), nt.parse_fastx_file(str(fq))):
        recs = list(reader)
        assert (recs[0].id, recs[0].seq, recs[0].qual) == ("EAS54_6_R1_2_1_413_324", "CCCTTCTTGTCTTCAGCGTTTCTCC", ";;3;;;;;;;;;;;;7;;;;;;;88")
        assert (recs[1].id, recs[1].seq, recs[1].qual) == ("EAS54_6_R1_2_1_540_792", "TTGGCAGGCCAAGGCCGATGGATCA", ";;;;;;;;;;;7;;;;;-;;;3;83")


def test_errors():
    with pytest.raises(nt.NeedletailError):
        nt.parse_fastx_file("hey")
    with pytest.raises(nt.NeedletailError):
        for _ in nt.parse_fastx_string("Not a valid file"):
            pass
