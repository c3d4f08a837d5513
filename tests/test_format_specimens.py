"""The reference's conformance corpus run (reference tests/format_specimens.rs:21-101) restated over this repo's
FastxReader: every `valid` specimen of tests/golden/specimen/{FASTA,FASTQ}/index.toml must parse to the end without an
error and every `invalid` FASTQ must raise, with exactly the reference test's own skip list.  The specimen files are data
fixtures copied from the reference's tests/specimen (BioJulia FormatSpecimens; licences in specimen/LICENSE.md).

On top of the accept/reject verdicts (all the reference asserts), each accepted file's records are compared with an
independent line-based splitter written here, so that "parses" also means "parses to the right records"."""
import os

import pytest
import tomli

import needletail_amd as nt

SPEC = os.path.join(os.path.dirname(__file__), "golden", "specimen")


def _index(kind):
    with open(os.path.join(SPEC, kind, "index.toml"), "rb") as f:
        return tomli.load(f)


def _drain(path):
    return [(r.id, r.seq, r.qual) for r in nt.parse_fastx_file(path)]


# reference tests/format_specimens.rs:34-42 — FASTA files tagged "comments" are skipped
FASTA_VALID = [t["filename"] for t in _index("FASTA")["valid"] if "comments" not in (t.get("tags") or [])]
# reference tests/format_specimens.rs:55-62 — line-wrapped quality strings are skipped
FASTQ_SKIP_VALID = {"wrapping_original_sanger.fastq", "longreads_original_sanger.fastq", "tricky.fastq"}
FASTQ_VALID = [t["filename"] for t in _index("FASTQ")["valid"] if t["filename"] not in FASTQ_SKIP_VALID]
# reference tests/format_specimens.rs:73-87 — id mismatch and quality-alphabet checks are not enforced
FASTQ_INVALID = [t["filename"] for t in (_index("FASTQ").get("invalid") or [])
                 if t["filename"] != "error_diff_ids.fastq" and not t["filename"].startswith("error_qual_")
                 and t["filename"] not in ("error_spaces.fastq", "error_tabs.fastq")]


def test_corpus_is_complete():
    assert len(FASTA_VALID) >= 30 and len(FASTQ_VALID) >= 30 and len(FASTQ_INVALID) >= 10
    for kind, names in (("FASTA", FASTA_VALID), ("FASTQ", FASTQ_VALID + FASTQ_INVALID)):
        for n in names:
            assert os.path.isfile(os.path.join(SPEC, kind, n)), n


def _split_fasta(data):
    """Independent splitter: records start at a '>' that begins a line; id = rest of that line, seq = the other lines
    joined (what Record.seq reports: line breaks removed)."""
    recs, cur = [], None
    for line in data.split(b"\n"):
        if line.endswith(b"\r"):
            line = line[:-1]
        if line.startswith(b">"):
            cur = [line[1:], []]
            recs.append(cur)
        elif cur is not None:
            cur[1].append(line)
    return [(i.decode("latin-1"), b"".join(s).decode("latin-1")) for i, s in recs]


@pytest.mark.parametrize("name", FASTA_VALID)
def test_specimen_fasta_valid(name):
    path = os.path.join(SPEC, "FASTA", name)
    got = _drain(path)
    data = open(path, "rb").read()
    want = _split_fasta(data)
    assert len(got) == len(want) > 0
    for (gid, gseq, gqual), (wid, wseq) in zip(got, want):
        assert gqual is None
        assert gid == wid
        assert gseq.replace("\r", "") == wseq


@pytest.mark.parametrize("name", FASTQ_VALID)
def test_specimen_fastq_valid(name):
    path = os.path.join(SPEC, "FASTQ", name)
    got = _drain(path)
    lines = open(path, "rb").read().split(b"\n")
    lines = [l[:-1] if l.endswith(b"\r") else l for l in lines]
    while lines and lines[-1] == b"":
        lines.pop()
    assert len(lines) % 4 == 0 and len(got) == len(lines) // 4
    for i, (gid, gseq, gqual) in enumerate(got):
        h, s, p, q = lines[4 * i: 4 * i + 4]
        assert h[:1] == b"@" and p[:1] == b"+"
        assert (gid, gseq, gqual) == (h[1:].decode("latin-1"), s.decode("latin-1"), q.decode("latin-1"))


@pytest.mark.parametrize("name", FASTQ_INVALID)
def test_specimen_fastq_invalid(name):
    with pytest.raises(nt.NeedletailError):
        _drain(os.path.join(SPEC, "FASTQ", name))
