"""Test-only record splitter reproducing what `SequenceRecord::sequence()` hands to the hot path
(reference: src/parser/record.rs:78-83,181-185; src/parser/fasta.rs:55-63; src/parser/fastq.rs:42-44):
FASTA -> raw_seq = everything between the header line's '\n' and the record's last '\n', interior line
breaks included, one trailing '\r' trimmed; FASTQ -> the sequence line, trailing '\r' trimmed."""


def _trim_cr(b: bytes) -> bytes:
    return b[:-1] if b.endswith(b"\r") else b


def fasta_raw_seqs(data: bytes):
    out = []
    assert data[:1] == b">"
    recs = data.split(b"\n>")
    for i, rec in enumerate(recs):
        nl = rec.find(b"\n")
        if nl < 0:
            out.append(b"")
            continue
        body = rec[nl + 1 :]
        if body.endswith(b"\n"):
            body = body[:-1]
        out.append(_trim_cr(body))
    return out


def fastq_raw_seqs(data: bytes):
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    assert len(lines) % 4 == 0
    return [_trim_cr(lines[i + 1]) for i in range(0, len(lines), 4)]


def bgzf_compress(data: bytes, block: int = 60000) -> bytes:
    """Block gzip (BGZF, bgzip/htslib): independent members of <= 64 KiB with their compressed size in a 'BC' extra subfield,
    closed by the empty EOF block.  Test helper (no bgzip binary in the image)."""
    import struct
    import zlib
    out = bytearray()
    for ch in [data[i:i + block] for i in range(0, len(data), block)] + [b""]:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(ch) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch))
    return bytes(out)
