"""Host-side mirror of needletail's entry points (reference src/parser/mod.rs:85-163, src/python.rs:293-340):
`parse_fastx_file` / `parse_fastx_string` give an iterator of records with the reference Python API's shape
(`id`, `seq`, `qual`, `name`, `description`, `is_fasta()`, `is_fastq()`, `normalize()`); parsing runs on the CPU in the
library's C++ reader, everything sequence-related goes to the GPU through the same C ABI."""
from __future__ import annotations

import ctypes as C
import re
from typing import Iterator, Optional

from . import _lib as L
from . import sequence as S

KIND_NAMES = {1: "Io", 2: "UnknownFormat", 3: "InvalidStart", 4: "InvalidSeparator", 5: "UnequalLengths",
              6: "UnexpectedEnd", 7: "EmptyFile"}


class NeedletailError(Exception):
    """reference src/python.rs:20 (NeedletailError); carries the ParseErrorKind name, line and record id."""

    def __init__(self, kind: int, msg: str, line: int, record_id: str):
        self.kind = KIND_NAMES.get(kind, str(kind))
        self.line = line
        self.record_id = record_id
        where = (f"record '{record_id}' at " if record_id else "") + f"line {line}"
        super().__init__(f"{msg} ({self.kind} at {where})")


# Rust's char::is_whitespace (the Unicode White_Space property; reference src/python.rs:148-163 splits and trims on it).
# NOT Python's \s / str.strip(): those also treat U+001C..U+001F as whitespace, which Rust does not.
_WS_CHARS = "\t\n\x0b\x0c\r \x85\xa0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000"
_WS = re.compile("[" + _WS_CHARS + "]")


class Record:
    """reference src/python.rs:100-290 / needletail.pyi: id, seq (line breaks stripped), qual, name, description."""

    def __init__(self, id: str, seq: str, qual: Optional[str] = None, raw_seq: bytes = None, line: int = 0,
                 num_bases: int = None, byte: int = 0, line_ending: str = "\n"):
        if qual is not None and len(qual) != len(seq):   # reference src/python.rs:205-216
            raise ValueError("Sequence and quality strings must have the same length")
        self.id = id
        self.seq = seq
        self.qual = qual
        self.raw_seq = raw_seq if raw_seq is not None else seq.encode()
        self.line = line
        self.num_bases = len(seq) if num_bases is None else num_bases
        self.byte = byte                  # SequenceRecord::position().byte()  (reference src/parser/record.rs:147-149)
        self.line_ending = line_ending    # SequenceRecord::line_ending()      (reference src/parser/record.rs:152-154)

    # reference src/python.rs:218-268
    def __eq__(self, other):
        return isinstance(other, Record) and (self.id, self.seq, self.qual) == (other.id, other.seq, other.qual)

    def __hash__(self):
        return hash((self.id, self.seq)) if self.qual is None else hash((self.id, self.seq, self.qual))

    def __len__(self):
        return len(self.seq)

    def __str__(self):
        return f">{self.id}\n{self.seq}\n" if self.qual is None else f"@{self.id}\n{self.seq}\n+\n{self.qual}\n"

    # reference src/python.rs:148-163: the id is split at its first whitespace character (char::is_whitespace, so tabs
    # too); the description is what follows with its leading whitespace trimmed, None when there is no whitespace
    @property
    def name(self) -> str:
        m = _WS.search(self.id)
        return self.id[: m.start()] if m else self.id

    @property
    def description(self) -> Optional[str]:
        m = _WS.search(self.id)
        return self.id[m.start():].lstrip(_WS_CHARS) if m else None

    def is_fasta(self) -> bool:
        return self.qual is None

    def is_fastq(self) -> bool:
        return self.qual is not None

    def normalize(self, iupac: bool = False) -> None:
        self.seq = S.normalize_seq(self.seq, iupac)

    def write(self, writer, forced_line_ending: Optional[str] = None) -> None:
        """SequenceRecord::write (reference src/parser/record.rs:156-179): the record back to a binary writer, with its own
        line ending unless one is forced ("\n" or "\r\n")."""
        ending = forced_line_ending or self.line_ending
        if self.qual is None:
            write_fasta(self.id.encode(), self.raw_seq, writer, ending)
        else:
            write_fastq(self.id.encode(), self.raw_seq, self.qual.encode(), writer, ending)

    def __repr__(self):
        def snippet(x, max_len=20):   # reference src/python.rs:37-45
            return x[: max_len - 4] + "\u2026" + x[-3:] if len(x) > max_len else x
        name = self.name
        id_snippet = name if name == self.id else name + "\u2026"
        return f"Record(id={id_snippet}, seq={snippet(self.seq)}, qual={snippet(self.qual) if self.qual is not None else 'None'})"


# (outside SURVEY.md section 8 - record writers / header and quality helpers of the reference's surface, SURVEY section 2 rows 5 and 11: host-side
# conveniences kept for callers of the Python facade; nothing on the hot path uses them and no further surface of this kind is added)
def write_fasta(id: bytes, seq: bytes, writer, line_ending: str = "\n") -> None:
    """reference src/parser/record.rs:207-220"""
    e = line_ending.encode()
    writer.write(b">" + id + e + seq + e)


def write_fastq(id: bytes, seq: bytes, qual: Optional[bytes], writer, line_ending: str = "\n") -> None:
    """reference src/parser/record.rs:222-247: a missing quality line is written as 'I' per base."""
    e = line_ending.encode()
    writer.write(b"@" + id + e + seq + e + b"+" + e + (qual if qual is not None else b"I" * len(seq)) + e)


class FastxReader:
    """Iterator over records (FastxReader::next, reference src/parser/utils.rs:119-130)."""

    def __init__(self, path: str = None, data: bytes = None):
        self._h = C.c_void_p()
        self._keep = data
        if path is not None:
            rc = L.lib().ntk_reader_open_file(str(path).encode(), C.byref(self._h))
        else:
            rc = L.lib().ntk_reader_open_memory(data, len(data), C.byref(self._h))
        if rc != L.NTK_OK:
            err = self._error() if self._h else None
            self.close()
            if err is not None and rc == 8:
                raise err
            L.check(rc, "ntk_reader_open")

    def _error(self) -> NeedletailError:
        kind, line = C.c_int(0), C.c_uint64(0)
        msg, rid = C.create_string_buffer(512), C.create_string_buffer(256)
        L.lib().ntk_reader_error(self._h, C.byref(kind), C.byref(line), msg, 512, rid, 256)
        return NeedletailError(kind.value, msg.value.decode(errors="replace"), line.value, rid.value.decode(errors="replace"))

    def next_raw(self):
        """(id, raw_seq, qual, line, num_bases) as bytes, or None at the end."""
        rec = L.Record()
        rc = L.lib().ntk_reader_next(self._h, C.byref(rec))
        if rc == 100:
            return None
        if rc == 8:
            raise self._error()
        L.check(rc, "ntk_reader_next")
        rid = C.string_at(rec.id, rec.id_len)
        seq = C.string_at(rec.seq, rec.seq_len)
        qual = C.string_at(rec.qual, rec.qual_len) if rec.qual else None
        self._last = (rec.byte, "\r\n" if rec.line_ending == 2 else "\n")
        return rid, seq, qual, rec.line, rec.num_bases

    def __iter__(self) -> Iterator[Record]:
        return self

    def __next__(self) -> Record:
        r = self.next_raw()
        if r is None:
            raise StopIteration
        rid, raw, qual, line, nb = r
        # SequenceRecord::seq(): raw_seq minus CR/LF (reference src/parser/record.rs:85-92, fasta.rs:65-99)
        seq = raw.replace(b"\n", b"").replace(b"\r", b"") if qual is None else raw
        return Record(rid.decode("utf-8", "replace"), seq.decode("utf-8", "replace"),
                      qual.decode("utf-8", "replace") if qual is not None else None, raw, line, nb, *self._last)

    def position(self):
        """FastxReader::position (reference src/parser/utils.rs:125-126): (line, byte) of the record handed out last."""
        line, byte = C.c_uint64(0), C.c_uint64(0)
        L.check(L.lib().ntk_reader_position(self._h, C.byref(line), C.byref(byte), None), "ntk_reader_position")
        return line.value, byte.value

    def line_ending(self):
        """FastxReader::line_ending (reference src/parser/utils.rs:127-130): None before the first record."""
        e = C.c_int(0)
        L.check(L.lib().ntk_reader_position(self._h, None, None, C.byref(e)), "ntk_reader_position")
        return {0: None, 1: "\n", 2: "\r\n"}[e.value]

    def close(self):
        if self._h:
            L.lib().ntk_reader_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_fastx_file(path) -> FastxReader:
    """needletail.parse_fastx_file (reference src/parser/mod.rs:161, src/python.rs:293)."""
    return FastxReader(path=path)


# (outside SURVEY.md section 8 - record writers / header and quality helpers of the reference's surface, SURVEY section 2 rows 5 and 11: host-side
# conveniences kept for callers of the Python facade; nothing on the hot path uses them and no further surface of this kind is added)
def decode_phred(qual: str, base_64: bool = False) -> tuple:
    """needletail.decode_phred (reference src/python.rs:416-427, src/quality.rs:10-28): quality characters minus the
    offset (33, or 64 with base_64); a character below the offset is a ValueError."""
    offset = 64 if base_64 else 33
    out = []
    for ch in qual.encode("latin-1", "replace"):
        if ch < offset:
            raise ValueError(f"Invalid Phred quality: character {ch} is below the offset {offset}")
        out.append(ch - offset)
    return tuple(out)


def parse_fastx_stdin() -> FastxReader:
    """needletail's parse_fastx_stdin (reference src/parser/mod.rs:154-159)."""
    return FastxReader(path="-")


def parse_fastx_string(content) -> FastxReader:
    """needletail.parse_fastx_string (reference src/python.rs:326)."""
    return FastxReader(data=content.encode() if isinstance(content, str) else bytes(content))


def scan_file_parallel(ctx, path, k: int, path_kind: int, pre: int, threads: int = 0, batch_bytes: int = 16 << 20, w: int = 0,
                       data: bytes = None, quality_cutoff: int = 0, streaming_fallback: bool = True) -> dict:
    """scan_file with parser threads that take pieces of the text on demand.  Plain FASTA/FASTQ directly; a gzip file is inflated by all
    threads WHILE the parsers consume the text (bounded memory, any size; out["gzip"] = what the front-end did: route, peak backlog, time of
    the first batch).  `data` scans an in-memory plain-text buffer instead of a path."""
    import os
    from .engine import result_to_dict  # noqa: F401
    threads = threads or min(os.cpu_count() or 1, 32)  # measured best 16-32 on a 256-thread host (tools/pipeline_bench.py)
    ctx.accum_reset()
    p = L.Params(k, path_kind, pre, L.flags(w, quality_cutoff))
    nrec, nb = C.c_uint64(0), C.c_uint64(0)
    if data is not None:
        rc = L.lib().ntk_scan_buffer_parallel(ctx._h, data, len(data), C.byref(p), batch_bytes, threads, C.byref(nrec), C.byref(nb))
    else:
        rc = L.lib().ntk_scan_file_parallel(ctx._h, str(path).encode(), C.byref(p), batch_bytes, threads, C.byref(nrec), C.byref(nb))
        if rc == 6 and streaming_fallback:
            # a gzip file that cannot be inflated into memory (no libdeflate, or beyond the limit): the streaming reader
            return scan_file(ctx, path, k, path_kind, pre, batch_bytes=batch_bytes, w=w, quality_cutoff=quality_cutoff)
    L.check(rc, "ntk_scan_file_parallel")
    out = ctx.accum_read()
    out["n_records"], out["n_bases"] = int(nrec.value), int(nb.value)
    if data is None:
        info = L.GunzipInfo()
        L.lib().ntk_scan_file_info(C.byref(info))
        out["gzip"] = info.as_dict()
    return out


def scan_file(ctx, path, k: int, path_kind: int, pre: int, batch_bytes: int = 64 << 20, n_batches: int = 3, w: int = 0,
              quality_cutoff: int = 0) -> dict:
    """The README program on the GPU: parse -> pinned batches -> overlapped H2D + scan; returns the reduced result plus
    n_records / n_bases (reference src/lib.rs:15-35).  quality_cutoff > 0 masks FASTQ bases below that raw quality byte
    first (QualitySequence::quality_mask, reference src/sequence.rs:285-296)."""
    from .engine import result_to_dict
    rd = FastxReader(path=path)
    try:
        ctx.accum_reset()
        p = L.Params(k, path_kind, pre, L.flags(w, quality_cutoff))
        nrec, nb = C.c_uint64(0), C.c_uint64(0)
        rc = L.lib().ntk_scan_reader(ctx._h, rd._h, C.byref(p), batch_bytes, n_batches, C.byref(nrec), C.byref(nb))
        if rc == 8:
            raise rd._error()
        L.check(rc, "ntk_scan_reader")
        out = ctx.accum_read()
        out["n_records"], out["n_bases"] = int(nrec.value), int(nb.value)
        return out
    finally:
        rd.close()
