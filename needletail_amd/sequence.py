"""Host-side mirror of needletail's `Sequence` trait (reference src/sequence.rs:156-253) and of the
two hot-path functions its Python module exposes (reference src/python.rs:365-371,391-399), all running on
the HIP engine through the C ABI's compat face.  Names, argument meaning and results follow the reference
so that the parity tests read like the reference's own tests."""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Tuple

import numpy as np

from . import _lib as L
from .engine import Context, default_context


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


def normalize(seq: bytes, iupac: bool = False, ctx: Context = None) -> bytes:
    """Sequence::normalize (reference src/sequence.rs:226-232): the normalised bytes (the input itself when
    nothing changed, like Cow::Borrowed)."""
    out, changed = normalize_opt(seq, iupac, ctx)
    return out if changed else seq


def normalize_opt(seq: bytes, iupac: bool = False, ctx: Context = None) -> Tuple[bytes, bool]:
    """sequence::normalize (reference src/sequence.rs:19-62): (bytes, changed); changed False <=> `None`."""
    c = _ctx(ctx)
    out = C.create_string_buffer(max(len(seq), 1))
    n, ch = C.c_uint64(0), C.c_int(0)
    L.check(L.lib().ntk_normalize(c._h, seq, len(seq), int(iupac), out, C.byref(n), C.byref(ch)), "ntk_normalize")
    return out.raw[: n.value], bool(ch.value)


def strip_returns(seq: bytes, ctx: Context = None) -> bytes:
    """Sequence::strip_returns (reference src/sequence.rs:165-191)."""
    c = _ctx(ctx)
    out = C.create_string_buffer(max(len(seq), 1))
    n, b = C.c_uint64(0), C.c_int(0)
    L.check(L.lib().ntk_strip_returns(c._h, seq, len(seq), out, C.byref(n), C.byref(b)), "ntk_strip_returns")
    return seq if b.value else out.raw[: n.value]


def reverse_complement(seq, ctx: Context = None):
    """Sequence::reverse_complement (reference src/sequence.rs:202-208); str in -> str out like the
    reference's Python function (reference src/python.rs:391-399)."""
    if isinstance(seq, str):
        return reverse_complement(seq.encode("utf-8"), ctx).decode("utf-8")
    c = _ctx(ctx)
    out = C.create_string_buffer(max(len(seq), 1))
    L.check(L.lib().ntk_reverse_complement(c._h, seq, len(seq), out), "ntk_reverse_complement")
    return out.raw[: len(seq)]


def normalize_seq(seq: str, iupac: bool = False, ctx: Context = None) -> str:
    """needletail.normalize_seq (reference src/python.rs:365-371)."""
    return normalize(seq.encode("utf-8"), iupac, ctx).decode("utf-8")


def kmers(seq: bytes, k: int) -> Iterator[bytes]:
    """Sequence::kmers (reference src/kmer.rs:13-41): plain windows, pure slicing (nothing to accelerate)."""
    for i in range(0, len(seq) - k + 1):
        yield seq[i : i + k]


def canonical_kmers_arrays(seq: bytes, k: int, ctx: Context = None):
    c = _ctx(ctx)
    if k < 1 or k > 255:
        raise ValueError("k must be 1..255")
    cap = max(len(seq), 1)
    pos = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    L.check(L.lib().ntk_canonical_kmers(c._h, seq, len(seq), k, pos.ctypes.data, flg.ctypes.data, cap, C.byref(n)),
            "ntk_canonical_kmers")
    return pos[: n.value], flg[: n.value]


def canonical_kmers(seq: bytes, k: int, reverse_complement: bytes, ctx: Context = None) -> List[Tuple[int, bytes, bool]]:
    """Sequence::canonical_kmers(k, &rc) (reference src/sequence.rs:237-239, src/kmer.rs:48-130): items
    (pos, slice, is_rc) with the slice drawn from `seq` or from the caller's `reverse_complement`."""
    rc = reverse_complement
    pos, flg = canonical_kmers_arrays(seq, k, ctx)
    out = []
    for p, f in zip(pos.tolist(), flg.tolist()):
        out.append((p, rc[len(rc) - p - k : len(rc) - p] if f else seq[p : p + k], bool(f)))
    return out


def bit_kmers_arrays(seq: bytes, k: int, canonical: bool, ctx: Context = None):
    c = _ctx(ctx)
    if k < 1 or k > 32:
        raise ValueError("k must be 1..32")
    cap = max(len(seq), 1)
    pos = np.empty(cap, dtype=np.uint64)
    val = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    L.check(L.lib().ntk_bit_kmers(c._h, seq, len(seq), k, int(canonical), pos.ctypes.data, val.ctypes.data,
                                  flg.ctypes.data, cap, C.byref(n)), "ntk_bit_kmers")
    return pos[: n.value], val[: n.value], flg[: n.value]


def bit_kmers(seq: bytes, k: int, canonical: bool, ctx: Context = None) -> List[Tuple[int, Tuple[int, int], bool]]:
    """Sequence::bit_kmers(k, canonical) (reference src/sequence.rs:250-252, src/bitkmer.rs:72-109)."""
    pos, val, flg = bit_kmers_arrays(seq, k, canonical, ctx)
    return [(p, (v, k), bool(f)) for p, v, f in zip(pos.tolist(), val.tolist(), flg.tolist())]


def _pack_records(records):
    offs = np.zeros(len(records) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in records], out=offs[1:])
    return b"".join(records), offs


def bit_kmers_batch(records, k: int, canonical: bool, ctx: Context = None):
    """Sequence::bit_kmers(k, canonical) for a whole batch of records in one device pass (ntk_bit_kmers_batch):
    returns (counts, pos, val, was_rc); record i's items are the slice [counts[:i].sum(), counts[:i+1].sum())."""
    c = _ctx(ctx)
    if k < 1 or k > 32:
        raise ValueError("k must be 1..32")
    seq, offs = _pack_records(records)
    cap = max(int(sum(max(0, len(r) - k + 1) for r in records)), 1)
    counts = np.zeros(max(len(records), 1), dtype=np.uint64)
    pos = np.empty(cap, dtype=np.uint64)
    val = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    L.check(L.lib().ntk_bit_kmers_batch(c._h, seq, offs.ctypes.data, len(records), k, int(canonical), counts.ctypes.data,
                                        pos.ctypes.data, val.ctypes.data, flg.ctypes.data, cap, C.byref(n)), "ntk_bit_kmers_batch")
    return counts[: len(records)], pos[: n.value], val[: n.value], flg[: n.value]


def canonical_kmers_batch(records, k: int, ctx: Context = None):
    """Sequence::canonical_kmers(k, &reverse_complement(record)) for a whole batch of records in one device pass
    (ntk_canonical_kmers_batch): returns (counts, pos, is_rc); the k-mer slices are drawn on the host as
    reference src/kmer.rs:121-128 does."""
    c = _ctx(ctx)
    if k < 1 or k > 255:
        raise ValueError("k must be 1..255")
    seq, offs = _pack_records(records)
    cap = max(int(sum(max(0, len(r) - k + 1) for r in records)), 1)
    counts = np.zeros(max(len(records), 1), dtype=np.uint64)
    pos = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    L.check(L.lib().ntk_canonical_kmers_batch(c._h, seq, offs.ctypes.data, len(records), k, counts.ctypes.data,
                                              pos.ctypes.data, flg.ctypes.data, cap, C.byref(n)), "ntk_canonical_kmers_batch")
    return counts[: len(records)], pos[: n.value], flg[: n.value]


class CanonicalKmersPlanes:
    """The items of Sequence::canonical_kmers(k, &rc) for a batch of records as two bit planes (ntk_canonical_kmers_batch_planes):
    per window start "emitted" and "is_rc".  `iter(i, buffer, rc)` walks record i's bits and yields exactly what the reference
    iterator yields for it - (pos, buffer[pos:pos+k] or the rc slice, is_rc), reference src/kmer.rs:114-129 - and `arrays(i)` the
    (pos, is_rc) arrays.  One quarter of a byte crosses PCIe per sequence byte (ntk_canonical_kmers_batch: nine bytes per item)."""

    def __init__(self, k, lengths, rec_bit, valid16, rc16, total):
        self.k, self.lengths, self.rec_bit, self.valid16, self.rc16, self.total = k, lengths, rec_bit, valid16, rc16, total

    def _bits(self, plane, i):
        n = int(self.lengths[i]) - self.k + 1
        if n <= 0:
            return np.zeros(0, dtype=np.uint8)
        b0 = int(self.rec_bit[i])
        w0, w1 = b0 >> 4, (b0 + n + 15) >> 4
        # bit (15 - b % 16) of word b / 16: big-endian bit order inside each 16-bit word
        bits = np.unpackbits(plane[w0:w1].astype(">u2").view(np.uint8))
        return bits[b0 - 16 * w0: b0 - 16 * w0 + n]

    def arrays(self, i):
        v = self._bits(self.valid16, i)
        pos = np.flatnonzero(v).astype(np.uint64)
        return pos, self._bits(self.rc16, i)[pos.astype(np.int64)]

    def count(self, i):
        return int(self._bits(self.valid16, i).sum())

    def iter(self, i, buffer: bytes, rc: bytes):
        k, n = self.k, len(rc)
        pos, flg = self.arrays(i)
        for p, f in zip(pos.tolist(), flg.tolist()):
            yield (p, rc[n - p - k: n - p], True) if f else (p, buffer[p: p + k], False)


def canonical_kmers_planes(records, k: int, ctx: Context = None) -> CanonicalKmersPlanes:
    """Sequence::canonical_kmers for a whole batch of records in one device pass, bit-plane result (see CanonicalKmersPlanes)."""
    c = _ctx(ctx)
    if k < 1 or k > 255:
        raise ValueError("k must be 1..255")
    seq, offs = _pack_records(records)
    cap = int(offs[-1]) // 16 + len(records) + 1
    rec_bit = np.zeros(len(records) + 1, dtype=np.uint64)
    valid16 = np.zeros(cap, dtype=np.uint16)
    rc16 = np.zeros(cap, dtype=np.uint16)
    nw, tot = C.c_uint64(0), C.c_uint64(0)
    L.check(L.lib().ntk_canonical_kmers_batch_planes(c._h, seq, offs.ctypes.data, len(records), k, rec_bit.ctypes.data, valid16.ctypes.data,
                                                     rc16.ctypes.data, cap, C.byref(nw), C.byref(tot)), "ntk_canonical_kmers_batch_planes")
    return CanonicalKmersPlanes(k, np.diff(offs.astype(np.int64)), rec_bit, valid16[: nw.value], rc16[: nw.value], tot.value)


class BitKmersPlanes(CanonicalKmersPlanes):
    """The items of Sequence::bit_kmers(k, canonical) for a batch of records (ntk_bit_kmers_batch_planes): per window start "emitted" and
    "was_rc" as bit planes, and the packed values dense, one u64 per plane position.  `iter(i)` yields exactly what the reference iterator
    yields for record i - (pos, (value, k), was_rc), reference src/bitkmer.rs:97-108 - and `arrays(i)` the (pos, value, was_rc) arrays.
    Without values (values=False at the call) a quarter byte per base crosses PCIe and the values are packed here from the window's bases."""

    def __init__(self, k, lengths, rec_bit, valid16, rc16, total, values, records):
        super().__init__(k, lengths, rec_bit, valid16, rc16, total)
        self.values, self.records = values, records

    def arrays(self, i):
        pos, flg = super().arrays(i)
        if self.values is not None:
            return pos, self.values[int(self.rec_bit[i]) + pos.astype(np.int64)], flg
        # the value of an emitted window from its bases: A0 C1 G2 T3, first base most significant (reference src/bitkmer.rs:5-36); the
        # reverse complement's value where was_rc says so (src/bitkmer.rs:112-132)
        rec = np.frombuffer(self.records[i], dtype=np.uint8)
        x = (rec >> 1) & 3
        code = (x ^ (x >> 1)).astype(np.uint64)
        vals = np.zeros(len(pos), dtype=np.uint64)
        for m in range(self.k):
            fwd = code[pos.astype(np.int64) + m]
            rcv = np.uint64(3) - code[pos.astype(np.int64) + self.k - 1 - m]
            vals = (vals << np.uint64(2)) | np.where(flg != 0, rcv, fwd)
        return pos, vals, flg

    def iter(self, i):
        pos, val, flg = self.arrays(i)
        for p, v, f in zip(pos.tolist(), val.tolist(), flg.tolist()):
            yield (p, (v, self.k), bool(f))


def bit_kmers_planes(records, k: int, canonical: bool, ctx: Context = None, values: bool = True) -> BitKmersPlanes:
    """Sequence::bit_kmers(k, canonical) for a whole batch of records in one device pass, bit-plane result + dense values (see BitKmersPlanes)."""
    c = _ctx(ctx)
    if k < 1 or k > 32:
        raise ValueError("k must be 1..32")
    seq, offs = _pack_records(records)
    cap = int(offs[-1]) // 16 + len(records) + 1
    rec_bit = np.zeros(len(records) + 1, dtype=np.uint64)
    valid16 = np.zeros(cap, dtype=np.uint16)
    rc16 = np.zeros(cap, dtype=np.uint16)
    vals = np.zeros(cap * 16, dtype=np.uint64) if values else None
    nw, tot = C.c_uint64(0), C.c_uint64(0)
    L.check(L.lib().ntk_bit_kmers_batch_planes(c._h, seq, offs.ctypes.data, len(records), k, int(canonical), rec_bit.ctypes.data, valid16.ctypes.data,
                                               rc16.ctypes.data, vals.ctypes.data if values else None, cap, C.byref(nw), C.byref(tot)),
            "ntk_bit_kmers_batch_planes")
    return BitKmersPlanes(k, np.diff(offs.astype(np.int64)), rec_bit, valid16[: nw.value], rc16[: nw.value], tot.value,
                          vals[: nw.value * 16] if values else None, list(records))


def minimizer(seq: bytes, length: int, ctx: Context = None) -> bytes:
    """sequence::minimizer (reference src/sequence.rs:139-152)."""
    c = _ctx(ctx)
    if length < 1 or len(seq) < length:
        raise ValueError("need 1 <= length <= len(seq)")
    out = C.create_string_buffer(length)
    L.check(L.lib().ntk_minimizer(c._h, seq, len(seq), length, out), "ntk_minimizer")
    return out.raw[:length]


def minimizer_batch(records, length: int, ctx: Context = None, with_positions: bool = False):
    """sequence::minimizer (reference src/sequence.rs:139-152) for every record of a batch in one device pass (ntk_minimizer_batch):
    the list of minimizers; with_positions: (minimizers, window starts on the winning strand's string, is_rc flags).  A record shorter
    than `length` raises ValueError naming it (the reference panics there)."""
    c = _ctx(ctx)
    if length < 1:
        raise ValueError("length must be >= 1")
    n = len(records)
    if n == 0:
        return ([], np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint8)) if with_positions else []
    seq, offs = _pack_records(records)
    out = np.empty(n * length, dtype=np.uint8)
    pos = np.empty(n, dtype=np.uint64)
    flg = np.empty(n, dtype=np.uint8)
    bad = C.c_uint64(0)
    rc = L.lib().ntk_minimizer_batch(c._h, seq, offs.ctypes.data, n, length, out.ctypes.data, pos.ctypes.data, flg.ctypes.data, C.byref(bad))
    if rc != L.NTK_OK and bad.value != 0xFFFFFFFFFFFFFFFF:
        raise ValueError(f"record {bad.value} is shorter than the minimizer length {length}")
    L.check(rc, "ntk_minimizer_batch")
    mins = [out[i * length:(i + 1) * length].tobytes() for i in range(n)]
    return (mins, pos, flg) if with_positions else mins


def canonical(seq: bytes, ctx: Context = None) -> bytes:
    """sequence::canonical (reference src/sequence.rs:110-134): the lower of seq and its reverse complement."""
    c = _ctx(ctx)
    out = C.create_string_buffer(max(len(seq), 1))
    L.check(L.lib().ntk_canonical(c._h, seq, len(seq), out, None), "ntk_canonical")
    return out.raw[:len(seq)]


# (outside SURVEY.md section 8 - record writers / header and quality helpers of the reference's surface, SURVEY section 2 rows 5 and 11: host-side
# conveniences kept for callers of the Python facade; nothing on the hot path uses them and no further surface of this kind is added)
def mask_header_tabs(id: bytes):
    """reference src/parser/record.rs:188-194: tabs -> '|'; None when there is nothing to mask."""
    return id.replace(b"\t", b"|") if b"\t" in id else None


def mask_header_utf8(id: bytes):
    """reference src/parser/record.rs:197-204: invalid UTF-8 -> U+FFFD; None when the header is valid UTF-8."""
    try:
        id.decode("utf-8")
        return None
    except UnicodeDecodeError:
        return id.decode("utf-8", "replace").encode("utf-8")


def bit_minimizers(values, k: int, m: int, ctx: Context = None) -> np.ndarray:
    """bitkmer::minimizer (reference src/bitkmer.rs:146-162) over an array of packed k-mers."""
    c = _ctx(ctx)
    v = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty_like(v)
    L.check(L.lib().ntk_bit_minimizers(c._h, v.ctypes.data, v.size, k, m, out.ctypes.data), "ntk_bit_minimizers")
    return out


def quality_mask(seq: bytes, qual: bytes, score: int, ctx: Context = None) -> bytes:
    """QualitySequence::quality_mask (reference src/sequence.rs:285-296); zip semantics: min(len(seq), len(qual))."""
    c = _ctx(ctx)
    n = min(len(seq), len(qual))
    out = C.create_string_buffer(max(n, 1))
    L.check(L.lib().ntk_quality_mask(c._h, seq[:n], qual[:n], n, score, out), "ntk_quality_mask")
    return out.raw[:n]


def bit_reverse_complement(values, k: int, ctx: Context = None) -> np.ndarray:
    """bitkmer::reverse_complement (reference src/bitkmer.rs:112-132) over an array of packed k-mers."""
    c = _ctx(ctx)
    v = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty_like(v)
    L.check(L.lib().ntk_bit_canonical(c._h, v.ctypes.data, v.size, k, 0, out.ctypes.data, None), "ntk_bit_canonical")
    return out


def bit_canonical(values, k: int, ctx: Context = None):
    """bitkmer::canonical (reference src/bitkmer.rs:136-143): (canonical values, was_rc flags)."""
    c = _ctx(ctx)
    v = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty_like(v)
    flg = np.empty(v.size, dtype=np.uint8)
    L.check(L.lib().ntk_bit_canonical(c._h, v.ctypes.data, v.size, k, 1, out.ctypes.data, flg.ctypes.data), "ntk_bit_canonical")
    return out, flg.astype(bool)


def bitmer_to_bytes(value: int, k: int) -> bytes:
    """bitkmer::bitmer_to_bytes (reference src/bitkmer.rs:164-186): pure formatting of one packed k-mer (host side)."""
    return bytes(b"ACGT"[(value >> (2 * (k - 1 - i))) & 3] for i in range(k))


def bytes_to_bitmer(kmer: bytes):
    """Inverse of bitmer_to_bytes, the reference's test helper (src/bitkmer.rs:288-296): (value, k)."""
    v = 0
    for ch in kmer:
        v = (v << 2) | {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}[ch]
    return v & ((1 << (2 * len(kmer))) - 1), len(kmer)
