"""ctypes binding of the CPU oracle (oracle/ntk_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under needletail_amd/ imports this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libntk_oracle.so")

PATH_BYTES_CANONICAL, PATH_BITS, PATH_BITS_CANONICAL = 0, 1, 2
PRE_NONE, PRE_STRIP_RETURNS, PRE_NORMALIZE, PRE_NORMALIZE_IUPAC = 0, 1, 2, 3
HIST_BINS = 4096


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ntk_oracle.c")
    hdr = os.path.join(_HERE, "ntk_oracle.h")
    if os.path.exists(src) and (
        force
        or not os.path.exists(_SO)
        or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libntk_oracle.so"])
    return _SO


def use_native_build() -> str:
    """Switch this process to a `-O3 -march=native` build of the same source (the CPU baseline is timed
    with the host's own instruction set).  It is compiled where it runs (oracle/_native/, git-ignored): an object built
    for another machine's `native` could fault here.  Returns the compiler flags used; falls back to the portable
    build (and says so) when gcc is not available."""
    global _SO, _lib
    out_dir = os.path.join(_HERE, "_native")
    out = os.path.join(out_dir, "libntk_oracle_native.so")
    flags = "-O3 -march=native -fPIC -std=c11"
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", *flags.split(), "-shared", "-o", out, os.path.join(_HERE, "ntk_oracle.c"), "-lpthread"])
    except (OSError, subprocess.CalledProcessError):
        return "-O3 -march=x86-64-v2 (portable build: gcc -march=native failed here)"
    _SO, _lib = out, None
    return flags


class BitKmer(C.Structure):
    _fields_ = [("seq", C.c_uint64), ("k", C.c_uint8)]


class Stats(C.Structure):
    _fields_ = [
        ("n_total", C.c_uint64),
        ("n_fwd", C.c_uint64),
        ("n_rc", C.c_uint64),
        ("sum", C.c_uint64),
        ("xr", C.c_uint64),
        ("hist", C.c_uint64 * HIST_BINS),
    ]

    def as_dict(self) -> dict:
        return {
            "n_total": int(self.n_total),
            "n_fwd": int(self.n_fwd),
            "n_rc": int(self.n_rc),
            "sum": int(self.sum),
            "xor": int(self.xr),
            "hist": np.ctypeslib.as_array(self.hist).copy(),
        }


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _SO.endswith("_native.so"):
            build()
        L = C.CDLL(_SO)
        u8p, u64p, sz = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_size_t
        L.ntko_normalize.restype = sz
        L.ntko_normalize.argtypes = [C.c_char_p, sz, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        L.ntko_strip_returns.restype = sz
        L.ntko_strip_returns.argtypes = [C.c_char_p, sz, C.c_char_p, C.POINTER(C.c_int)]
        L.ntko_complement.restype = C.c_uint8
        L.ntko_complement.argtypes = [C.c_uint8]
        L.ntko_reverse_complement.restype = None
        L.ntko_reverse_complement.argtypes = [C.c_char_p, sz, C.c_char_p]
        L.ntko_canonical.restype = C.c_int
        L.ntko_canonical.argtypes = [C.c_char_p, sz, C.c_char_p]
        L.ntko_minimizer.restype = None
        L.ntko_minimizer.argtypes = [C.c_char_p, sz, sz, C.c_char_p]
        L.ntko_quality_mask.restype = sz
        L.ntko_quality_mask.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_uint8, C.c_char_p]
        L.ntko_is_good_base.restype = C.c_int
        L.ntko_is_good_base.argtypes = [C.c_uint8]
        L.ntko_kmers_count.restype = sz
        L.ntko_kmers_count.argtypes = [sz, C.c_uint8]
        L.ntko_nuc2bit.restype = C.c_int
        L.ntko_nuc2bit.argtypes = [C.c_uint8]
        L.ntko_bit_reverse_complement.restype = BitKmer
        L.ntko_bit_reverse_complement.argtypes = [BitKmer]
        L.ntko_bit_canonical.restype = BitKmer
        L.ntko_bit_canonical.argtypes = [BitKmer, C.POINTER(C.c_int)]
        L.ntko_bit_minimizer.restype = BitKmer
        L.ntko_bit_minimizer.argtypes = [BitKmer, C.c_uint8]
        L.ntko_bitmer_to_bytes.restype = None
        L.ntko_bitmer_to_bytes.argtypes = [BitKmer, C.c_char_p]
        L.ntko_bytes_to_bitmer.restype = BitKmer
        L.ntko_bytes_to_bitmer.argtypes = [C.c_char_p, C.c_uint8]
        L.ntko_canonical_kmers_all.restype = sz
        L.ntko_canonical_kmers_all.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_uint8, u64p, u8p, sz]
        L.ntko_bit_kmers_all.restype = sz
        L.ntko_bit_kmers_all.argtypes = [C.c_char_p, sz, C.c_uint8, C.c_int, u64p, u64p, u8p, sz]
        L.ntko_stats_clear.argtypes = [C.POINTER(Stats)]
        L.ntko_reduce_record.restype = C.c_int
        L.ntko_reduce_record.argtypes = [C.POINTER(Stats), C.c_char_p, sz, C.c_uint8, C.c_int, C.c_int]
        L.ntko_reduce_batch.restype = C.c_int
        L.ntko_reduce_batch.argtypes = [C.POINTER(Stats), C.c_void_p, C.c_void_p, sz, sz, C.c_uint8, C.c_int, C.c_int]
        L.ntko_reduce_batch_mt.restype = C.c_int
        L.ntko_reduce_batch_mt.argtypes = [C.POINTER(Stats), C.c_void_p, C.c_void_p, sz, sz, C.c_uint8, C.c_int, C.c_int, C.c_int]
        L.ntko_reduce_batch_mt2.restype = C.c_int
        L.ntko_reduce_batch_mt2.argtypes = [C.POINTER(Stats), C.c_void_p, C.c_void_p, sz, sz, C.c_uint8, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ntko_count_batch_mt.restype = C.c_int
        L.ntko_count_batch_mt.argtypes = [u64p, u64p, C.c_void_p, C.c_void_p, sz, sz, C.c_uint8, C.c_int, C.c_int, C.c_int]
        L.ntko_reduce_fused.restype = C.c_int
        L.ntko_reduce_fused.argtypes = [C.POINTER(Stats), C.c_void_p, sz, C.c_uint8, C.c_int, C.c_int, C.c_int]
        L.ntko_minimizers_reduce.restype = C.c_int
        L.ntko_minimizers_reduce.argtypes = [C.POINTER(Stats), C.c_void_p, sz, C.c_uint8, C.c_uint32, C.c_int, C.c_int]
        L.ntko_splitmix64_at.restype = C.c_uint64
        L.ntko_splitmix64_at.argtypes = [C.c_uint64, C.c_uint64]
        L.ntko_synth_reads.restype = None
        L.ntko_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


# ---- pythonic wrappers ---------------------------------------------------------------------


def normalize(seq: bytes, iupac: bool = False):
    """Returns (bytes, changed) following sequence::normalize; changed False == reference `None`."""
    out = C.create_string_buffer(max(len(seq), 1))
    ch = C.c_int(0)
    n = lib().ntko_normalize(seq, len(seq), int(iupac), out, C.byref(ch))
    return out.raw[:n], bool(ch.value)


def strip_returns(seq: bytes):
    out = C.create_string_buffer(max(len(seq), 1))
    b = C.c_int(0)
    n = lib().ntko_strip_returns(seq, len(seq), out, C.byref(b))
    return out.raw[:n], bool(b.value)


def complement(c: int) -> int:
    return lib().ntko_complement(c)


def reverse_complement(seq: bytes) -> bytes:
    out = C.create_string_buffer(max(len(seq), 1))
    lib().ntko_reverse_complement(seq, len(seq), out)
    return out.raw[: len(seq)]


def canonical(seq: bytes) -> bytes:
    out = C.create_string_buffer(max(len(seq), 1))
    lib().ntko_canonical(seq, len(seq), out)
    return out.raw[: len(seq)]


def minimizer(seq: bytes, length: int) -> bytes:
    out = C.create_string_buffer(max(length, 1))
    lib().ntko_minimizer(seq, len(seq), length, out)
    return out.raw[:length]


def quality_mask(seq: bytes, qual: bytes, score: int) -> bytes:
    out = C.create_string_buffer(max(len(seq), 1))
    n = lib().ntko_quality_mask(seq, len(seq), qual, len(qual), score, out)
    return out.raw[:n]


def kmers(seq: bytes, k: int):
    """Kmers iterator (src/kmer.rs:13-41) as a list of slices."""
    return [seq[i : i + k] for i in range(lib().ntko_kmers_count(len(seq), k))]


def canonical_kmers(seq: bytes, rc: bytes, k: int):
    """List of (pos, slice, is_rc) exactly as CanonicalKmers yields them."""
    cap = max(len(seq), 1)
    pos = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = lib().ntko_canonical_kmers_all(
        seq, len(seq), rc, len(rc), k,
        pos.ctypes.data_as(C.POINTER(C.c_uint64)), flg.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
    out = []
    for i in range(n):
        p, f = int(pos[i]), bool(flg[i])
        sl = rc[len(rc) - p - k : len(rc) - p] if f else seq[p : p + k]
        out.append((p, sl, f))
    return out


def bit_kmers_arrays(seq: bytes, k: int, canonical: bool):
    cap = max(len(seq), 1)
    pos = np.empty(cap, dtype=np.uint64)
    val = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = lib().ntko_bit_kmers_all(
        seq, len(seq), k, int(canonical),
        pos.ctypes.data_as(C.POINTER(C.c_uint64)), val.ctypes.data_as(C.POINTER(C.c_uint64)),
        flg.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
    return pos[:n], val[:n], flg[:n]


def bit_kmers(seq: bytes, k: int, canonical: bool):
    """List of (pos, (value, k), was_rc) exactly as BitNuclKmer yields them."""
    pos, val, flg = bit_kmers_arrays(seq, k, canonical)
    return [(int(p), (int(v), k), bool(f)) for p, v, f in zip(pos, val, flg)]


def canonical_kmers_arrays(seq: bytes, rc: bytes, k: int):
    cap = max(len(seq), 1)
    pos = np.empty(cap, dtype=np.uint64)
    flg = np.empty(cap, dtype=np.uint8)
    n = lib().ntko_canonical_kmers_all(
        seq, len(seq), rc, len(rc), k,
        pos.ctypes.data_as(C.POINTER(C.c_uint64)), flg.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
    return pos[:n], flg[:n]


def bit_reverse_complement(v: int, k: int) -> int:
    return int(lib().ntko_bit_reverse_complement(BitKmer(v, k)).seq)


def bit_canonical(v: int, k: int):
    f = C.c_int(0)
    r = lib().ntko_bit_canonical(BitKmer(v, k), C.byref(f))
    return int(r.seq), bool(f.value)


def bit_minimizer(v: int, k: int, m: int) -> int:
    return int(lib().ntko_bit_minimizer(BitKmer(v, k), m).seq)


def bitmer_to_bytes(v: int, k: int) -> bytes:
    out = C.create_string_buffer(max(k, 1))
    lib().ntko_bitmer_to_bytes(BitKmer(v, k), out)
    return out.raw[:k]


def bytes_to_bitmer(kmer: bytes) -> int:
    return int(lib().ntko_bytes_to_bitmer(kmer, len(kmer)).seq)


def reduce_records(records, k: int, path: int, pre: int) -> dict:
    st = Stats()
    lib().ntko_stats_clear(C.byref(st))
    for r in records:
        rc = lib().ntko_reduce_record(C.byref(st), r, len(r), k, path, pre)
        if rc:
            raise ValueError("ntko_reduce_record failed")
    return st.as_dict()


def reduce_batch(buf: np.ndarray, offsets: np.ndarray, gap: int, k: int, path: int, pre: int,
                 threads: int = 1, reuse_buffers: bool = False) -> dict:
    """reuse_buffers: every thread keeps its normalize / reverse-complement buffers across records - NOT what the reference does
    (it allocates per record); the cpu_baseline's informational "arena" variant."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    st = Stats()
    lib().ntko_stats_clear(C.byref(st))
    nrec = len(offsets) - 1
    if threads <= 1 and not reuse_buffers:
        rc = lib().ntko_reduce_batch(C.byref(st), buf.ctypes.data, offsets.ctypes.data, nrec, gap, k, path, pre)
    else:
        rc = lib().ntko_reduce_batch_mt2(C.byref(st), buf.ctypes.data, offsets.ctypes.data, nrec, gap, k, path, pre, max(1, threads),
                                         int(reuse_buffers))
    if rc:
        raise ValueError("ntko_reduce_batch failed")
    return st.as_dict()


def count_batch(buf: np.ndarray, offsets: np.ndarray, gap: int, k: int, path: int, pre: int, threads: int = 1):
    """(n_total, n_fwd) by the reference benchmark's own loop (benches/benchmark.rs:32-41)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nt, nf = C.c_uint64(0), C.c_uint64(0)
    rc = lib().ntko_count_batch_mt(C.byref(nt), C.byref(nf), buf.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                   gap, k, path, pre, threads)
    if rc:
        raise ValueError("ntko_count_batch_mt failed")
    return int(nt.value), int(nf.value)


def reduce_fused(buf, k: int, canonical: bool, tie_rc: bool, accept_u: bool) -> dict:
    buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf,
                               dtype=np.uint8)
    st = Stats()
    lib().ntko_stats_clear(C.byref(st))
    rc = lib().ntko_reduce_fused(C.byref(st), buf.ctypes.data, buf.size, k, int(canonical), int(tie_rc), int(accept_u))
    if rc:
        raise ValueError("ntko_reduce_fused failed")
    return st.as_dict()


def reduce_fused_parallel(buf: np.ndarray, stride: int, k: int, canonical: bool, tie_rc: bool, accept_u: bool, threads: int) -> dict:
    """reduce_fused over a batch of fixed-stride records (each followed by its break byte), the records split over `threads`
    Python threads (ctypes releases the GIL; no k-mer spans a record, so the parts simply add up)."""
    from concurrent.futures import ThreadPoolExecutor
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n_rec = len(buf) // stride
    threads = max(1, min(threads, n_rec or 1))
    cuts = [n_rec * t // threads * stride for t in range(threads)] + [len(buf)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda i: reduce_fused(buf[cuts[i]:cuts[i + 1]], k, canonical, tie_rc, accept_u), range(threads)))
    out = parts[0]
    for q in parts[1:]:
        for key in ("n_total", "n_fwd", "n_rc"):
            out[key] += q[key]
        out["sum"] = (out["sum"] + q["sum"]) & (2 ** 64 - 1)
        out["xor"] ^= q["xor"]
        out["hist"] = out["hist"] + q["hist"]
    return out


def minimizers_reduce(buf, k: int, w: int, accept_u: bool = True, tie_rc: bool = True) -> dict:
    buf = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf,
                               dtype=np.uint8)
    st = Stats()
    lib().ntko_stats_clear(C.byref(st))
    if lib().ntko_minimizers_reduce(C.byref(st), buf.ctypes.data, buf.size, k, w, int(accept_u), int(tie_rc)):
        raise ValueError("ntko_minimizers_reduce failed")
    return st.as_dict()


def synth_reads(seed: int, first_read: int, n_reads: int, read_len: int, n_per_1024: int) -> np.ndarray:
    out = np.empty(n_reads * (read_len + 1), dtype=np.uint8)
    lib().ntko_synth_reads(seed, first_read, n_reads, read_len, n_per_1024, out.ctypes.data)
    return out


def splitmix64_at(seed: int, index: int) -> int:
    return int(lib().ntko_splitmix64_at(seed, index))
