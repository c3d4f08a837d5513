"""ctypes binding of libneedletail_amd.so (the C ABI in include/needletail_amd.h).

There is no fallback: if the HIP library is missing or fails to load this raises, and every compute
entry point fails with a status code when no gfx950 device is usable."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEEDLETAIL_AMD_LIB") or os.path.join(_HERE, "libneedletail_amd.so")   # (the override: A/B builds, tools/)

NTK_OK = 0
PATH_BYTES_CANONICAL, PATH_BITS, PATH_BITS_CANONICAL = 0, 1, 2
PRE_NONE, PRE_STRIP_RETURNS, PRE_NORMALIZE, PRE_NORMALIZE_IUPAC = 0, 1, 2, 3
HIST_BINS = 4096
ACC_N_TOTAL, ACC_N_FWD, ACC_N_RC, ACC_SUM, ACC_XOR, ACC_HIST = 0, 1, 2, 3, 4, 8
ACC_UNDIGESTED, ACC_REDONE = 5, 6   # REDONE: diagnostic - speculative launches whose result came from the byte-walking kernel
ACC_XOR_BITS, ACC_WORDS = 8 + 4096, 8 + 4096 + 64
# ntk_ctx_set_option (test / A-B support)
OPT_COMPAT_CHUNK_BYTES, OPT_MINIMIZER_CHUNK_BYTES, OPT_MINIMIZER_ROUTE, OPT_COMPAT_PACK_THREADS, OPT_COPY_STREAMS = 1, 2, 3, 4, 5
OPT_BATCH_WAIT_POLL_US, OPT_GZ_INMEM_LIMIT_BYTES, OPT_GZ_STREAM_WINDOW_BYTES, OPT_PIPE_STATS = 6, 7, 8, 9
POLL_BLOCK = 0xFFFFFFFF
ROUTE_NO_REGFUSED, ROUTE_NO_GENERIC, ROUTE_NO_F64, ROUTE_NO_SPECULATION = 1, 2, 4, 8
ROUTE_TWO_PASS = ROUTE_NO_REGFUSED | ROUTE_NO_GENERIC

# every symbol include/needletail_amd.h declares (tests/test_abi.py checks header <-> library <-> this list)
SYMBOLS = [
    "ntk_strerror", "ntk_last_hip_error", "ntk_last_rccl_error", "ntk_abi_version", "ntk_device_count",
    "ntk_comm_init_all", "ntk_comm_unique_id", "ntk_comm_init_rank", "ntk_comm_size", "ntk_allreduce_accumulators", "ntk_comm_allreduce_time_ms", "ntk_comm_destroy",
    "ntk_ctx_create", "ntk_ctx_create_on_stream", "ntk_ctx_destroy", "ntk_ctx_synchronize",
    "ntk_ctx_set_launch", "ntk_ctx_set_option", "ntk_ctx_get_option", "ntk_ctx_enable_timing", "ntk_ctx_scan_time_ms",
    "ntk_accum_reset", "ntk_reduce_device", "ntk_reduce_device_quality", "ntk_accum_read", "ntk_accum_device_ptr", "ntk_accum_bind_device",
    "ntk_materialize_device", "ntk_materialize_device_quality",
    "ntk_batch_acquire", "ntk_batch_append", "ntk_batch_append_quality", "ntk_batch_buffers", "ntk_batch_submit", "ntk_batch_wait",
    "ntk_batch_release",
    "ntk_normalize", "ntk_strip_returns", "ntk_reverse_complement", "ntk_canonical_kmers", "ntk_bit_kmers", "ntk_canonical_kmers_batch", "ntk_bit_kmers_batch", "ntk_pinned_alloc", "ntk_pinned_free",
    "ntk_canonical_kmers_batch_planes", "ntk_bit_kmers_batch_planes", "ntk_ctx_trim", "ntk_minimizer_batch",
    "ntk_synth_reads_device", "ntk_reverse_complement_records_device",
    "ntk_reader_open_file", "ntk_reader_open_memory", "ntk_reader_next", "ntk_reader_error", "ntk_reader_position", "ntk_reader_close",
    "ntk_scan_reader", "ntk_scan_buffer_parallel", "ntk_scan_file_parallel", "ntk_fastx_split_points", "ntk_gunzip", "ntk_gunzip_free", "ntk_scan_file_info",
    "ntk_minimizers_reduce_device", "ntk_minimizer", "ntk_canonical", "ntk_bit_minimizers", "ntk_quality_mask", "ntk_bit_canonical",
]


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("path", C.c_uint32), ("pre", C.c_uint32), ("flags", C.c_uint32)]


FLAG_RESET = 1 << 16   # NTK_FLAG_RESET: the reduce call zeroes the accumulators first, inside its own kernel launch


def flags(w: int = 0, quality_cutoff: int = 0, reset: bool = False) -> int:
    """ntk_params.flags (NTK_FLAGS): bits 7:0 minimizer window, bits 15:8 quality cutoff, bit 16 NTK_FLAG_RESET."""
    if not (0 <= w <= 255 and 0 <= quality_cutoff <= 255):
        raise ValueError("w and quality_cutoff must be 0..255")
    return w | (quality_cutoff << 8) | (FLAG_RESET if reset else 0)


class Result(C.Structure):
    _fields_ = [("n_total", C.c_uint64), ("n_fwd", C.c_uint64), ("n_rc", C.c_uint64), ("sum", C.c_uint64),
                ("xr", C.c_uint64), ("hist", C.c_uint64 * HIST_BINS), ("n_undigested", C.c_uint64)]


class Record(C.Structure):
    _fields_ = [("id", C.c_void_p), ("id_len", C.c_uint64), ("seq", C.c_void_p), ("seq_len", C.c_uint64),
                ("qual", C.c_void_p), ("qual_len", C.c_uint64), ("format", C.c_uint32), ("line_ending", C.c_uint32),
                ("line", C.c_uint64), ("num_bases", C.c_uint64), ("byte", C.c_uint64)]


class GunzipInfo(C.Structure):
    _fields_ = [("route", C.c_uint32), ("threads", C.c_uint32), ("chunks", C.c_uint32), ("chunks_dropped", C.c_uint32),
                ("members", C.c_uint32), ("streamed", C.c_uint32), ("search_s", C.c_double), ("decode_s", C.c_double),
                ("decode_busy_s", C.c_double), ("crc_s", C.c_double), ("marker_symbols", C.c_uint64),
                ("chunks_deferred", C.c_uint32), ("parse_threads", C.c_uint32), ("peak_backlog_bytes", C.c_uint64), ("text_bytes", C.c_uint64),
                ("first_batch_s", C.c_double), ("total_s", C.c_double), ("resolve_busy_s", C.c_double)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class NtkError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        super().__init__(f"{what}: status {status} ({strerror(status)}; hip={last_hip_error()}, rccl={lib().ntk_last_rccl_error()})")


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C needletail_amd/csrc). needletail_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    pp = C.POINTER(C.c_void_p)
    L.ntk_strerror.restype = C.c_char_p
    L.ntk_strerror.argtypes = [i32]
    L.ntk_last_hip_error.restype = i32
    L.ntk_abi_version.restype = i32
    L.ntk_last_rccl_error.restype = i32
    L.ntk_device_count.argtypes = [C.POINTER(i32)]
    L.ntk_comm_init_all.argtypes = [C.POINTER(C.c_void_p), i32, pp]
    L.ntk_comm_unique_id.argtypes = [C.c_char_p]
    L.ntk_comm_init_rank.argtypes = [vp, i32, i32, C.c_char_p, pp]
    L.ntk_comm_size.argtypes = [vp]
    L.ntk_allreduce_accumulators.argtypes = [vp]
    L.ntk_comm_destroy.restype = None
    L.ntk_comm_destroy.argtypes = [vp]
    L.ntk_comm_allreduce_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    L.ntk_ctx_create.argtypes = [i32, pp]
    L.ntk_ctx_create_on_stream.argtypes = [i32, vp, pp]
    L.ntk_ctx_destroy.restype = None
    L.ntk_ctx_destroy.argtypes = [vp]
    L.ntk_ctx_synchronize.argtypes = [vp]
    L.ntk_ctx_set_launch.argtypes = [vp, i32, i32]
    L.ntk_ctx_set_option.argtypes = [vp, i32, u64]
    L.ntk_ctx_enable_timing.argtypes = [vp, i32]
    L.ntk_ctx_scan_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    L.ntk_accum_reset.argtypes = [vp]
    L.ntk_reduce_device.argtypes = [vp, vp, u64, C.POINTER(Params)]
    L.ntk_reduce_device_quality.argtypes = [vp, vp, vp, u64, C.POINTER(Params)]
    L.ntk_accum_read.argtypes = [vp, C.POINTER(Result)]
    L.ntk_accum_device_ptr.argtypes = [vp, pp]
    L.ntk_accum_bind_device.argtypes = [vp, vp]
    L.ntk_materialize_device.argtypes = [vp, vp, u64, C.POINTER(Params), vp, vp, vp]
    L.ntk_materialize_device_quality.argtypes = [vp, vp, vp, u64, C.POINTER(Params), vp, vp, vp]
    L.ntk_batch_acquire.argtypes = [vp, u64, u64, pp]
    L.ntk_batch_append.argtypes = [vp, C.c_char_p, u64, u32]
    L.ntk_batch_append_quality.argtypes = [vp, C.c_char_p, C.c_char_p, u64, u32, u32]
    L.ntk_batch_buffers.argtypes = [vp, pp, pp, C.POINTER(u64), C.POINTER(u64)]
    L.ntk_batch_submit.argtypes = [vp, vp, C.POINTER(Params)]
    L.ntk_batch_wait.argtypes = [vp, vp]
    L.ntk_batch_release.restype = None
    L.ntk_batch_release.argtypes = [vp, vp]
    L.ntk_normalize.argtypes = [vp, C.c_char_p, u64, i32, C.c_char_p, C.POINTER(u64), C.POINTER(i32)]
    L.ntk_strip_returns.argtypes = [vp, C.c_char_p, u64, C.c_char_p, C.POINTER(u64), C.POINTER(i32)]
    L.ntk_reverse_complement.argtypes = [vp, C.c_char_p, u64, C.c_char_p]
    L.ntk_canonical_kmers.argtypes = [vp, C.c_char_p, u64, u32, vp, vp, u64, C.POINTER(u64)]
    L.ntk_bit_kmers.argtypes = [vp, C.c_char_p, u64, u32, i32, vp, vp, vp, u64, C.POINTER(u64)]
    L.ntk_canonical_kmers_batch.argtypes = [vp, C.c_char_p, vp, u64, u32, vp, vp, vp, u64, C.POINTER(u64)]
    L.ntk_bit_kmers_batch.argtypes = [vp, C.c_char_p, vp, u64, u32, i32, vp, vp, vp, vp, u64, C.POINTER(u64)]
    L.ntk_pinned_alloc.argtypes = [u64, pp]
    L.ntk_pinned_free.restype = None
    L.ntk_pinned_free.argtypes = [vp]
    L.ntk_canonical_kmers_batch_planes.argtypes = [vp, C.c_char_p, vp, u64, u32, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.ntk_bit_kmers_batch_planes.argtypes = [vp, C.c_char_p, vp, u64, u32, i32, vp, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.ntk_ctx_trim.argtypes = [vp]
    L.ntk_synth_reads_device.argtypes = [vp, u64, u64, u64, u32, u32, vp]
    L.ntk_reverse_complement_records_device.argtypes = [vp, vp, vp, u64, u32, u32]
    L.ntk_reader_open_file.argtypes = [C.c_char_p, pp]
    L.ntk_reader_open_memory.argtypes = [C.c_char_p, u64, pp]
    L.ntk_reader_next.argtypes = [vp, C.POINTER(Record)]
    L.ntk_reader_error.argtypes = [vp, C.POINTER(i32), C.POINTER(u64), C.c_char_p, u64, C.c_char_p, u64]
    L.ntk_reader_position.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    L.ntk_reader_close.restype = None
    L.ntk_reader_close.argtypes = [vp]
    L.ntk_scan_reader.argtypes = [vp, vp, C.POINTER(Params), u64, u32, C.POINTER(u64), C.POINTER(u64)]
    L.ntk_fastx_split_points.argtypes = [C.c_char_p, u64, u32, vp]
    L.ntk_scan_buffer_parallel.argtypes = [vp, C.c_char_p, u64, C.POINTER(Params), u64, u32, C.POINTER(u64), C.POINTER(u64)]
    L.ntk_scan_file_parallel.argtypes = [vp, C.c_char_p, C.POINTER(Params), u64, u32, C.POINTER(u64), C.POINTER(u64)]
    L.ntk_gunzip.argtypes = [C.c_char_p, u64, u32, pp, C.POINTER(u64), C.POINTER(GunzipInfo)]
    L.ntk_scan_file_info.argtypes = [C.POINTER(GunzipInfo)]
    L.ntk_ctx_get_option.argtypes = [vp, i32, C.POINTER(u64)]
    L.ntk_gunzip_free.restype = None
    L.ntk_gunzip_free.argtypes = [vp, u64]
    L.ntk_minimizers_reduce_device.argtypes = [vp, vp, u64, C.POINTER(Params), u32]
    L.ntk_minimizer.argtypes = [vp, C.c_char_p, u64, u32, C.c_char_p]
    L.ntk_minimizer_batch.argtypes = [vp, C.c_char_p, vp, u64, u32, vp, vp, vp, C.POINTER(u64)]
    L.ntk_canonical.argtypes = [vp, C.c_char_p, u64, C.c_char_p, C.POINTER(i32)]
    L.ntk_bit_minimizers.argtypes = [vp, vp, u64, u32, u32, vp]
    L.ntk_bit_canonical.argtypes = [vp, vp, u64, u32, i32, vp, vp]
    L.ntk_quality_mask.argtypes = [vp, C.c_char_p, C.c_char_p, u64, C.c_uint8, C.c_char_p]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("ntk_last_hip_error", "ntk_last_rccl_error", "ntk_abi_version"):
            fn.restype = C.c_int
    _lib = L
    return L


def strerror(status: int) -> str:
    return lib().ntk_strerror(status).decode()


def last_hip_error() -> int:
    return lib().ntk_last_hip_error()


def device_count() -> int:
    """Usable gfx950 devices (ntk_device_count); 0 without a GPU."""
    n = C.c_int(0)
    check(lib().ntk_device_count(C.byref(n)), "ntk_device_count")
    return n.value


def check(status: int, what: str) -> None:
    if status != NTK_OK:
        raise NtkError(status, what)
