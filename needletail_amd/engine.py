"""Context: one (thread, device) handle on the HIP engine — the batch face of the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(x) -> int:
    """Device pointer of a torch tensor / int."""
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError("expected a device pointer (int) or a torch tensor")


def result_to_dict(r: L.Result) -> dict:
    return {"n_total": int(r.n_total), "n_fwd": int(r.n_fwd), "n_rc": int(r.n_rc), "sum": int(r.sum),
            "xor": int(r.xr), "hist": np.ctypeslib.as_array(r.hist).copy(), "n_undigested": int(r.n_undigested)}


class Context:
    def __init__(self, device: int = 0, stream=None):
        """stream: None -> the ctx owns a new HIP stream; an int (hipStream_t, e.g.
        torch.cuda.current_stream().cuda_stream) -> kernels are enqueued on the caller's stream."""
        self._h = C.c_void_p()
        if stream is None:
            L.check(L.lib().ntk_ctx_create(device, C.byref(self._h)), "ntk_ctx_create")
        else:
            L.check(L.lib().ntk_ctx_create_on_stream(device, C.c_void_p(int(stream)), C.byref(self._h)),
                    "ntk_ctx_create_on_stream")
        self.device = device

    def close(self):
        if self._h:
            L.lib().ntk_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- launch / timing -----------------------------------------------------------------------
    def set_launch(self, blocks: int, threads: int):
        L.check(L.lib().ntk_ctx_set_launch(self._h, blocks, threads), "ntk_ctx_set_launch")

    def set_option(self, option: int, value: int = 0):
        """ntk_ctx_set_option: test / A-B support (chunk sizes, minimizer routes switched off); value 0 = the default."""
        L.check(L.lib().ntk_ctx_set_option(self._h, option, value), "ntk_ctx_set_option")

    def get_option(self, option: int) -> int:
        v = C.c_uint64(0)
        L.check(L.lib().ntk_ctx_get_option(self._h, option, C.byref(v)), "ntk_ctx_get_option")
        return int(v.value)

    def enable_timing(self, on: bool = True):
        L.check(L.lib().ntk_ctx_enable_timing(self._h, int(on)), "ntk_ctx_enable_timing")

    def scan_time_ms(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        L.check(L.lib().ntk_ctx_scan_time_ms(self._h, C.byref(ms), C.byref(n)), "ntk_ctx_scan_time_ms")
        return ms.value, int(n.value)

    def synchronize(self):
        L.check(L.lib().ntk_ctx_synchronize(self._h), "ntk_ctx_synchronize")

    # -- reduce mode -----------------------------------------------------------------------------
    def accum_reset(self):
        L.check(L.lib().ntk_accum_reset(self._h), "ntk_accum_reset")

    def reduce_device(self, d_seq, n_bytes: int, k: int, path: int, pre: int, w: int = 0, d_qual=None,
                      quality_cutoff: int = 0, reset: bool = False):
        """w > 0: fold windowed minimizers (w k-mers per window) instead of every k-mer.  d_qual + quality_cutoff: mask
        bases whose quality byte is below the cutoff first (QualitySequence::quality_mask, reference src/sequence.rs:285-296).
        reset: start a new result (NTK_FLAG_RESET: accum_reset() folded into this call's kernel launch)."""
        p = L.Params(k, path, pre, L.flags(w, quality_cutoff, reset))
        if d_qual is None:
            L.check(L.lib().ntk_reduce_device(self._h, C.c_void_p(_ptr(d_seq)), n_bytes, C.byref(p)), "ntk_reduce_device")
        else:
            L.check(L.lib().ntk_reduce_device_quality(self._h, C.c_void_p(_ptr(d_seq)), C.c_void_p(_ptr(d_qual)), n_bytes,
                                                      C.byref(p)), "ntk_reduce_device_quality")

    def accum_read(self) -> dict:
        r = L.Result()
        L.check(L.lib().ntk_accum_read(self._h, C.byref(r)), "ntk_accum_read")
        return result_to_dict(r)

    def accum_device_ptr(self) -> int:
        p = C.c_void_p()
        L.check(L.lib().ntk_accum_device_ptr(self._h, C.byref(p)), "ntk_accum_device_ptr")
        return int(p.value)

    def accum_bind_device(self, d_words):
        """Accumulate into caller-owned device memory (ACC_WORDS int64, e.g. a torch tensor)."""
        L.check(L.lib().ntk_accum_bind_device(self._h, C.c_void_p(_ptr(d_words)) if d_words is not None else None),
                "ntk_accum_bind_device")

    # -- materialise mode --------------------------------------------------------------------------
    def materialize_device(self, d_seq, n_bytes: int, k: int, path: int, pre: int, d_values, d_valid16, d_rc16,
                           d_qual=None, quality_cutoff: int = 0):
        p = L.Params(k, path, pre, L.flags(0, quality_cutoff))
        dv = C.c_void_p(_ptr(d_values)) if d_values is not None else None
        dq = C.c_void_p(_ptr(d_qual)) if d_qual is not None else None
        L.check(L.lib().ntk_materialize_device_quality(self._h, C.c_void_p(_ptr(d_seq)), dq, n_bytes, C.byref(p), dv,
                                                       C.c_void_p(_ptr(d_valid16)), C.c_void_p(_ptr(d_rc16))),
                "ntk_materialize_device")

    def minimizers_reduce_device(self, d_seq, n_bytes: int, k: int, w: int, path: int, pre: int):
        """Windowed minimizers (w k-mers per window) reduced into the accumulators."""
        p = L.Params(k, path, pre, 0)
        L.check(L.lib().ntk_minimizers_reduce_device(self._h, C.c_void_p(_ptr(d_seq)), n_bytes, C.byref(p), w),
                "ntk_minimizers_reduce_device")

    # -- device utilities ------------------------------------------------------------------------------
    def synth_reads_device(self, seed: int, first_read: int, n_reads: int, read_len: int, n_per_1024: int, d_out):
        L.check(L.lib().ntk_synth_reads_device(self._h, seed, first_read, n_reads, read_len, n_per_1024,
                                               C.c_void_p(_ptr(d_out))), "ntk_synth_reads_device")

    def reverse_complement_records_device(self, d_in, d_out, n_records: int, record_len: int, stride: int):
        L.check(L.lib().ntk_reverse_complement_records_device(self._h, C.c_void_p(_ptr(d_in)), C.c_void_p(_ptr(d_out)),
                                                              n_records, record_len, stride),
                "ntk_reverse_complement_records_device")

    # -- pinned batches -----------------------------------------------------------------------------------
    def batch(self, max_bytes: int, max_records: int) -> "Batch":
        return Batch(self, max_bytes, max_records)


class Batch:
    """Pinned host batch: append records (the CPU parser's job), submit (async H2D + scan), wait, reuse."""

    def __init__(self, ctx: Context, max_bytes: int, max_records: int):
        self.ctx = ctx
        self._h = C.c_void_p()
        L.check(L.lib().ntk_batch_acquire(ctx._h, max_bytes, max_records, C.byref(self._h)), "ntk_batch_acquire")

    def append(self, seq: bytes, pre: int, qual: bytes = None, quality_cutoff: int = 0) -> bool:
        """False when the batch is full (submit it and use another).  qual + quality_cutoff: carry the record's quality
        line for masking at that cutoff (submit with the same cutoff)."""
        if qual is not None and quality_cutoff:
            if len(qual) != len(seq):
                raise ValueError("sequence and quality lengths differ")
            rc = L.lib().ntk_batch_append_quality(self._h, seq, qual, len(seq), pre, quality_cutoff)
        else:
            rc = L.lib().ntk_batch_append(self._h, seq, len(seq), pre)
        if rc == 5:  # NTK_ERR_CAPACITY
            return False
        L.check(rc, "ntk_batch_append")
        return True

    def buffers(self):
        seq, off = C.c_void_p(), C.c_void_p()
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        L.check(L.lib().ntk_batch_buffers(self._h, C.byref(seq), C.byref(off), C.byref(nb), C.byref(nr)), "ntk_batch_buffers")
        s = np.ctypeslib.as_array(C.cast(seq, C.POINTER(C.c_uint8)), shape=(max(int(nb.value), 1),))[: int(nb.value)]
        o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(int(nr.value) + 1,))
        return s, o

    def submit(self, k: int, path: int, pre: int, w: int = 0, quality_cutoff: int = 0, reset: bool = False):
        p = L.Params(k, path, pre, L.flags(w, quality_cutoff, reset))
        L.check(L.lib().ntk_batch_submit(self.ctx._h, self._h, C.byref(p)), "ntk_batch_submit")

    def wait(self):
        L.check(L.lib().ntk_batch_wait(self.ctx._h, self._h), "ntk_batch_wait")

    def release(self):
        if self._h:
            L.lib().ntk_batch_release(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx
