"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/needletail_amd.h declares, reports a loud error when no device exists, and the product package never
touches the oracle."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    so = os.path.join(ROOT, "needletail_amd", "libneedletail_amd.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "needletail_amd", "csrc")])
    return so


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "needletail_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ntk_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    so = _ensure_built()
    lib = C.CDLL(so)
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/needletail_amd.h but not exported"
    from needletail_amd import _lib
    assert sorted(_lib.SYMBOLS) == syms


def test_abi_version_and_strerror():
    from needletail_amd import _lib
    L = _lib.lib()
    assert L.ntk_abi_version() == 1
    assert _lib.strerror(0) == "ok"
    assert "k" in _lib.strerror(1)


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from needletail_amd import _lib
    h = C.c_void_p()
    assert _lib.lib().ntk_ctx_create(0, C.byref(h)) == 4  # NTK_ERR_NO_DEVICE
    import needletail_amd as nt
    with pytest.raises(nt.NtkError):
        nt.normalize(b"ACGT")


def test_product_never_references_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "needletail_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b|ntko_|ntk_oracle", txt):
                    bad.append(os.path.join(base, f))
    hdr = open(os.path.join(ROOT, "include", "needletail_amd.h")).read()
    assert "ntko_" not in hdr
    assert not bad, bad
    out = subprocess.run(["nm", "-D", "--undefined-only", _ensure_built()], capture_output=True, text=True).stdout
    assert "ntko_" not in out
