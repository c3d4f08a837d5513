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
    hdr = open(os.path.join(ROOT, "include", "needletail_amd.h")).read()
    assert L.ntk_abi_version() == int(re.search(r"#define NTK_ABI_VERSION (\d+)", hdr).group(1)) == 4   # 4: round 6, ntk_result.n_undigested (k > 32 on the reduce face)
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


def _build_example(name="stdin_pipe"):
    exe = os.path.join(ROOT, "examples", name)
    src = os.path.join(ROOT, "examples", name + ".cpp")
    hdr = os.path.join(ROOT, "include", "needletail_amd.hpp")
    _ensure_built()
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
                               "-L" + os.path.join(ROOT, "needletail_amd"),
                               "-lneedletail_amd", "-Wl,-rpath,$ORIGIN/../needletail_amd"])
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu():
    """include/needletail_amd.hpp (C++ mirror of the reference surface) + examples/stdin_pipe.cpp build with plain g++."""
    import torch
    exe = _build_example()
    _build_example("mirror_check")
    if torch.cuda.is_available():
        pytest.skip("GPU present: the example is run by the gpu tests")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "test.fa")], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable gfx950 device" in r.stderr


@pytest.mark.gpu
def test_cpp_example_program_on_gpu():
    """The reference's example program (examples/stdin_pipe.rs) ported onto the C++ mirror: 28S.fasta -> 738 580 bases
    (benches/benchmark.rs:151), 8 108 AAAAs; '>id1\\nAGTCGTCA' -> 8 bases, 0 AAAAs (tests/test_stdin.rs:30-31)."""
    import tempfile
    exe = _build_example()
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "28S.fasta")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "There are 738580 bases in your file." in r.stdout and "There are 8108 AAAAs in your file." in r.stdout
    assert "batched: 570 records, 738580 bases, 8108 AAAAs" in r.stdout
    with tempfile.NamedTemporaryFile(suffix=".fa") as f:
        f.write(b">id1\nAGTCGTCA\n"); f.flush()
        r = subprocess.run([exe, f.name], capture_output=True, text=True)
    assert "There are 8 bases in your file." in r.stdout and "There are 0 AAAAs in your file." in r.stdout
    # the reference's own invocation: records piped into standard input (tests/test_stdin.rs:8-32)
    r = subprocess.run([exe], input=">id1\nAGTCGTCA", capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "There are 8 bases in your file." in r.stdout and "There are 0 AAAAs in your file." in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_quality_and_minimizers_on_gpu():
    """QualitySequence::quality_mask, minimizer and the bitkmer free functions of the C++ mirror against the reference's
    unit-test literals (src/sequence.rs:363-374) and the oracle."""
    import oracle as O
    exe = _build_example("mirror_check")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    assert lines[0] == "AGCN" and lines[1] == "AAA"
    cv, crc = O.bit_canonical(0xE4, 4)
    assert (int(lines[2]), int(lines[3])) == (cv, int(crc))
    assert int(lines[4]) == O.bit_minimizer(0x1B, 4, 2)
    assert [int(x) for x in lines[5:18]] == [2, 27, 14, 27, 14, 33, 33, 37, 37, 37, 33, 37, 27]
    # CanonicalKmersPlanes: three records in one buffer, every item as the reference iterator yields it (oracle = its restatement)
    recs = [b"ACGT", b"AGTCGTCA", b"nACGTACGTN"]
    want = []
    for i, rec in enumerate(recs):
        rc = O.reverse_complement(rec)
        want += [f"{i}:{p}:{kmer.decode()}:{int(f)}" for p, kmer, f in O.canonical_kmers(rec, rc, 2)]
    assert lines[18] == "planes" and int(lines[19]) == len(want)
    assert lines[20:20 + len(want)] == want
    # minimizer_batch: sequence::minimizer per record (the reference's literal among them), window start and strand of the winner
    at = 20 + len(want)
    assert lines[at] == "minimizer_batch"
    got = lines[at + 1: at + 4]
    from _refs import minimizer_with_position
    exp = []
    for rec in (b"ATTTCG", b"ACGT", b"TTGGCA"):
        best = minimizer_with_position(rec, 3)     # the reference's loop order decides between equal strings
        assert best[0] == O.minimizer(rec, 3)
        exp.append(f"{best[0].decode()}:{best[1]}:{best[2]}")
    assert got == exp and got[0].startswith("AAA:")
    # BitKmersPlanes: Sequence::bit_kmers(3, true) for the three records, every item as the reference iterator yields it
    at += 4
    wantb = []
    for i, rec in enumerate(recs):
        wantb += [f"{i}:{p}:{v}:{k}:{int(f)}" for p, (v, k), f in O.bit_kmers(rec, 3, True)]
    assert lines[at] == "bit_planes" and int(lines[at + 1]) == len(wantb)
    assert lines[at + 2: at + 2 + len(wantb)] == wantb


def _c_smoke():
    exe = os.path.join(ROOT, "tests", "abi_smoke")
    src = os.path.join(ROOT, "tests", "abi_smoke.c")
    _ensure_built()
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
                           "-L" + os.path.join(ROOT, "needletail_amd"), "-lneedletail_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "needletail_amd")])
    return exe


def test_header_is_plain_c_and_links():
    """include/needletail_amd.h compiled by gcc -std=c11 -pedantic (no C++, no torch types) and linked with the library;
    without a GPU the program must report the loud NTK_ERR_NO_DEVICE and nothing else."""
    import torch
    r = subprocess.run([_c_smoke()], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout
    if not torch.cuda.is_available():
        assert "no device" in r.stdout


@pytest.mark.gpu
def test_c_smoke_program_on_gpu():
    r = subprocess.run([_c_smoke()], capture_output=True, text=True)
    assert r.returncode == 0 and "abi_smoke ok (gpu)" in r.stdout, r.stdout + r.stderr


def test_rust_binding_is_generated_from_the_header():
    """rust/src/amd.rs (shipped uncompiled: no rustc in the image): its `extern "C"` block IS the output of
    tools/gen_rust_ffi.py over include/needletail_amd.h - names, arity, parameter types and constness (a u32 <-> u64 or
    *const <-> *mut drift fails here) - and the generator's type mapping is pinned on hand-checked cases."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import gen_rust_ffi as G
    finally:
        sys.path.pop(0)
    assert G.rust_type("const uint8_t *") == "*const u8" and G.rust_type("uint64_t **") == "*mut *mut u64"
    assert G.rust_type("ntk_ctx *const *") == "*const *mut NtkCtx" and G.rust_type("const ntk_comm *") == "*const NtkComm"
    assert G.rust_type("void *") == "*mut c_void" and G.rust_type("const char *") == "*const c_char" and G.rust_type("double *") == "*mut f64"
    fns = {name: (ret, params) for name, ret, params in G.parse_header(open(G.HEADER).read())}
    assert sorted(fns) == _header_symbols()
    assert fns["ntk_comm_init_rank"][1][3] == ("id", "const uint8_t *")          # array parameter decays, const kept
    assert fns["ntk_reduce_device"] == ("int", [("ctx", "ntk_ctx *"), ("d_seq", "const uint8_t *"), ("n_bytes", "uint64_t"), ("p", "const ntk_params *")])
    rs = open(G.RUST).read()
    assert G.block_in_file(rs).group(0) == G.generate(), "rust/src/amd.rs is stale: run python tools/gen_rust_ffi.py --write"
    # every declared function is used or at least visible to the adapters below the block; the struct mirrors keep their layout
    assert "pub struct NtkParams { pub k: u32, pub path: u32, pub pre: u32, pub flags: u32 }" in rs
    assert os.path.exists(os.path.join(ROOT, "rust", "build.rs"))


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md section 1 maps the C ABI to the reference interface it replaces: every symbol the header declares is named there
    (literally, or as a member of an `ntk_prefix_a / b / c` group)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "needletail_amd.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    syms = sorted(set(re.findall(r"\b(ntk_[a-z0-9_]+)\s*\(", hdr)))
    assert len(syms) >= 64
    groups = re.findall(r"`(ntk_[a-z0-9_]+(?: / [a-z0-9_]+)+)`", doc)   # `ntk_reader_open_file / open_memory / next`
    grouped = set()
    for g in groups:
        parts = g.split(" / ")
        head = parts[0]
        grouped.add(head)
        prefix = head[: head.rfind("_") + 1]
        for tail in parts[1:]:
            for cut in range(len(head), 3, -1):   # the shared prefix is some `ntk_..._` of the first member
                if head[cut - 1] == "_" and (head[:cut] + tail) in syms:
                    grouped.add(head[:cut] + tail)
                    break
    wild = [w[:-1] for w in re.findall(r"`(ntk_[a-z0-9_]+\*)`", doc)]      # `ntk_accum_*`
    missing = [s for s in syms if s not in doc and s not in grouped and not any(s.startswith(w) for w in wild)]
    assert not missing, missing
