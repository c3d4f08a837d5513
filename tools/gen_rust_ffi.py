#!/usr/bin/env python3
"""gen_rust_ffi.py - the `extern "C"` block of rust/src/amd.rs, generated from include/needletail_amd.h.

    python tools/gen_rust_ffi.py            prints the block
    python tools/gen_rust_ffi.py --write    rewrites the block inside rust/src/amd.rs in place

tests/test_abi.py::test_rust_binding_is_generated_from_the_header compares the block in the file with this output, so a
drift in parameter TYPES or constness (u32 <-> u64, *const <-> *mut) fails the CPU suite, not only names and arity.
The C subset the header uses is small and fixed: scalar typedefs, pointers (one or two levels, const on either),
array parameters (decay to pointers), opaque handle structs and the four plain structs."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "needletail_amd.h")
RUST = os.path.join(ROOT, "rust", "src", "amd.rs")

SCALARS = {"int": "c_int", "double": "f64", "char": "c_char", "void": "c_void", "uint8_t": "u8", "uint16_t": "u16",
           "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "int32_t": "i32", "size_t": "usize"}
STRUCTS = {"ntk_ctx": "NtkCtx", "ntk_batch": "NtkBatch", "ntk_reader": "NtkReader", "ntk_comm": "NtkComm",
           "ntk_params": "NtkParams", "ntk_result": "NtkResult", "ntk_record": "NtkRecord", "ntk_gunzip_info": "NtkGunzipInfo"}


def rust_type(ctype: str) -> str:
    """`const uint8_t *` -> `*const u8`; `ntk_ctx *const *` -> `*const *mut NtkCtx`; `uint64_t **` -> `*mut *mut u64`."""
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", ctype)
    base, base_const, i = None, False, 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        elif toks[i] in ("struct", "unsigned"):
            pass
        else:
            base = toks[i]
        i += 1
    if base in SCALARS:
        t = SCALARS[base]
    elif base in STRUCTS:
        t = STRUCTS[base]
    else:
        raise ValueError(f"type {ctype!r}: unknown base {base!r}")
    # pointer levels, innermost first; a `const` AFTER a star qualifies that pointer, i.e. what the NEXT star points to
    const_here = base_const
    while i < len(toks):
        assert toks[i] == "*", ctype
        t = ("*const " if const_here else "*mut ") + t
        const_here = False
        i += 1
        while i < len(toks) and toks[i] == "const":
            const_here = True
            i += 1
    return t


def parse_header(text: str):
    """[(name, return C type, [(param name, param C type), ...]), ...] in header order."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = []
    for ret, name, args in re.findall(r"(?m)^\s*([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(ntk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        params = []
        args = " ".join(args.split())
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?$", a)
                if not m:
                    raise ValueError(f"{name}: cannot parse parameter {a!r}")
                ctype, pname, arr = m.group(1).strip(), m.group(2), m.group(3)
                if arr:
                    ctype += " *"   # array parameters decay
                params.append((pname, ctype))
        out.append((name, " ".join(ret.split()), params))
    return out


def generate() -> str:
    lines = ['extern "C" {']
    for name, ret, params in parse_header(open(HEADER).read()):
        ps = ", ".join(f"{p}: {rust_type(t)}" for p, t in params)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        lines.append(f"    pub fn {name}({ps}){r};")
    lines.append("}")
    return "\n".join(lines) + "\n"


def block_in_file(text: str):
    m = re.search(r'(?ms)^extern "C" \{\n.*?^\}\n', text)
    if not m:
        raise ValueError('rust/src/amd.rs has no `extern "C" { ... }` block')
    return m


def main():
    block = generate()
    if "--write" in sys.argv[1:]:
        text = open(RUST).read()
        m = block_in_file(text)
        open(RUST, "w").write(text[: m.start()] + block + text[m.end():])
        print(f"rewrote the extern block of {os.path.relpath(RUST, ROOT)} ({block.count(chr(10)) - 2} functions)")
    else:
        sys.stdout.write(block)


if __name__ == "__main__":
    main()
