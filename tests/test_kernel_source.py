"""Static checks of the device source that need no GPU: every inline-asm statement that runs a scalar instruction which writes SCC must
say so in its clobber list.  (An `s_and_b64 exec, A, B` inside a masked region overwrites SCC; without the clobber the compiler is free to
keep a 64-bit add's carry live across the region - `s_add_u32` before it, `s_addc_u32` after it - and the tile offset goes wrong for every
wave that runs more than one tile.  That happened once, in round 5, and only a forced few-block launch in the fuzz saw it.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "needletail_amd", "csrc", f) for f in ("ntk_kernels.hpp", "ntk_tile.hpp", "ntk_api.hip", "ntk_scan2.hip")]
# scalar ALU mnemonics that write SCC (s_mov / s_cselect / s_getreg / s_waitcnt / s_nop do not), and the region macros that expand to them
SCC_WRITERS = re.compile(r"\bs_(and|or|xor|andn2|orn2|nand|nor|xnor|not|add|addc|sub|subb|lshl|lshr|ashr|bcnt[01]|cmp|bitcmp|min|max|abs|bfe|wqm|"
                         r"and_saveexec|or_saveexec|andn2_saveexec|absdiff)\w*\b|NTK_R_EXEC|NTK_R_POS|NTK_R_CNT")


def asm_statements(text):
    i = 0
    while True:
        m = re.search(r"\basm\s*(volatile)?\s*\(", text[i:])
        if not m:
            return
        start = i + m.end()
        depth, j = 1, start
        while depth and j < len(text):
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        yield text[i + m.start():j], text[:i + m.start()].count("\n") + 1
        i = j


def test_asm_blocks_that_write_scc_say_so():
    seen = 0
    for path in SRC:
        text = open(path).read()
        macros = dict(re.findall(r"#define\s+(NTK_\w+)(?:\([^)]*\))?\s+((?:.*\\\n)*.*)", text))
        for stmt, line in asm_statements(text):
            body = stmt
            for _ in range(3):   # expand the region macros used inside the statement (they hold the s_and_b64 / s_bcnt1 / s_add_u32)
                for name, val in macros.items():
                    if name in body:
                        body = body + " " + val
            if SCC_WRITERS.search(body):
                seen += 1
                assert '"scc"' in body, f"{os.path.basename(path)}:{line}: inline asm writes SCC without the clobber:\n{stmt[:300]}"
            # the same for VCC: an instruction string that names vcc (v_cmp ... vcc, v_cndmask ... vcc) needs the "vcc" clobber
            if re.search(r'"[^"]*\bvcc\b[^"]*\\n', body) or re.search(r'"[^"]*\bvcc\b[^"]*"\s*(?::|\))', body):
                if re.search(r'"[^"]*\b(v_cmp|s_bcnt1|v_cndmask)\w*[^"]*\bvcc\b', body):
                    assert '"vcc"' in body, f"{os.path.basename(path)}:{line}: inline asm uses VCC without the clobber:\n{stmt[:300]}"
    assert seen >= 5   # the masked regions of DevMasks2 (canonical, wide, plain, forward-only) and the round-1 regions


def test_the_checker_sees_a_missing_clobber():
    bad = 'asm volatile("s_and_b64 exec, %1, %2\\n v_xor_b32 %0, %0, %3\\n s_mov_b64 exec, -1" : "+v"(x) : "s"(a), "s"(b), "v"(c) : "memory");'
    stmts = list(asm_statements(bad))
    assert len(stmts) == 1 and SCC_WRITERS.search(stmts[0][0]) and '"scc"' not in stmts[0][0]
