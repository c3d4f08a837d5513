"""Static checks of the device source that need no GPU.  (1) Every inline-asm statement that runs a scalar instruction which writes SCC must
say so in its clobber list; the same for VCC.  (2) Round 6: every statement that narrows exec restores it (its LAST instruction is
`s_mov_b64 exec, -1`) and is a memory barrier for the compiler ("memory": it holds LDS atomics the compiler cannot see, and code moved across
a narrowed exec mask would run on a subset of the lanes).  (3) Round 6: a kernel whose asm statements issue LDS operations waits for them
(`s_waitcnt lgkmcnt(0)`, by hand: the compiler does not count what it cannot see) after its tile loop and before it reads the histogram.
Each rule comes with a test that the checker FAILS on a deliberately broken copy.  Rule (1):  (An `s_and_b64 exec, A, B` inside a masked region overwrites SCC; without the clobber the compiler is free to
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


# ---- round 6: exec restore + "memory", and the hand-written LDS wait -------------------------------------------------------------------

MNEMONIC = re.compile(r"\b((?:s|v|ds|buffer)_[a-z0-9_]+)\b")
EXEC_WRITE = re.compile(r"\bs_(?:and|or|xor|andn2|mov|not)\w*_b64\s+exec\b|\bs_\w+_saveexec_b64\b|\bv_cmpx_")


def macro_table(text):
    """name -> body of every NTK_* function-like or object-like macro; of several definitions (#ifdef ablation / #else product) the LAST one
    is the product's (the ablation branch comes first in every such pair of ntk_kernels.hpp)."""
    return dict(re.findall(r"#define\s+(NTK_\w+)(?:\([^)]*\))?\s+((?:.*\\\n)*.*)", text))


def expand_in_place(stmt, macros, depth=8):
    """The statement with its macros expanded where they stand (arguments are not substituted: only the ORDER of the instructions matters)."""
    for _ in range(depth):
        changed = False
        for name, val in macros.items():
            new = re.sub(r"\b" + name + r"\b(\([^()]*\))?", lambda m: " " + val.replace("\\\n", " ") + " ", stmt)
            if new != stmt:
                stmt, changed = new, True
        if not changed:
            break
    return stmt


def instruction_strings(expanded):
    """The statement's string literals up to the operand lists, concatenated (what the assembler sees, macro stringification aside)."""
    out = []
    for lit in re.findall(r'"((?:[^"\\]|\\.)*)"', expanded):
        if MNEMONIC.search(lit) or lit.strip() in ("", "\\n"):
            out.append(lit)
    return " ".join(out)


def check_exec_regions(text, name="source"):
    """Rule (2).  Returns the number of exec-writing statements seen; raises AssertionError on a violation."""
    macros = macro_table(text)
    seen = 0
    for stmt, line in asm_statements(text):
        body = instruction_strings(expand_in_place(stmt, macros))
        if not EXEC_WRITE.search(body):
            continue
        seen += 1
        ms = list(MNEMONIC.finditer(body))
        last = body[ms[-1].start():][:40] if ms else ""
        assert ms and ms[-1].group(1) == "s_mov_b64" and re.match(r"\s+exec,\s*-1", body[ms[-1].end():]), \
            f"{name}:{line}: a region that narrows exec must end by restoring it, ends with: {last!r}"
        full = expand_in_place(stmt, macros)
        assert '"memory"' in full, f"{name}:{line}: a region that narrows exec must clobber \"memory\""
    return seen


def kernel_bodies(text):
    """(name, body) of every __global__ function of the text."""
    for m in re.finditer(r"__global__[^;{]*?void\s+(\w+)\s*\(", text):
        i = text.index("{", m.end())
        depth, j = 1, i + 1
        while depth and j < len(text):
            depth += {"{": 1, "}": -1}.get(text[j], 0)
            j += 1
        yield m.group(1), text[i:j]


def check_lds_waits(text, name="source"):
    """Rule (3).  A kernel that reaches inline-asm LDS operations (directly or through the region emitters of DevMasks2 / DevMinSink) must
    hold `s_waitcnt lgkmcnt(0)` in an asm statement of its own after the tile loop (the last `next = ...readfirstlane(next)` of the pull
    loop) and before the first READ of its LDS histogram.  Returns how many kernels the rule applied to."""
    macros = macro_table(text)
    emitters = set()   # struct / function names whose bodies hold asm with ds_ instructions
    for m in re.finditer(r"\bstruct\s+(\w+)\s*\{", text):
        i = m.end()
        depth, j = 1, i
        while depth and j < len(text):
            depth += {"{": 1, "}": -1}.get(text[j], 0)
            j += 1
        body = text[i:j]
        if any("ds_" in instruction_strings(expand_in_place(st, macros)) for st, _ in asm_statements(body)):
            emitters.add(m.group(1))
    applied = 0
    for kname, body in kernel_bodies(text):
        own = any("ds_" in instruction_strings(expand_in_place(st, macros)) for st, _ in asm_statements(body))
        if not own and not any(re.search(r"\b" + e + r"\b", body) for e in emitters):
            continue
        applied += 1
        loop_end = max((m.end() for m in re.finditer(r"next\s*=\s*__builtin_amdgcn_readfirstlane\(next\)", body)), default=-1)
        assert loop_end >= 0, f"{name}: {kname}: no pull loop found"
        wait = re.search(r'asm\s+volatile\s*\(\s*"s_waitcnt lgkmcnt\(0\)"\s*:::\s*"memory"\s*\)', body[loop_end:])
        read = re.search(r"[=+(,]\s*s_hist\[", body[loop_end:])
        assert wait, f"{name}: {kname}: no s_waitcnt lgkmcnt(0) after the tile loop (the asm regions' LDS atomics are invisible to the compiler)"
        assert read is None or wait.start() < read.start(), f"{name}: {kname}: the LDS histogram is read before the hand-written wait"
    return applied


def test_exec_regions_restore_exec_and_clobber_memory():
    seen = sum(check_exec_regions(open(path).read(), os.path.basename(path)) for path in SRC)
    assert seen >= 6   # emit_canon, emit_canon_wide, emit_word, region_plain x 2, emit_fwd (wide), the generic minimizer sink


def test_kernels_wait_for_their_asm_lds_operations():
    applied = sum(check_lds_waits(open(path).read(), os.path.basename(path)) for path in SRC)
    assert applied >= 2   # scan2_kernel, minimizer_scan_kernel


def test_the_checker_sees_a_region_that_does_not_restore_exec():
    good = 'asm volatile("s_and_b64 exec, %1, %2\\n v_xor_b32 %0, %0, %3\\n s_mov_b64 exec, -1\\n" : "+v"(x) : "s"(a), "s"(b), "v"(c) : "memory", "scc");'
    assert check_exec_regions(good) == 1
    for bad in (good.replace(' s_mov_b64 exec, -1\\n', ''),                                     # never restored
                good.replace(' s_mov_b64 exec, -1\\n', ' s_mov_b64 exec, -1\\n v_mov_b32 %0, 0\\n'),   # an instruction after the restore
                good.replace('"memory", ', '')):                                                # no compiler barrier
        try:
            check_exec_regions(bad)
        except AssertionError:
            continue
        raise AssertionError("the checker accepted: " + bad)
    # through macros, as the product writes its regions
    macro_src = ('#define NTK_T_EXEC(i) "s_and_b64 exec, %[A" #i "], %[B" #i "]\\n"\n#define NTK_T_POS(i) NTK_T_EXEC(i) "v_add_u32 %0, %0, %1\\n"\n'
                 'void f() { asm volatile(NTK_T_POS(0) NTK_T_POS(1) XXX : "+v"(x) : "v"(y) : "memory", "scc"); }')
    assert check_exec_regions(macro_src.replace("XXX", '"s_mov_b64 exec, -1\\n"')) == 1
    try:
        check_exec_regions(macro_src.replace("XXX", ""))
    except AssertionError:
        pass
    else:
        raise AssertionError("the checker accepted a macro-built region without the restore")


def test_the_checker_sees_a_missing_lds_wait():
    kern = ('struct Sink { void emit() { asm volatile("s_and_b64 exec, %0, %1\\n ds_add_u32 %2, %3\\n s_mov_b64 exec, -1\\n" :: "s"(a), "s"(b), "v"(c), "v"(d) : "memory", "scc"); } };\n'
            '__global__ void k(Args a) { Sink s; while (x) { s.emit(); next = __builtin_amdgcn_readfirstlane(next); }\n WAIT\n for (int c = 0; c < 4096; c++) tot += s_hist[c]; }')
    assert check_lds_waits(kern.replace("WAIT", 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')) == 1
    for bad in (kern.replace("WAIT", ""),
                kern.replace("WAIT\n for (int c = 0; c < 4096; c++) tot += s_hist[c];", 'tot += s_hist[0]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')):
        try:
            check_lds_waits(bad)
        except AssertionError:
            continue
        raise AssertionError("the checker accepted: " + bad)
