#!/usr/bin/env python3
"""ubench10: the k = 21 scan2 tile loop AS THE COMPILER SCHEDULED IT (instruction classes taken from the ISA listing, operands replaced
by pooled scratch registers: no true dependencies, no loads) against the same multiset of instructions in other orders.  Question:
is the 4.04 cycles per VALU instruction of the shipped loop a property of the instruction ORDER (full-rate ops that never get their
2-cycle issue because of how half-rate and scalar ops surround them)?"""
import os, re, subprocess, sys, tempfile
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
K = 21
asm_path = '/tmp/isa/k.s'
if not os.path.exists(asm_path):
    os.makedirs('/tmp/isa', exist_ok=True)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-DNTK_KB_FIX', '-DNTK_KB_SV', '-DNTK_KB_SV2', '-DNTK_KB_HB=14', '-mllvm',
                           '-amdgpu-sched-strategy=iterative-ilp', '-S', '--cuda-device-only', '-o', asm_path, os.path.join(here, 'kbench.hip')], stderr=subprocess.DEVNULL)
text = open(asm_path).read()
name = f'_ZN3ntk12scan2_kernelILi{K}ELb1ELb1ELb0ELi14ELi0ELb0EEEvNS_8ScanArgsE'
body = text[text.index(name + ':'):]
body = body[:body.index('.end_amdhsa_kernel')].splitlines()
def block(label_re, stop_re):
    i = next(i for i, l in enumerate(body) if re.match(label_re, l))
    out = []
    for l in body[i + 1:]:
        if re.match(stop_re, l): break
        out.append(l)
    return out
hdr = block(r'^\.LBB\d+_33:', r'^; %bb\.34:')
b35 = block(r'^\.LBB\d+_35:', r'^\.LBB\d+_37:')
b32 = block(r'^\.LBB\d+_32:', r'^\.LBB\d+_33:')
seq = []
for l in hdr + b35 + b32:
    m = re.match(r'^\s*([vs]_\w+|ds_\w+|buffer_\w+)', l)
    if m: seq.append((m.group(1), l.strip()))
FULL = {'v_and_b32_e32', 'v_and_b32', 'v_xor_b32', 'v_xor_b32_e32', 'v_add_u32', 'v_add_u32_e32', 'v_lshrrev_b32_e32', 'v_bitop3_b32', 'v_cndmask_b32', 'v_mov_b32_e32'}
def cls(mn, line):
    if mn.startswith('buffer_') or mn in ('s_waitcnt', 's_branch') or mn.startswith('s_cbranch'): return None
    if mn.startswith('ds_'): return 'D'
    if mn.startswith('s_'): return 'E' if ' exec' in line.split(',')[0] else 'S'
    if mn.startswith('v_cmp') and 'sdwa' in mn: return 'HCS'
    if mn.startswith('v_cmp'): return 'HC'
    if mn == 'v_cndmask_b32': return 'FC'
    if mn.startswith('v_mad_u64'): return 'HM'
    if mn in FULL: return 'F'
    return 'H'
toks = [c for c in (cls(m, l) for m, l in seq) if c]
from collections import Counter
print('tile loop classes:', Counter(toks), file=sys.stderr)
EMIT = {
    'H': lambda i: 'H(%d)' % (i % 8), 'F': lambda i: 'F(%d)' % (i % 8), 'S': lambda i: 'S(0)', 'E': lambda i: 'XE(0)', 'D': lambda i: 'XD(%d)' % (i % 7),
    'HC': lambda i: 'XC(%d)' % (i % 7), 'HCS': lambda i: 'HCS(%d)' % (i % 8), 'FC': lambda i: 'CNDV(%d)' % (i % 8), 'HM': lambda i: 'MADU64(%d)' % (i % 8),
}
VAL = {'H', 'F', 'HC', 'HCS', 'FC', 'HM'}
def emit(ts):
    cnt = {}; out = []
    for t in ts:
        i = cnt.get(t, 0); cnt[t] = i + 1
        out.append(EMIT[t](i))
    return ' '.join(out)
def is_h(t): return t in ('H', 'HC', 'HCS', 'HM')
def is_f(t): return t in ('F', 'FC')
variants = [('as scheduled by the compiler', toks)]
# (1) same VALU/LDS order, the scalar ops dealt out evenly: one after every half-rate op while they last
vs = [t for t in toks if t in VAL or t == 'D']; ss = [t for t in toks if t in ('S', 'E')]
out = []; k = 0
nh = sum(1 for t in vs if is_h(t))
for t in vs:
    out.append(t)
    if is_h(t) and k < len(ss): out.append(ss[k]); k += 1
out += ss[k:]
variants.append(('same VALU order, one scalar op after each half-rate op', out))
# (2) half-rate ops (with the scalar ops between them) first, all full-rate ops at the end
hs = [t for t in vs if is_h(t) or t == 'D']; fs = [t for t in vs if is_f(t)]
out = []; k = 0
for t in hs:
    out.append(t)
    if k < len(ss): out.append(ss[k]); k += 1
out += ss[k:] + fs
variants.append(('half-rate ops + scalar ops first, all full-rate ops last', out))
# (3) no scalar ops at all (what the VALU stream alone costs)
variants.append(('VALU + LDS only, compiler order (no scalar ops)', vs))
# (4) every full-rate op made half-rate (floor if none of them ever co-issues) / every half-rate kept, full-rate removed
variants.append(('compiler order, full-rate ops removed (half-rate + scalar + LDS only)', [t for t in toks if not is_f(t)]))
# (5) groups [H S F]: full-rate ops dealt out among the half-rate ops
out = []; k = 0; f = 0
for t in hs:
    out.append(t)
    if k < len(ss): out.append(ss[k]); k += 1
    if t != 'D' and f < len(fs) and (len(out) % 2 == 0): out.append(fs[f]); f += 1
out += ss[k:] + fs[f:]
variants.append(('[H S F] groups: full-rate ops dealt out among the half-rate ops', out))
nvalu = sum(1 for t in toks if t in VAL)
src = open(os.path.join(here, 'ubench8.hip')).read()
pre = src[:src.index('template <int PAT>')]
pre = pre.replace('#define OPS2', '#define HCS(i) "v_cmp_eq_u32_sdwa s[20:21], %[c" #i "], %[d" #i "] src0_sel:BYTE_1 src1_sel:BYTE_1\\n"\n#define OPS2')
body_lines = []
pats = []
for i, (nm, ts) in enumerate(variants):
    nv = sum(1 for t in ts if t in VAL)
    body_lines.append('        %sif constexpr (PAT == %d) asm volatile(".rept 4\\n" %s ".endr\\n s_mov_b64 exec, -1\\n" OPS2);' % ('else ' if i else '', i, emit(ts)))
    pats.append((nm, nv * 4))
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
kern = src[k0:k1].replace('__shared__ uint32_t lds[4096];', '__shared__ uint32_t lds[16384];')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[1]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('{{256, 512}, {256, 1024}, {512, 768}, {512, 1024}}', '{{256, 256}, {256, 512}, {512, 512}, {512, 768}}')
rest = rest.replace('printf("%-44s", p.name);', 'printf("%-76s", p.name);').replace('printf("%-44s", "pattern");', 'printf("%-76s", "pattern (cycles per VALU instruction; %d VALU per tile)");' .replace('%d', str(nvalu)))
open(os.path.join(here, 'ubench10.hip'), 'w').write(pre + kern + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body_lines) + '\n    }\n' + rest)
