#!/usr/bin/env python3
"""ubench11: [H S F F] runs its full-rate ops at 2 cycles (ubench9).  Which feature of the real scan2 loop takes that away?
One feature added at a time: exec writes, LDS atomics, the vcc dependency of compare -> select, true data dependencies,
SGPR-writing compares, then a whole synthetic tile with a scalar op after every half-rate op."""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ubench8.hip')).read()
pre = src[:src.index('template <int PAT>')]
pre = pre.replace('#define OPS2', r'''#define HCS(i) "v_cmp_eq_u32_sdwa s[20:21], %[c" #i "], %[d" #i "] src0_sel:BYTE_1 src1_sel:BYTE_1\n"
#define XSd(i) "v_cndmask_b32 %[a" #i "], %[b" #i "], %[c" #i "], vcc\n"       /* select straight after its compare */
#define XXd(i) "v_xor_b32 %[b7], %[b7], %[a" #i "]\n"                          /* xor of the selected word */
#define XMd(i) "v_mad_u64_u32 %[q" #i "], s[26:27], %[a" #i "], 1, %[q" #i "]\n"
#define ABd(i) "v_alignbit_b32 %[c" #i "], %[d" #i "], %[b" #i "], 6\n"
#define FS(i) "v_and_b32 %[a" #i "], s30, %[b" #i "]\n"                        /* full-rate op with an SGPR operand */
#define OPS2''')
pre = pre.replace('"s26", "s27", "s28", "s29"', '"s26", "s27", "s28", "s29", "s30"')
pats = []
VALU = {'F', 'H', 'XC', 'XS', 'XM', 'XX', 'AB', 'DP', 'PM', 'SD', 'AL', 'CNDV', 'MADU64', 'HCS', 'XSd', 'XXd', 'XMd', 'ABd', 'FD', 'FS', 'C'}
def add(name, spec, rept=8):
    cnt = {}; s = []
    for tok in spec:
        i = cnt.get(tok, 0); cnt[tok] = i + 1
        s.append('%s(%d)' % (tok, i % 7))
    pats.append((name, s, len([t for t in spec if t in VALU]), rept))
add('[H S F F] x8 (reference: ideal 2.73)', ['H', 'S', 'F', 'F'] * 8)
add('[H E F F]  scalar = exec write', ['H', 'XE', 'F', 'F'] * 8)
add('[E H S F F]  + exec write', ['XE', 'H', 'S', 'F', 'F'] * 8)
add('[H S F F D]  + LDS atomic', ['H', 'S', 'F', 'F', 'XD'] * 8)
add('[H S F F] x3 + D  (LDS 1 per 9 VALU)', (['H', 'S', 'F', 'F'] * 3 + ['XD']) * 3)
add('[cmp S cndmask(vcc dep) F]', ['XC', 'S', 'XSd', 'F'] * 8)
add('[cmp S F cndmask(vcc dep)]', ['XC', 'S', 'F', 'XSd'] * 8)
add('[H S FD FD]  F reads the H result', ['H', 'S', 'FD', 'FD'] * 8)
add('[mad S F F]', ['XM', 'S', 'F', 'F'] * 8)
add('[cmp_sdwa->sgpr S F F]', ['HCS', 'S', 'F', 'F'] * 8)
add('[H S F(sgpr operand) F]', ['H', 'S', 'FS', 'F'] * 8)
add('[dpp S F F]', ['DP', 'S', 'F', 'F'] * 8)
add('[H S F F] with s_bcnt1/s_add as the scalars', ['H', 'XB', 'F', 'F', 'H', 'XA', 'F', 'F'] * 4)
regB = ['XE', 'XC', 'XB', 'XSd', 'XXd', 'XMd', 'XA']            # one position, no LDS: E cmp bcnt cnd xor mad add
regBD = ['XE', 'XC', 'XB', 'XSd', 'XXd', 'XMd', 'XD', 'XA']
reg0 = ['XE', 'XC', 'XSd', 'XMd', 'XXd', 'XB', 'XA']
add('region, order B, no LDS, true deps (ideal 3.08)', regB * 7, rept=4)
add('region as shipped order, no LDS, true deps', reg0 * 7, rept=4)
add('region, order B, with LDS atomic', regBD * 7, rept=4)
outS = ['AB', 'S'] * 8 + ['DP', 'S'] * 2 + ['PM', 'S'] * 2 + ['SD', 'S'] * 2 + ['AL', 'AL']
out0 = ['AB'] * 8 + ['DP'] * 2 + ['PM'] * 2 + ['SD'] * 2 + ['AL', 'AL'] + ['S'] * 14
add('tile: outside ops with a scalar after each + region B + LDS (ideal 3.46)', (outS + regBD * 4) , rept=8)
add('tile: outside ops then their scalars clustered + region as shipped + LDS', (out0 + (reg0[:-2] + ['XD'] + reg0[-2:]) * 4), rept=8)
add('tile: outside+S, region B, no LDS', (outS + regB * 4), rept=8)
add('outside ops only, scalar after each', outS * 3, rept=4)
add('outside ops only, scalars clustered', out0 * 3, rept=4)
body = ['        %sif constexpr (PAT == %d) asm volatile(".rept %d\\n" %s ".endr\\n s_mov_b64 exec, -1\\n" OPS2);' % ('else ' if i else '', i, p[3], ' '.join(p[1])) for i, p in enumerate(pats)]
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
kern = src[k0:k1].replace('__shared__ uint32_t lds[4096];', '__shared__ uint32_t lds[16384];').replace('s_mov_b32 s29, 0"', 's_mov_b32 s29, 0\\n s_mov_b32 s30, 0xfffc"').replace('"s24", "s25", "s29")', '"s24", "s25", "s29", "s30")')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[2] * p[3]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('{{256, 512}, {256, 1024}, {512, 768}, {512, 1024}}', '{{256, 512}, {512, 512}, {512, 768}}')
rest = rest.replace('printf("%-44s", p.name);', 'printf("%-80s", p.name);').replace('printf("%-44s", "pattern");', 'printf("%-80s", "pattern");')
open(os.path.join(here, 'ubench11.hip'), 'w').write(pre + kern + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body) + '\n    }\n' + rest)
