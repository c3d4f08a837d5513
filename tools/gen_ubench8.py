#!/usr/bin/env python3
"""ubench8: (1) does ONE scalar instruction between a half-rate op and the full-rate ops that follow it restore their co-issue
([H S F] against [H F S])?  (2) the scan2 masked region as shipped against the same instructions reordered so that every
full-rate op follows a scalar op.  (3) odd waves run only half-rate ops, even waves only full-rate ops: do they overlap?"""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ubench6.hip')).read()
pre = src[:src.index('template <int PAT>')]
pre = pre.replace('#define OPS', r'''#define S0(i) "s_nop 0\n"
#define FD(i) "v_xor_b32 %[a" #i "], %[a" #i "], %[c" #i "]\n"            /* full-rate op reading the result of H(i) */
#define XC(i) "v_cmp_lt_u32 vcc, %[c" #i "], %[d" #i "]\n"
#define XS(i) "v_cndmask_b32 %[a" #i "], %[b" #i "], %[c" #i "], vcc\n"
#define XM(i) "v_mad_u64_u32 %[q" #i "], s[26:27], %[a" #i "], 1, %[q" #i "]\n"
#define XX(i) "v_xor_b32 %[b7], %[b7], %[a" #i "]\n"
#define XD(i) "ds_add_u32 %[d" #i "], %[one]\n"
#define XE(i) "s_and_b64 exec, s[22:23], s[24:25]\n"
#define XB(i) "s_bcnt1_i32_b64 s28, vcc\n"
#define XA(i) "s_add_u32 s29, s29, s28\n"
#define AB(i) "v_alignbit_b32 %[c" #i "], %[d" #i "], %[b" #i "], 6\n"       /* window word: result only read by compares */
#define DP(i) "v_mov_b32_dpp %[c" #i "], %[b" #i "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define PM(i) "v_pk_min_u16 %[a" #i "], %[c" #i "], %[b" #i "] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define SD(i) "v_and_b32_sdwa %[a" #i "], %[b" #i "], %[c" #i "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define AL(i) "v_and_b32 %[a" #i "], 0xfffc, %[b" #i "]\n"
#define OPS''')
pre = pre.replace('"s20", "s21", "s22", "s23", "s24", "s25"', '"s20", "s21", "s26", "s27", "s28", "s29"')
pats = []
VALU = {'F', 'H', 'G', 'X64', 'C', 'FD', 'XC', 'XS', 'XM', 'XX', 'AB', 'DP', 'PM', 'SD', 'AL', 'L', 'R', 'B3', 'M', 'CNDV', 'MADU64'}
def add(name, spec, rept=8):
    cnt = {}; s = []
    for tok in spec:
        i = cnt.get(tok, 0); cnt[tok] = i + 1
        s.append('%s(%d)' % (tok, i % 7 if tok in ('XC', 'XS', 'XM', 'XD') else i % 8))
    pats.append((name, s, len([t for t in spec if t in VALU]), rept))
add('pure F', ['F'] * 32)
add('pure H', ['H'] * 32)
add('[H F S] x16 (F straight after H)', ['H', 'F', 'S'] * 16)
add('[H S F] x16 (scalar op between)', ['H', 'S', 'F'] * 16)
add('[H s_nop F] x16', ['H', 'S0', 'F'] * 16)
add('[H S F F] x12', ['H', 'S', 'F', 'F'] * 12)
add('[H H S F F] x8', ['H', 'H', 'S', 'F', 'F'] * 8)
add('[H S F S] x12', ['H', 'S', 'F', 'S'] * 12)
add('[H S FD] x16 (F reads H result)', ['H', 'S', 'FD'] * 16)
add('[H dsadd F] x16', ['H', 'XD', 'F'] * 16)
add('[H H H S F] x8', ['H', 'H', 'H', 'S', 'F'] * 8)
add('[H S F] x16 with F = v_cndmask vcc', ['H', 'S', 'CNDV'] * 16)
add('[cmp S cndmask(dep)] x16', ['XC', 'S', 'XS'] * 16)
add('[cmp cndmask(dep) S] x16', ['XC', 'XS', 'S'] * 16)
cur = ['XE', 'XC', 'XS', 'XM', 'XX', 'XD', 'XB', 'XA']
rb = ['XE', 'XC', 'XB', 'XS', 'XX', 'XM', 'XD', 'XA']
rc = ['XE', 'XC', 'XB', 'XA', 'XS', 'XX', 'XD', 'XM']
rd = ['XE', 'XC', 'XB', 'XS', 'XX', 'XA', 'XM', 'XD']
add('region as shipped: E cmp cnd mad xor ds bcnt add', cur * 8, rept=4)
add('region B: E cmp bcnt cnd xor mad ds add', rb * 8, rept=4)
add('region C: E cmp bcnt add cnd xor ds mad', rc * 8, rept=4)
add('region D: E cmp bcnt cnd xor add mad ds', rd * 8, rept=4)
out4 = ['AB'] * 8 + ['DP'] * 2 + ['PM'] * 2 + ['SD'] * 2 + ['AL'] * 2   # the work outside the region for four positions (window words, cells)
add('4 positions as shipped (outside ops + region)', (out4 + cur * 4) * 2, rept=4)
add('4 positions, region B', (out4 + rb * 4) * 2, rept=4)
out4b = ['AB'] * 8 + ['DP'] * 2 + ['PM'] * 2 + ['SD'] * 2 + ['S', 'AL', 'AL']
add('4 positions, region B, scalar before the two ANDs', (out4b + rb * 4) * 2, rept=4)
body = ['        %sif constexpr (PAT == %d) asm volatile(".rept %d\\n" %s ".endr\\n s_mov_b64 exec, -1\\n" OPS2);' % ('else ' if i else '', i, p[3], ' '.join(p[1])) for i, p in enumerate(pats)]
npat = len(pats)
body.append('        else if constexpr (PAT == %d) { if (threadIdx.x & 64) asm volatile(".rept 16\\n" ALL8(H) ALL8(H) ".endr\\n" OPS2); else asm volatile(".rept 16\\n" ALL8(F) ALL8(F) ".endr\\n" OPS2); }' % npat)
pats.append(('odd waves pure H, even waves pure F', [], 16, 16))
body.append('        else if constexpr (PAT == %d) { if (threadIdx.x & 64) asm volatile(".rept 16\\n" ALL8(H) ALL8(H) ".endr\\n" OPS2); else asm volatile(".rept 16\\n" ALL8(F) ALL8(F) ALL8(F) ALL8(F) ".endr\\n" OPS2); }' % (npat + 1))
pats.append(('odd waves 16 H, even waves 32 F per trip (counted 24)', [], 24, 16))
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
kern = src[k0:k1].replace('lds[threadIdx.x] = 0; __syncthreads();', 'lds[threadIdx.x] = 0; __syncthreads();\n    const uint32_t one = 1;\n    asm volatile("s_mov_b64 s[22:23], -1\\n s_mov_b64 s[24:25], -1\\n s_mov_b32 s29, 0" ::: "s22", "s23", "s24", "s25", "s29");')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[2] * p[3]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('{{256, 256}, {256, 512}, {256, 768}, {256, 1024}, {512, 768}, {512, 1024}}', '{{256, 512}, {256, 1024}, {512, 768}, {512, 1024}}')
ops2 = '#define OPS2 ' + pre[pre.index('#define OPS') + len('#define OPS'):].split('\n#define ALL8')[0].replace('    : : "vcc"', '    : [one] "v"(one) : "vcc", "memory"')
pre = pre.replace('#define ALL8(X)', ops2 + '\n#define ALL8(X)')
open(os.path.join(here, 'ubench8.hip'), 'w').write(pre + kern + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body) + '\n    }\n' + rest)
