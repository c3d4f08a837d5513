#!/usr/bin/env python3
"""ubench12: cycles per POSITION of candidate formulations of the scan2 masked region (true dependencies: compare -> select ->
digests; no loads), 6 waves per SIMD.  V0 = as shipped.  What does each SGPR-writing VALU op (v_cmp -> vcc, v_mad_u64_u32's carry)
really cost next to full-rate ops, and what do the alternatives cost?"""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ubench8.hip')).read()
pre = src[:src.index('template <int PAT>')]
pre = pre.replace('#define OPS2', r'''#define PE(i) "s_and_b64 exec, s[22:23], s[24:25]\n"
#define PC(i) "v_cmp_lt_u32 vcc, %[c" #i "], %[d" #i "]\n"
#define PS(i) "v_cndmask_b32 %[a" #i "], %[b" #i "], %[c" #i "], vcc\n"
#define PM(i) "v_mad_u64_u32 %[q" #i "], s[26:27], %[a" #i "], 1, %[q" #i "]\n"
#define PMV(i) "v_mad_u64_u32 %[q" #i "], vcc, %[a" #i "], 1, %[q" #i "]\n"
#define PX(i) "v_xor_b32 %[b7], %[b7], %[a" #i "]\n"
#define PD(i) "ds_add_u32 %[d" #i "], %[one]\n"
#define PB(i) "s_bcnt1_i32_b64 s28, vcc\n"
#define PA(i) "s_add_u32 s29, s29, s28\n"
#define PS2(i) "v_cndmask_b32 v" #i "0, %[b" #i "], %[c" #i "], vcc\n"        /* select into the low half of the pair v[i0:i1] (v_i1 = 0) */
#define PL(i) "v_lshl_add_u64 %[q" #i "], v[" #i "0:" #i "1], 0, %[q" #i "]\n"
#define PX2(i) "v_xor_b32 %[b7], %[b7], v" #i "0\n"
#define PN(i) "v_and_b32 %[c7], 0xfffffff, %[a" #i "]\n"                      /* 28 low bits, then a 32-bit add: full-rate sum */
#define PU(i) "v_add_u32 %[a7], %[a7], %[c7]\n"
#define PUF(i) "v_add_u32 %[a7], %[a7], %[a" #i "]\n"                         /* 32-bit wrapping sum only */
#define PMIN(i) "v_min_u32 %[a" #i "], %[c" #i "], %[d" #i "]\n"
#define PCE(i) "v_cmp_lt_u32_e64 s[20:21], %[c" #i "], %[d" #i "]\n"
#define PSE(i) "v_cndmask_b32_e64 %[a" #i "], %[b" #i "], %[c" #i "], s[20:21]\n"
#define PBE(i) "s_bcnt1_i32_b64 s28, s[20:21]\n"
#define PSUB(i) "v_sub_u32 %[a" #i "], %[c" #i "], %[d" #i "]\n"
#define PASH(i) "v_ashrrev_i32 %[a" #i "], 31, %[a" #i "]\n"
#define PBO(i) "v_bitop3_b32 %[a" #i "], %[a" #i "], %[b" #i "], %[c" #i "] bitop3:0xca\n"
#define PNF(i) "v_sub_u32 %[d7], %[d7], %[a" #i "]\n"
#define OPS2''')
pre = pre.replace('"s26", "s27", "s28", "s29"', '"s26", "s27", "s28", "s29", "v10", "v11", "v20", "v21", "v30", "v31", "v40", "v41"')
variants = [
    ('V0 shipped: E cmp cnd mad xor ds bcnt add', ['PE', 'PC', 'PS', 'PM', 'PX', 'PD', 'PB', 'PA']),
    ('V0 without the LDS atomic', ['PE', 'PC', 'PS', 'PM', 'PX', 'PB', 'PA']),
    ('order B: E cmp bcnt cnd xor mad add (no LDS)', ['PE', 'PC', 'PB', 'PS', 'PX', 'PM', 'PA']),
    ('no exec write: cmp cnd mad xor bcnt add', ['PC', 'PS', 'PM', 'PX', 'PB', 'PA']),
    ('no count: E cmp cnd mad xor', ['PE', 'PC', 'PS', 'PM', 'PX']),
    ('mad with vcc as its carry-out: E cmp bcnt cnd mad(vcc) xor add', ['PE', 'PC', 'PB', 'PS', 'PMV', 'PX', 'PA']),
    ('lshl_add_u64 on a (t:0) pair: E cmp bcnt cnd lshladd xor add', ['PE', 'PC', 'PB', 'PS2', 'PL', 'PX2', 'PA']),
    ('32-bit sum of 28-bit words: E cmp bcnt cnd and add xor add', ['PE', 'PC', 'PB', 'PS', 'PN', 'PU', 'PX', 'PA']),
    ('32-bit wrapping sum: E cmp bcnt cnd add xor add', ['PE', 'PC', 'PB', 'PS', 'PUF', 'PX', 'PA']),
    ('no sum at all: E cmp bcnt cnd xor add', ['PE', 'PC', 'PB', 'PS', 'PX', 'PA']),
    ('no select/digests: E cmp bcnt add', ['PE', 'PC', 'PB', 'PA']),
    ('cmp into s[20:21] (VOP3), cnd_e64: E cmpE bcnt cndE mad xor add', ['PE', 'PCE', 'PBE', 'PSE', 'PM', 'PX', 'PA']),
    ('arithmetic select (31-bit T): E sub ashr bitop3 nf-=m and add xor', ['PE', 'PSUB', 'PASH', 'PNF', 'PBO', 'PN', 'PU', 'PX']),
    ('min only: E min mad xor', ['PE', 'PMIN', 'PM', 'PX']),
    ('min + full-rate sum: E min and add xor', ['PE', 'PMIN', 'PN', 'PU', 'PX']),
]
body = []; pats = []
for i, (nm, pos) in enumerate(variants):
    s = []
    for j in range(4):
        for t in pos: s.append('%s(%d)' % (t, j + 1))
    body.append('        %sif constexpr (PAT == %d) asm volatile("v_mov_b32 v11, 0\\n v_mov_b32 v21, 0\\n v_mov_b32 v31, 0\\n v_mov_b32 v41, 0\\n .rept 16\\n" %s ".endr\\n s_mov_b64 exec, -1\\n" OPS2);' % ('else ' if i else '', i, ' '.join(s)))
    pats.append((nm, 4 * 16))
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
kern = src[k0:k1].replace('__shared__ uint32_t lds[4096];', '__shared__ uint32_t lds[16384];')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[1]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('{{256, 512}, {256, 1024}, {512, 768}, {512, 1024}}', '{{256, 512}, {512, 512}, {512, 768}}')
rest = rest.replace('printf("%-44s", p.name);', 'printf("%-80s", p.name);').replace('printf("%-44s", "pattern");', 'printf("%-80s", "formulation (cycles per POSITION per SIMD)");')
open(os.path.join(here, 'ubench12.hip'), 'w').write(pre + kern + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body) + '\n    }\n' + rest)
