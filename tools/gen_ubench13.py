#!/usr/bin/env python3
"""ubench13: candidate region formulations INSIDE a synthetic tile (the 16 outside ops of four positions - 8 alignbit, 2 DPP, 2 pk_min,
2 SDWA and, 2 and - then four positions of the region with true dependencies and the LDS atomic), 6 waves per SIMD.
Cycles per group of FOUR positions per SIMD (the shipped kernel: 709 / 4 = 177 per four positions including encode and validity)."""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ubench12.hip')).read()
pre = src[:src.index('template <int PAT>')]
pre = pre.replace('#define OPS2', r'''#define QC(i) "v_cmp_lt_u32_e64 s[" #i "0:" #i "1], %[c" #i "], %[d" #i "]\n"          /* compare under the full exec mask into its own SGPR pair (i = 4..7 -> s[40:41] ..) */
#define QS(i) "v_cndmask_b32_e64 %[a" #i "], %[b" #i "], %[c" #i "], s[" #i "0:" #i "1]\n"
#define QF(i) "s_and_b64 s[20:21], exec, s[" #i "0:" #i "1]\n s_bcnt1_i32_b64 s28, s[20:21]\n"
#define OPS2''')
pre = pre.replace('"v10", "v11"', '"s40", "s41", "s50", "s51", "s60", "s61", "s70", "s71", "v10", "v11"')
outS = ['AB', 'S'] * 8 + ['DP', 'S'] * 2 + ['PKM', 'S'] * 2 + ['SD', 'S'] * 2 + ['AL', 'AL']
pre = pre.replace('#define OPS2', '#define PKM(i) "v_pk_min_u16 %[a" #i "], %[c" #i "], %[b" #i "] op_sel:[0,1] op_sel_hi:[1,0]\\n"\n#define OPS2')
def region(pos, idx):
    return ['%s(%d)' % (t, idx) for t in pos]
variants = [
    ('shipped region: E cmp cnd mad xor ds bcnt add', None, ['PE', 'PC', 'PS', 'PM', 'PX', 'PD', 'PB', 'PA']),
    ('order B: E cmp bcnt cnd xor mad ds add', None, ['PE', 'PC', 'PB', 'PS', 'PX', 'PM', 'PD', 'PA']),
    ('lshl_add_u64 pair sum: E cmp bcnt cnd lshladd xor ds add', None, ['PE', 'PC', 'PB', 'PS2', 'PL', 'PX2', 'PD', 'PA']),
    ('28-bit words, 32-bit sum: E cmp bcnt cnd and add xor ds add', None, ['PE', 'PC', 'PB', 'PS', 'PN', 'PU', 'PX', 'PD', 'PA']),
    ('no sum (ablation): E cmp bcnt cnd xor ds add', None, ['PE', 'PC', 'PB', 'PS', 'PX', 'PD', 'PA']),
    ('no count (ablation): E cmp cnd mad xor ds', None, ['PE', 'PC', 'PS', 'PM', 'PX', 'PD']),
    ('no exec write (ablation): cmp cnd mad xor ds bcnt add', None, ['PC', 'PS', 'PM', 'PX', 'PD', 'PB', 'PA']),
    ('no LDS (ablation): E cmp cnd mad xor bcnt add', None, ['PE', 'PC', 'PS', 'PM', 'PX', 'PB', 'PA']),
    ('min only (ablation): E min mad xor ds', None, ['PE', 'PMIN', 'PM', 'PX', 'PD']),
    ('min + 32-bit sum (ablation): E min and add xor ds', None, ['PE', 'PMIN', 'PN', 'PU', 'PX', 'PD']),
    ('arithmetic select: E sub ashr nf-=m bitop3 and add xor ds', None, ['PE', 'PSUB', 'PASH', 'PNF', 'PBO', 'PN', 'PU', 'PX', 'PD']),
    ('compares outside (own SGPR pairs), region: E cndE mad xor ds and+bcnt add', ['QC'], ['PE', 'QS', 'PM', 'PX', 'PD', 'QF', 'PA']),
    ('compares outside, 28-bit 32-bit sums: E cndE and add xor ds and+bcnt add', ['QC'], ['PE', 'QS', 'PN', 'PU', 'PX', 'PD', 'QF', 'PA']),
    ('outside ops only', None, []),
]
body = []; pats = []
for i, (nm, outside_extra, pos) in enumerate(variants):
    cnt = {}; s = []
    for t in outS:
        j = cnt.get(t, 0); cnt[t] = j + 1
        s.append('%s(%d)' % (t, j % 7))
    if outside_extra:
        for idx in (4, 5, 6, 7):
            for t in outside_extra: s.append('%s(%d)' % (t, idx))
        for idx in (4, 5, 6, 7): s += region(pos, idx)
    else:
        for idx in (1, 2, 3, 4): s += region(pos, idx)
    body.append('        %sif constexpr (PAT == %d) asm volatile("v_mov_b32 v11, 0\\n v_mov_b32 v21, 0\\n v_mov_b32 v31, 0\\n v_mov_b32 v41, 0\\n .rept 16\\n" %s ".endr\\n s_mov_b64 exec, -1\\n" OPS2);' % ('else ' if i else '', i, ' '.join(s)))
    pats.append((nm, 16))
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[1]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('"formulation (cycles per POSITION per SIMD)"', '"formulation (cycles per FOUR positions per SIMD)"')
open(os.path.join(here, 'ubench13.hip'), 'w').write(pre + src[k0:k1] + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body) + '\n    }\n' + rest)
