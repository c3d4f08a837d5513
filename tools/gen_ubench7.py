#!/usr/bin/env python3
"""ubench7: after ONE half-rate op a stream of full-rate ops runs at 4 cycles each (ubench6).  Is there a cheap instruction that
brings it back to 2 (i.e. re-pairs the two waves that co-issue)?  Pattern: [H, X, 32 F] for many X."""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ubench6.hip')).read()
pre = src[:src.index('template <int PAT>')]
pre = pre.replace('#define OPS', '''#define BAR(i) "s_barrier\\n"
#define SLP0(i) "s_sleep 0\\n"
#define SLP1(i) "s_sleep 1\\n"
#define NOP7(i) "s_nop 7\\n"
#define WAIT0(i) "s_waitcnt vmcnt(0) lgkmcnt(0)\\n"
#define PRIO(i) "s_setprio 1\\n s_setprio 0\\n"
#define VNOP(i) "v_nop\\n"
#define MEMT(i) "s_memtime s[26:27]\\n s_waitcnt lgkmcnt(0)\\n"
#define GETREG(i) "s_getreg_b32 s26, hwreg(HW_REG_HW_ID)\\n"
#define DSRD(i) "ds_read_b32 %[c7], %[d7]\\n s_waitcnt lgkmcnt(0)\\n"
#define RFL(i) "v_readfirstlane_b32 s26, %[a7]\\n"
#define BRN(i) "s_cmp_eq_u32 s26, s26\\n s_cbranch_scc0 1f\\n 1:\\n"
#define BRT(i) "s_branch 1f\\n s_nop 0\\n 1:\\n"
#define EXECW(i) "s_mov_b64 exec, -1\\n"
#define SETH(i) "s_sethalt 0\\n"
#define WAKE(i) "s_wakeup\\n"
#define ICINV(i) "s_nop 0\\n"
#define OPS''')
pre = pre.replace('"s20", "s21", "s22", "s23", "s24", "s25"', '"s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"')
pats = []
def add(name, spec, rept=8):
    cnt = {}; s = []
    for tok in spec:
        i = cnt.get(tok, 0); cnt[tok] = i + 1
        s.append('%s(%d)' % (tok, i % 8))
    nvalu = len([t for t in spec if t in ('F', 'H', 'G', 'X64', 'C')])
    pats.append((name, s, nvalu, rept))
add('pure F', ['F'] * 33)
add('H + 32 F', ['H'] + ['F'] * 32)
for x in ['BAR', 'SLP0', 'SLP1', 'NOP7', 'WAIT0', 'PRIO', 'VNOP', 'MEMT', 'GETREG', 'DSRD', 'RFL', 'BRN', 'BRT', 'EXECW', 'WAKE']:
    add('H + %s + 32 F' % x, ['H', x] + ['F'] * 32)
add('H H + 32 F', ['H', 'H'] + ['F'] * 32)
add('8 H + 32 F', ['H'] * 8 + ['F'] * 32)
add('8 H + BAR + 32 F', ['H'] * 8 + ['BAR'] + ['F'] * 32)
add('8 H + SLP0 + 32 F', ['H'] * 8 + ['SLP0'] + ['F'] * 32)
add('8 H + DSRD + 32 F', ['H'] * 8 + ['DSRD'] + ['F'] * 32)
add('32 H + 32 F', ['H'] * 32 + ['F'] * 32)
add('32 H + BAR + 32 F', ['H'] * 32 + ['BAR'] + ['F'] * 32)
body = ['        %sif constexpr (PAT == %d) asm volatile(".rept %d\\n" %s ".endr\\n" OPS);' % ('else ' if i else '', i, p[3], ' '.join(p[1])) for i, p in enumerate(pats)]
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[2] * p[3]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('{{256, 256}, {256, 512}, {256, 768}, {256, 1024}, {512, 768}, {512, 1024}}', '{{256, 128}, {256, 512}, {512, 768}, {512, 1024}}')
open(os.path.join(here, 'ubench7.hip'), 'w').write(pre + src[k0:k1] + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body) + '\n    }\n' + rest)
