#!/usr/bin/env python3
"""ubench9: [m H, n F] and [m H, S, n F] for small m, n: where exactly do the full-rate ops after a half-rate op stop being cheap?"""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'ubench8.hip')).read()
pre = src[:src.index('template <int PAT>')]
pats = []
def add(name, spec, rept=8):
    cnt = {}; s = []
    for tok in spec:
        i = cnt.get(tok, 0); cnt[tok] = i + 1
        s.append('%s(%d)' % (tok, i % 8))
    pats.append((name, s, len([t for t in spec if t in ('F', 'H')]), rept))
for m in (1, 2):
    for n in (1, 2, 3, 4, 5, 6, 8):
        g = max(1, 24 // (m + n))
        add('[%dH %dF] x%d' % (m, n, g), (['H'] * m + ['F'] * n) * g)
for m in (1, 2, 3):
    for n in (1, 2, 3, 4, 6, 8):
        g = max(1, 24 // (m + n))
        add('[%dH S %dF] x%d' % (m, n, g), (['H'] * m + ['S'] + ['F'] * n) * g)
for n in (2, 4, 8):
    g = max(1, 24 // (1 + n))
    add('[1H S0 %dF] x%d (s_nop)' % (n, g), (['H', 'S0'] + ['F'] * n) * g)
    add('[1H %dF S] x%d' % (n, g), (['H'] + ['F'] * n + ['S']) * g)
    add('[1H F S %dF] x%d' % (n - 1, g), (['H', 'F', 'S'] + ['F'] * (n - 1)) * g)
body = ['        %sif constexpr (PAT == %d) asm volatile(".rept %d\\n" %s ".endr\\n s_mov_b64 exec, -1\\n" OPS2);' % ('else ' if i else '', i, p[3], ' '.join(p[1])) for i, p in enumerate(pats)]
k0 = src.index('template <int PAT>'); k1 = src.index('    for (int it = 0; it < iters; it++) {')
k2 = src.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();')
rest = src[k2:]
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[2] * p[3]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('{{256, 512}, {256, 1024}, {512, 768}, {512, 1024}}', '{{256, 512}, {512, 768}}')
open(os.path.join(here, 'ubench9.hip'), 'w').write(pre + src[k0:k1] + '    for (int it = 0; it < iters; it++) {\n' + '\n'.join(body) + '\n    }\n' + rest)
