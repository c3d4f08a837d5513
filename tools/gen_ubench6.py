#!/usr/bin/env python3
"""Generates tools/ubench6.hip from tools/ubench4.hip: .rept-unrolled bodies (no loop effects at 1 wave per SIMD) of
'one X + eight full-rate ops' for many X - what exactly takes a stream of full-rate VALU ops off the 2.3-cycle rate?"""
import os
here = os.path.dirname(os.path.abspath(__file__))
hdr = open(os.path.join(here, 'ubench4.hip')).read()
start = hdr.index('template <int PAT>')
pre = hdr[:start]
X = {'H': 'H', 'xor_e64': 'X64', 'v_cmp': 'C', 's_and': 'S', 'and_lit': 'L', 'bitop3': 'B3', 'v_mov': 'M', 'snop': 'NOP', 'cndmask': 'CNDV',
     'dpp': 'DPPF', 'sdwa': 'SDWAF', 'mad64': 'MADU64', 'lshr': 'R', 'add': 'G', 'pkmin': 'PKMIN', 'perm': 'PERM', 'dsadd': 'DSADD'}
pre = pre.replace('#define OPS', '#define NOP(i) "s_nop 0\\n"\n#define PKMIN(i) "v_pk_min_u16 %[c" #i "], %[c" #i "], %[d" #i "]\\n"\n'
                  '#define PERM(i) "v_perm_b32 %[c" #i "], %[c" #i "], %[d" #i "], %[d" #i "]\\n"\n#define DSADD(i) "ds_add_u32 %[d" #i "], %[c" #i "]\\n"\n#define OPS')
pats = []
def add(name, spec, rept=32):
    cnt = {}; s = []
    for tok in spec:
        if tok in ('F0', 'H0'): s.append('%s(0)' % tok[0]); continue
        i = cnt.get(tok, 0); cnt[tok] = i + 1
        s.append('%s(%d)' % (tok, i % 8))
    nvalu = len([t for t in spec if t not in ('S', 'NOP', 'DSADD')])
    pats.append((name, s, nvalu, rept))
add('pure F (xor VOP2)', ['F'] * 16)
add('pure xor_e64', ['X64'] * 16)
add('pure H alignbit', ['H'] * 16)
for nm, mac in X.items(): add('1 %s + 8 F' % nm, [mac] + ['F'] * 8)
add('1 H + 16 F', ['H'] + ['F'] * 16)
add('1 H + 32 F', ['H'] + ['F'] * 32, rept=16)
add('F dependent chain (same reg)', ['F0'] * 16)
add('H dependent chain (same reg)', ['H0'] * 16)
add('F G alternating (xor a^=b, add b+=a: dependent)', ['F', 'G'] * 8)
add('8 F + 8 xor_e64', ['F'] * 8 + ['X64'] * 8)
add('cmp cndmask mad xor x4', ['C', 'CNDV', 'MADU64', 'F'] * 4)
add('cmp cndmask mad xor s_and s_and x4', ['C', 'CNDV', 'MADU64', 'F', 'S', 'S'] * 4)
body = ['        %sif constexpr (PAT == %d) asm volatile(".rept %d\\n" %s ".endr\\n" OPS);' % ('else ' if i else '', i, p[3], ' '.join(p[1])) for i, p in enumerate(pats)]
k = '''template <int PAT>
__global__ __launch_bounds__(1024) void ub(uint64_t *out, uint32_t seed, int iters)
{
    __shared__ uint32_t lds[4096];
    uint32_t a[8], b[8], c[8], d[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * (2 * i + 1); b[i] = a[i] ^ (0x1234u + i); c[i] = a[i] * 3u; d[i] = (b[i] * 5u) & 0x3FFCu; q[i] = ((uint64_t)a[i] << 32) | b[i]; }
    lds[threadIdx.x] = 0; __syncthreads();
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
%s
    }
''' % '\n'.join(body)
rest = hdr[hdr.index('    const uint64_t c1 = clock64(), w1 = wall_clock64();'):]
rest = rest.replace('for (int i = 0; i < 8; i++) acc +=', 'acc += lds[threadIdx.x & 4095];\n    for (int i = 0; i < 8; i++) acc +=')
r0 = rest.index('    Pat pats[] = {'); r1 = rest.index('    struct Geo')
table = '    Pat pats[] = {\n' + ''.join('        {%d, "%s", ub<%d>, %d},\n' % (i, p[0], i, p[2] * p[3]) for i, p in enumerate(pats)) + '    };\n'
rest = rest[:r0] + table + rest[r1:]
rest = rest.replace('const bool full = p.id <= 8 || p.id >= 30;', 'const bool full = true;')
open(os.path.join(here, 'ubench6.hip'), 'w').write(pre + k + rest)
