// ubench4.hip - how do full-rate and half-rate VALU instructions share a SIMD on gfx950?
// (round 4: the scan2 kernel runs 175 VALU instructions per tile at 4.04 cycles each although 55 of them are "full-rate"
//  ones that a pure stream issues every 2.45 cycles: where do the 1.3 cycles go?)
// Every kernel runs ITERS trips of a 16-instruction pattern over 16 independent register chains (8 for the full-rate op,
// 8 for the half-rate op), with 1 .. 8 waves per SIMD.  Per wave: s_memtime at both ends, the SIMD it ran on (HW_ID, XCC_ID);
// the host groups waves by SIMD and reports cycles per wave-instruction per SIMD = span of the SIMD's waves * clock / instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define F(i) "v_xor_b32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define G(i) "v_add_u32 %[b" #i "], %[b" #i "], %[a" #i "]\n"
#define H(i) "v_alignbit_b32 %[c" #i "], %[c" #i "], %[d" #i "], 7\n"
#define C(i) "v_cmp_lt_u32 vcc, %[c" #i "], %[d" #i "]\n"
#define S(i) "s_and_b64 s[20:21], s[22:23], s[24:25]\n"
#define L(i) "v_and_b32 %[a" #i "], 0xfffc, %[a" #i "]\n"
#define R(i) "v_lshrrev_b32 %[b" #i "], 1, %[b" #i "]\n"
#define B3(i) "v_bitop3_b32 %[a" #i "], %[a" #i "], %[b" #i "], %[c" #i "] bitop3:0x96\n"
#define X64(i) "v_xor_b32_e64 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define M(i) "v_mov_b32 %[a" #i "], %[b" #i "]\n"
#define DPPF(i) "v_xor_b32_dpp %[a" #i "], %[a" #i "], %[b" #i "] row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define SDWAF(i) "v_and_b32_sdwa %[a" #i "], %[a" #i "], %[b" #i "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define ASHR(i) "v_ashrrev_i32 %[a" #i "], 3, %[b" #i "]\n"
#define SUBF(i) "v_sub_u32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define OR2(i) "v_or_b32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define NOT1(i) "v_not_b32 %[a" #i "], %[b" #i "]\n"
#define ADDC(i) "v_addc_co_u32 %[a" #i "], vcc, %[a" #i "], %[b" #i "], vcc\n"
#define PKMOV(i) "v_pk_mov_b32 %[q" #i "], %[q" #i "], %[q" #i "]\n"
#define LSHLADD(i) "v_lshl_add_u32 %[a" #i "], %[a" #i "], 2, %[b" #i "]\n"
#define ADDLSHL(i) "v_add_lshl_u32 %[a" #i "], %[a" #i "], %[b" #i "], 2\n"
#define MAX3(i) "v_max3_u32 %[a" #i "], %[a" #i "], %[b" #i "], %[c" #i "]\n"
#define MED3(i) "v_med3_u32 %[a" #i "], %[a" #i "], %[b" #i "], %[c" #i "]\n"
#define BFI(i) "v_bfi_b32 %[a" #i "], %[a" #i "], %[b" #i "], %[c" #i "]\n"
#define SUBREV(i) "v_subrev_u32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define XNOR(i) "v_xnor_b32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define MBCNT(i) "v_mbcnt_lo_u32_b32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define CNDV(i) "v_cndmask_b32 %[a" #i "], %[a" #i "], %[b" #i "], vcc\n"
#define MINU16(i) "v_min_u16 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define ADDU16(i) "v_add_u16 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define PKADDF32(i) "v_pk_add_f32 %[q" #i "], %[q" #i "], %[q" #i "]\n"
#define ADDF32(i) "v_add_f32 %[a" #i "], %[a" #i "], %[b" #i "]\n"
#define CVT(i) "v_cvt_f32_ubyte0 %[a" #i "], %[b" #i "]\n"
#define MADU64(i) "v_mad_u64_u32 %[q" #i "], vcc, %[a" #i "], 1, %[q" #i "]\n"
#define LSHLADD64(i) "v_lshl_add_u64 %[q" #i "], %[q" #i "], 0, %[q" #i "]\n"

#define NOP(i) "s_nop 0\n"
#define PKMIN(i) "v_pk_min_u16 %[c" #i "], %[c" #i "], %[d" #i "]\n"
#define PERM(i) "v_perm_b32 %[c" #i "], %[c" #i "], %[d" #i "], %[d" #i "]\n"
#define DSADD(i) "ds_add_u32 %[d" #i "], %[c" #i "]\n"
#define OPS                                                                                                                          \
    : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), \
      [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [b4] "+v"(b[4]), [b5] "+v"(b[5]), [b6] "+v"(b[6]), [b7] "+v"(b[7]), \
      [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), \
      [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]), [d6] "+v"(d[6]), [d7] "+v"(d[7]), \
      [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [q5] "+v"(q[5]), [q6] "+v"(q[6]), [q7] "+v"(q[7])  \
    : : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25"

#define ALL8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int PAT>
__global__ __launch_bounds__(1024) void ub(uint64_t *out, uint32_t seed, int iters)
{
    __shared__ uint32_t lds[4096];
    uint32_t a[8], b[8], c[8], d[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * (2 * i + 1); b[i] = a[i] ^ (0x1234u + i); c[i] = a[i] * 3u; d[i] = (b[i] * 5u) & 0x3FFCu; q[i] = ((uint64_t)a[i] << 32) | b[i]; }
    lds[threadIdx.x] = 0; __syncthreads();
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (PAT == 0) asm volatile(".rept 32\n" F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 1) asm volatile(".rept 32\n" X64(0) X64(1) X64(2) X64(3) X64(4) X64(5) X64(6) X64(7) X64(0) X64(1) X64(2) X64(3) X64(4) X64(5) X64(6) X64(7) ".endr\n" OPS);
        else if constexpr (PAT == 2) asm volatile(".rept 32\n" H(0) H(1) H(2) H(3) H(4) H(5) H(6) H(7) H(0) H(1) H(2) H(3) H(4) H(5) H(6) H(7) ".endr\n" OPS);
        else if constexpr (PAT == 3) asm volatile(".rept 32\n" H(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 4) asm volatile(".rept 32\n" X64(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 5) asm volatile(".rept 32\n" C(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 6) asm volatile(".rept 32\n" S(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 7) asm volatile(".rept 32\n" L(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 8) asm volatile(".rept 32\n" B3(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 9) asm volatile(".rept 32\n" M(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 10) asm volatile(".rept 32\n" NOP(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 11) asm volatile(".rept 32\n" CNDV(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 12) asm volatile(".rept 32\n" DPPF(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 13) asm volatile(".rept 32\n" SDWAF(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 14) asm volatile(".rept 32\n" MADU64(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 15) asm volatile(".rept 32\n" R(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 16) asm volatile(".rept 32\n" G(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 17) asm volatile(".rept 32\n" PKMIN(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 18) asm volatile(".rept 32\n" PERM(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 19) asm volatile(".rept 32\n" DSADD(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 20) asm volatile(".rept 32\n" H(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 21) asm volatile(".rept 16\n" H(0) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) ".endr\n" OPS);
        else if constexpr (PAT == 22) asm volatile(".rept 32\n" F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) ".endr\n" OPS);
        else if constexpr (PAT == 23) asm volatile(".rept 32\n" H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) H(0) ".endr\n" OPS);
        else if constexpr (PAT == 24) asm volatile(".rept 32\n" F(0) G(0) F(1) G(1) F(2) G(2) F(3) G(3) F(4) G(4) F(5) G(5) F(6) G(6) F(7) G(7) ".endr\n" OPS);
        else if constexpr (PAT == 25) asm volatile(".rept 32\n" F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) X64(0) X64(1) X64(2) X64(3) X64(4) X64(5) X64(6) X64(7) ".endr\n" OPS);
        else if constexpr (PAT == 26) asm volatile(".rept 32\n" C(0) CNDV(0) MADU64(0) F(0) C(1) CNDV(1) MADU64(1) F(1) C(2) CNDV(2) MADU64(2) F(2) C(3) CNDV(3) MADU64(3) F(3) ".endr\n" OPS);
        else if constexpr (PAT == 27) asm volatile(".rept 32\n" C(0) CNDV(0) MADU64(0) F(0) S(0) S(1) C(1) CNDV(1) MADU64(1) F(1) S(2) S(3) C(2) CNDV(2) MADU64(2) F(2) S(4) S(5) C(3) CNDV(3) MADU64(3) F(3) S(6) S(7) ".endr\n" OPS);
    }
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    uint32_t acc = 0;
    acc += lds[threadIdx.x & 4095];
    for (int i = 0; i < 8; i++) acc += a[i] + b[i] + c[i] + d[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    if ((threadIdx.x & 63) == 0) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[w * 4 + 0] = w0; out[w * 4 + 1] = w1; out[w * 4 + 2] = c1 - c0; out[w * 4 + 3] = ((uint64_t)(xcc & 0xF) << 32) | hwid | ((uint64_t)(acc == 0x12345u) << 63);
    }
}

struct Pat { int id; const char *name; void (*fn)(uint64_t *, uint32_t, int); int nvalu; };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2048;
    uint64_t *d; CHK(hipMalloc(&d, 512 * 16 * 32 + 4096));
    Pat pats[] = {
        {0, "pure F (xor VOP2)", ub<0>, 512},
        {1, "pure xor_e64", ub<1>, 512},
        {2, "pure H alignbit", ub<2>, 512},
        {3, "1 H + 8 F", ub<3>, 288},
        {4, "1 xor_e64 + 8 F", ub<4>, 288},
        {5, "1 v_cmp + 8 F", ub<5>, 288},
        {6, "1 s_and + 8 F", ub<6>, 256},
        {7, "1 and_lit + 8 F", ub<7>, 288},
        {8, "1 bitop3 + 8 F", ub<8>, 288},
        {9, "1 v_mov + 8 F", ub<9>, 288},
        {10, "1 snop + 8 F", ub<10>, 256},
        {11, "1 cndmask + 8 F", ub<11>, 288},
        {12, "1 dpp + 8 F", ub<12>, 288},
        {13, "1 sdwa + 8 F", ub<13>, 288},
        {14, "1 mad64 + 8 F", ub<14>, 288},
        {15, "1 lshr + 8 F", ub<15>, 288},
        {16, "1 add + 8 F", ub<16>, 288},
        {17, "1 pkmin + 8 F", ub<17>, 288},
        {18, "1 perm + 8 F", ub<18>, 288},
        {19, "1 dsadd + 8 F", ub<19>, 256},
        {20, "1 H + 16 F", ub<20>, 544},
        {21, "1 H + 32 F", ub<21>, 528},
        {22, "F dependent chain (same reg)", ub<22>, 512},
        {23, "H dependent chain (same reg)", ub<23>, 512},
        {24, "F G alternating (xor a^=b, add b+=a: dependent)", ub<24>, 512},
        {25, "8 F + 8 xor_e64", ub<25>, 512},
        {26, "cmp cndmask mad xor x4", ub<26>, 512},
        {27, "cmp cndmask mad xor s_and s_and x4", ub<27>, 512},
    };
    struct Geo { int blocks, threads; } geos[] = {{256, 256}, {256, 512}, {256, 768}, {256, 1024}, {512, 768}, {512, 1024}};
    printf("cycles per wave-instruction per SIMD (median over SIMDs; [waves per SIMD seen: min..max]); columns = launch geometry\n");
    printf("%-44s", "pattern");
    for (auto &g : geos) printf("  %4dx%-4d      ", g.blocks, g.threads);
    printf("\n");
    for (auto &p : pats) {
        printf("%-44s", p.name);
        const bool full = true;
        for (auto &g : geos) {
            if (!full && !(g.blocks == 512 && g.threads == 1024) && !(g.blocks == 256 && g.threads == 256)) { printf("  %-14s", "-"); continue; }
            const int waves = g.blocks * (g.threads / 64);
            for (int rep = 0; rep < 2; rep++) {   // first run warms the clocks
                hipLaunchKernelGGL(p.fn, dim3(g.blocks), dim3(g.threads), 0, 0, d, 1u + rep, iters);
                CHK(hipDeviceSynchronize());
            }
            std::vector<uint64_t> h((size_t)waves * 4);
            CHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            struct Simd { uint64_t w0 = ~0ull, w1 = 0; double cyc = 0, wall = 0; int n = 0; };
            std::map<uint64_t, Simd> simds;
            for (int w = 0; w < waves; w++) {
                const uint64_t id = h[w * 4 + 3] & 0x7FFFFFFFFFFFFFFFull;
                const uint64_t key = ((id >> 32) << 16) | ((uint32_t)id & 0xFF30u);   // xcc | se, sh, cu, simd
                Simd &s = simds[key];
                s.w0 = std::min(s.w0, h[w * 4 + 0]); s.w1 = std::max(s.w1, h[w * 4 + 1]);
                s.cyc += (double)h[w * 4 + 2]; s.wall += (double)(h[w * 4 + 1] - h[w * 4 + 0]); s.n++;
            }
            std::vector<double> cpi; int nmin = 1 << 30, nmax = 0;
            for (auto &kv : simds) {
                const Simd &s = kv.second;
                const double ghz = s.cyc / (s.wall * 10.0);                // shader cycles per 10 ns tick of the 100 MHz wall clock
                const double span_cycles = (double)(s.w1 - s.w0) * 10.0 * ghz;
                cpi.push_back(span_cycles / ((double)s.n * iters * p.nvalu));
                nmin = std::min(nmin, s.n); nmax = std::max(nmax, s.n);
            }
            std::sort(cpi.begin(), cpi.end());
            printf("  %5.2f [%d..%d]%s", cpi[cpi.size() / 2], nmin, nmax, nmax > 9 ? "" : " ");
        }
        printf("\n");
    }
    return 0;
}
