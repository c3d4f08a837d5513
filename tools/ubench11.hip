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
#define S0(i) "s_nop 0\n"
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
#define OPS                                                                                                                          \
    : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), \
      [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [b4] "+v"(b[4]), [b5] "+v"(b[5]), [b6] "+v"(b[6]), [b7] "+v"(b[7]), \
      [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), \
      [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]), [d6] "+v"(d[6]), [d7] "+v"(d[7]), \
      [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [q5] "+v"(q[5]), [q6] "+v"(q[6]), [q7] "+v"(q[7])  \
    : : "vcc", "scc", "s20", "s21", "s26", "s27", "s28", "s29", "s30"

#define HCS(i) "v_cmp_eq_u32_sdwa s[20:21], %[c" #i "], %[d" #i "] src0_sel:BYTE_1 src1_sel:BYTE_1\n"
#define XSd(i) "v_cndmask_b32 %[a" #i "], %[b" #i "], %[c" #i "], vcc\n"       /* select straight after its compare */
#define XXd(i) "v_xor_b32 %[b7], %[b7], %[a" #i "]\n"                          /* xor of the selected word */
#define XMd(i) "v_mad_u64_u32 %[q" #i "], s[26:27], %[a" #i "], 1, %[q" #i "]\n"
#define ABd(i) "v_alignbit_b32 %[c" #i "], %[d" #i "], %[b" #i "], 6\n"
#define FS(i) "v_and_b32 %[a" #i "], s30, %[b" #i "]\n"                        /* full-rate op with an SGPR operand */
#define OPS2                                                                                                                           \
    : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), \
      [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [b4] "+v"(b[4]), [b5] "+v"(b[5]), [b6] "+v"(b[6]), [b7] "+v"(b[7]), \
      [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), \
      [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]), [d6] "+v"(d[6]), [d7] "+v"(d[7]), \
      [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [q5] "+v"(q[5]), [q6] "+v"(q[6]), [q7] "+v"(q[7])  \
    : [one] "v"(one) : "vcc", "memory", "scc", "s20", "s21", "s26", "s27", "s28", "s29", "s30"

#define ALL8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int PAT>
__global__ __launch_bounds__(1024) void ub(uint64_t *out, uint32_t seed, int iters)
{
    __shared__ uint32_t lds[16384];
    uint32_t a[8], b[8], c[8], d[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * (2 * i + 1); b[i] = a[i] ^ (0x1234u + i); c[i] = a[i] * 3u; d[i] = (b[i] * 5u) & 0x3FFCu; q[i] = ((uint64_t)a[i] << 32) | b[i]; }
    lds[threadIdx.x] = 0; __syncthreads();
    const uint32_t one = 1;
    asm volatile("s_mov_b64 s[22:23], -1\n s_mov_b64 s[24:25], -1\n s_mov_b32 s29, 0\n s_mov_b32 s30, 0xfffc" ::: "s22", "s23", "s24", "s25", "s29", "s30");
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (PAT == 0) asm volatile(".rept 8\n" H(0) S(0) F(0) F(1) H(1) S(1) F(2) F(3) H(2) S(2) F(4) F(5) H(3) S(3) F(6) F(0) H(4) S(4) F(1) F(2) H(5) S(5) F(3) F(4) H(6) S(6) F(5) F(6) H(0) S(0) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 1) asm volatile(".rept 8\n" H(0) XE(0) F(0) F(1) H(1) XE(1) F(2) F(3) H(2) XE(2) F(4) F(5) H(3) XE(3) F(6) F(0) H(4) XE(4) F(1) F(2) H(5) XE(5) F(3) F(4) H(6) XE(6) F(5) F(6) H(0) XE(0) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 2) asm volatile(".rept 8\n" XE(0) H(0) S(0) F(0) F(1) XE(1) H(1) S(1) F(2) F(3) XE(2) H(2) S(2) F(4) F(5) XE(3) H(3) S(3) F(6) F(0) XE(4) H(4) S(4) F(1) F(2) XE(5) H(5) S(5) F(3) F(4) XE(6) H(6) S(6) F(5) F(6) XE(0) H(0) S(0) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 3) asm volatile(".rept 8\n" H(0) S(0) F(0) F(1) XD(0) H(1) S(1) F(2) F(3) XD(1) H(2) S(2) F(4) F(5) XD(2) H(3) S(3) F(6) F(0) XD(3) H(4) S(4) F(1) F(2) XD(4) H(5) S(5) F(3) F(4) XD(5) H(6) S(6) F(5) F(6) XD(6) H(0) S(0) F(0) F(1) XD(0) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 4) asm volatile(".rept 8\n" H(0) S(0) F(0) F(1) H(1) S(1) F(2) F(3) H(2) S(2) F(4) F(5) XD(0) H(3) S(3) F(6) F(0) H(4) S(4) F(1) F(2) H(5) S(5) F(3) F(4) XD(1) H(6) S(6) F(5) F(6) H(0) S(0) F(0) F(1) H(1) S(1) F(2) F(3) XD(2) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 5) asm volatile(".rept 8\n" XC(0) S(0) XSd(0) F(0) XC(1) S(1) XSd(1) F(1) XC(2) S(2) XSd(2) F(2) XC(3) S(3) XSd(3) F(3) XC(4) S(4) XSd(4) F(4) XC(5) S(5) XSd(5) F(5) XC(6) S(6) XSd(6) F(6) XC(0) S(0) XSd(0) F(0) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 6) asm volatile(".rept 8\n" XC(0) S(0) F(0) XSd(0) XC(1) S(1) F(1) XSd(1) XC(2) S(2) F(2) XSd(2) XC(3) S(3) F(3) XSd(3) XC(4) S(4) F(4) XSd(4) XC(5) S(5) F(5) XSd(5) XC(6) S(6) F(6) XSd(6) XC(0) S(0) F(0) XSd(0) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 7) asm volatile(".rept 8\n" H(0) S(0) FD(0) FD(1) H(1) S(1) FD(2) FD(3) H(2) S(2) FD(4) FD(5) H(3) S(3) FD(6) FD(0) H(4) S(4) FD(1) FD(2) H(5) S(5) FD(3) FD(4) H(6) S(6) FD(5) FD(6) H(0) S(0) FD(0) FD(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 8) asm volatile(".rept 8\n" XM(0) S(0) F(0) F(1) XM(1) S(1) F(2) F(3) XM(2) S(2) F(4) F(5) XM(3) S(3) F(6) F(0) XM(4) S(4) F(1) F(2) XM(5) S(5) F(3) F(4) XM(6) S(6) F(5) F(6) XM(0) S(0) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 9) asm volatile(".rept 8\n" HCS(0) S(0) F(0) F(1) HCS(1) S(1) F(2) F(3) HCS(2) S(2) F(4) F(5) HCS(3) S(3) F(6) F(0) HCS(4) S(4) F(1) F(2) HCS(5) S(5) F(3) F(4) HCS(6) S(6) F(5) F(6) HCS(0) S(0) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 10) asm volatile(".rept 8\n" H(0) S(0) FS(0) F(0) H(1) S(1) FS(1) F(1) H(2) S(2) FS(2) F(2) H(3) S(3) FS(3) F(3) H(4) S(4) FS(4) F(4) H(5) S(5) FS(5) F(5) H(6) S(6) FS(6) F(6) H(0) S(0) FS(0) F(0) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 11) asm volatile(".rept 8\n" DP(0) S(0) F(0) F(1) DP(1) S(1) F(2) F(3) DP(2) S(2) F(4) F(5) DP(3) S(3) F(6) F(0) DP(4) S(4) F(1) F(2) DP(5) S(5) F(3) F(4) DP(6) S(6) F(5) F(6) DP(0) S(0) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 12) asm volatile(".rept 8\n" H(0) XB(0) F(0) F(1) H(1) XA(0) F(2) F(3) H(2) XB(1) F(4) F(5) H(3) XA(1) F(6) F(0) H(4) XB(2) F(1) F(2) H(5) XA(2) F(3) F(4) H(6) XB(3) F(5) F(6) H(0) XA(3) F(0) F(1) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 13) asm volatile(".rept 4\n" XE(0) XC(0) XB(0) XSd(0) XXd(0) XMd(0) XA(0) XE(1) XC(1) XB(1) XSd(1) XXd(1) XMd(1) XA(1) XE(2) XC(2) XB(2) XSd(2) XXd(2) XMd(2) XA(2) XE(3) XC(3) XB(3) XSd(3) XXd(3) XMd(3) XA(3) XE(4) XC(4) XB(4) XSd(4) XXd(4) XMd(4) XA(4) XE(5) XC(5) XB(5) XSd(5) XXd(5) XMd(5) XA(5) XE(6) XC(6) XB(6) XSd(6) XXd(6) XMd(6) XA(6) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 14) asm volatile(".rept 4\n" XE(0) XC(0) XSd(0) XMd(0) XXd(0) XB(0) XA(0) XE(1) XC(1) XSd(1) XMd(1) XXd(1) XB(1) XA(1) XE(2) XC(2) XSd(2) XMd(2) XXd(2) XB(2) XA(2) XE(3) XC(3) XSd(3) XMd(3) XXd(3) XB(3) XA(3) XE(4) XC(4) XSd(4) XMd(4) XXd(4) XB(4) XA(4) XE(5) XC(5) XSd(5) XMd(5) XXd(5) XB(5) XA(5) XE(6) XC(6) XSd(6) XMd(6) XXd(6) XB(6) XA(6) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 15) asm volatile(".rept 4\n" XE(0) XC(0) XB(0) XSd(0) XXd(0) XMd(0) XD(0) XA(0) XE(1) XC(1) XB(1) XSd(1) XXd(1) XMd(1) XD(1) XA(1) XE(2) XC(2) XB(2) XSd(2) XXd(2) XMd(2) XD(2) XA(2) XE(3) XC(3) XB(3) XSd(3) XXd(3) XMd(3) XD(3) XA(3) XE(4) XC(4) XB(4) XSd(4) XXd(4) XMd(4) XD(4) XA(4) XE(5) XC(5) XB(5) XSd(5) XXd(5) XMd(5) XD(5) XA(5) XE(6) XC(6) XB(6) XSd(6) XXd(6) XMd(6) XD(6) XA(6) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 16) asm volatile(".rept 8\n" AB(0) S(0) AB(1) S(1) AB(2) S(2) AB(3) S(3) AB(4) S(4) AB(5) S(5) AB(6) S(6) AB(0) S(0) DP(0) S(1) DP(1) S(2) PM(0) S(3) PM(1) S(4) SD(0) S(5) SD(1) S(6) AL(0) AL(1) XE(0) XC(0) XB(0) XSd(0) XXd(0) XMd(0) XD(0) XA(0) XE(1) XC(1) XB(1) XSd(1) XXd(1) XMd(1) XD(1) XA(1) XE(2) XC(2) XB(2) XSd(2) XXd(2) XMd(2) XD(2) XA(2) XE(3) XC(3) XB(3) XSd(3) XXd(3) XMd(3) XD(3) XA(3) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 17) asm volatile(".rept 8\n" AB(0) AB(1) AB(2) AB(3) AB(4) AB(5) AB(6) AB(0) DP(0) DP(1) PM(0) PM(1) SD(0) SD(1) AL(0) AL(1) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(0) S(1) S(2) S(3) S(4) S(5) S(6) XE(0) XC(0) XSd(0) XMd(0) XXd(0) XD(0) XB(0) XA(0) XE(1) XC(1) XSd(1) XMd(1) XXd(1) XD(1) XB(1) XA(1) XE(2) XC(2) XSd(2) XMd(2) XXd(2) XD(2) XB(2) XA(2) XE(3) XC(3) XSd(3) XMd(3) XXd(3) XD(3) XB(3) XA(3) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 18) asm volatile(".rept 8\n" AB(0) S(0) AB(1) S(1) AB(2) S(2) AB(3) S(3) AB(4) S(4) AB(5) S(5) AB(6) S(6) AB(0) S(0) DP(0) S(1) DP(1) S(2) PM(0) S(3) PM(1) S(4) SD(0) S(5) SD(1) S(6) AL(0) AL(1) XE(0) XC(0) XB(0) XSd(0) XXd(0) XMd(0) XA(0) XE(1) XC(1) XB(1) XSd(1) XXd(1) XMd(1) XA(1) XE(2) XC(2) XB(2) XSd(2) XXd(2) XMd(2) XA(2) XE(3) XC(3) XB(3) XSd(3) XXd(3) XMd(3) XA(3) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 19) asm volatile(".rept 4\n" AB(0) S(0) AB(1) S(1) AB(2) S(2) AB(3) S(3) AB(4) S(4) AB(5) S(5) AB(6) S(6) AB(0) S(0) DP(0) S(1) DP(1) S(2) PM(0) S(3) PM(1) S(4) SD(0) S(5) SD(1) S(6) AL(0) AL(1) AB(1) S(0) AB(2) S(1) AB(3) S(2) AB(4) S(3) AB(5) S(4) AB(6) S(5) AB(0) S(6) AB(1) S(0) DP(2) S(1) DP(3) S(2) PM(2) S(3) PM(3) S(4) SD(2) S(5) SD(3) S(6) AL(2) AL(3) AB(2) S(0) AB(3) S(1) AB(4) S(2) AB(5) S(3) AB(6) S(4) AB(0) S(5) AB(1) S(6) AB(2) S(0) DP(4) S(1) DP(5) S(2) PM(4) S(3) PM(5) S(4) SD(4) S(5) SD(5) S(6) AL(4) AL(5) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 20) asm volatile(".rept 4\n" AB(0) AB(1) AB(2) AB(3) AB(4) AB(5) AB(6) AB(0) DP(0) DP(1) PM(0) PM(1) SD(0) SD(1) AL(0) AL(1) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(0) S(1) S(2) S(3) S(4) S(5) S(6) AB(1) AB(2) AB(3) AB(4) AB(5) AB(6) AB(0) AB(1) DP(2) DP(3) PM(2) PM(3) SD(2) SD(3) AL(2) AL(3) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(0) S(1) S(2) S(3) S(4) S(5) S(6) AB(2) AB(3) AB(4) AB(5) AB(6) AB(0) AB(1) AB(2) DP(4) DP(5) PM(4) PM(5) SD(4) SD(5) AL(4) AL(5) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(0) S(1) S(2) S(3) S(4) S(5) S(6) ".endr\n s_mov_b64 exec, -1\n" OPS2);
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
        {0, "[H S F F] x8 (reference: ideal 2.73)", ub<0>, 192},
        {1, "[H E F F]  scalar = exec write", ub<1>, 192},
        {2, "[E H S F F]  + exec write", ub<2>, 192},
        {3, "[H S F F D]  + LDS atomic", ub<3>, 192},
        {4, "[H S F F] x3 + D  (LDS 1 per 9 VALU)", ub<4>, 216},
        {5, "[cmp S cndmask(vcc dep) F]", ub<5>, 192},
        {6, "[cmp S F cndmask(vcc dep)]", ub<6>, 192},
        {7, "[H S FD FD]  F reads the H result", ub<7>, 192},
        {8, "[mad S F F]", ub<8>, 192},
        {9, "[cmp_sdwa->sgpr S F F]", ub<9>, 192},
        {10, "[H S F(sgpr operand) F]", ub<10>, 192},
        {11, "[dpp S F F]", ub<11>, 192},
        {12, "[H S F F] with s_bcnt1/s_add as the scalars", ub<12>, 192},
        {13, "region, order B, no LDS, true deps (ideal 3.08)", ub<13>, 112},
        {14, "region as shipped order, no LDS, true deps", ub<14>, 112},
        {15, "region, order B, with LDS atomic", ub<15>, 112},
        {16, "tile: outside ops with a scalar after each + region B + LDS (ideal 3.46)", ub<16>, 256},
        {17, "tile: outside ops then their scalars clustered + region as shipped + LDS", ub<17>, 256},
        {18, "tile: outside+S, region B, no LDS", ub<18>, 256},
        {19, "outside ops only, scalar after each", ub<19>, 192},
        {20, "outside ops only, scalars clustered", ub<20>, 192},
    };
    struct Geo { int blocks, threads; } geos[] = {{256, 512}, {512, 512}, {512, 768}};
    printf("cycles per wave-instruction per SIMD (median over SIMDs; [waves per SIMD seen: min..max]); columns = launch geometry\n");
    printf("%-80s", "pattern");
    for (auto &g : geos) printf("  %4dx%-4d      ", g.blocks, g.threads);
    printf("\n");
    for (auto &p : pats) {
        printf("%-80s", p.name);
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
