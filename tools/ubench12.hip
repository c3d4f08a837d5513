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
    : : "vcc", "scc", "s20", "s21", "s26", "s27", "s28", "s29", "v10", "v11", "v20", "v21", "v30", "v31", "v40", "v41"

#define PE(i) "s_and_b64 exec, s[22:23], s[24:25]\n"
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
#define OPS2                                                                                                                           \
    : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), \
      [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [b4] "+v"(b[4]), [b5] "+v"(b[5]), [b6] "+v"(b[6]), [b7] "+v"(b[7]), \
      [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), \
      [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]), [d6] "+v"(d[6]), [d7] "+v"(d[7]), \
      [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [q5] "+v"(q[5]), [q6] "+v"(q[6]), [q7] "+v"(q[7])  \
    : [one] "v"(one) : "vcc", "memory", "scc", "s20", "s21", "s26", "s27", "s28", "s29", "v10", "v11", "v20", "v21", "v30", "v31", "v40", "v41"

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
    asm volatile("s_mov_b64 s[22:23], -1\n s_mov_b64 s[24:25], -1\n s_mov_b32 s29, 0" ::: "s22", "s23", "s24", "s25", "s29");
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (PAT == 0) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PS(1) PM(1) PX(1) PD(1) PB(1) PA(1) PE(2) PC(2) PS(2) PM(2) PX(2) PD(2) PB(2) PA(2) PE(3) PC(3) PS(3) PM(3) PX(3) PD(3) PB(3) PA(3) PE(4) PC(4) PS(4) PM(4) PX(4) PD(4) PB(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 1) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PS(1) PM(1) PX(1) PB(1) PA(1) PE(2) PC(2) PS(2) PM(2) PX(2) PB(2) PA(2) PE(3) PC(3) PS(3) PM(3) PX(3) PB(3) PA(3) PE(4) PC(4) PS(4) PM(4) PX(4) PB(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 2) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PS(1) PX(1) PM(1) PA(1) PE(2) PC(2) PB(2) PS(2) PX(2) PM(2) PA(2) PE(3) PC(3) PB(3) PS(3) PX(3) PM(3) PA(3) PE(4) PC(4) PB(4) PS(4) PX(4) PM(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 3) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PC(1) PS(1) PM(1) PX(1) PB(1) PA(1) PC(2) PS(2) PM(2) PX(2) PB(2) PA(2) PC(3) PS(3) PM(3) PX(3) PB(3) PA(3) PC(4) PS(4) PM(4) PX(4) PB(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 4) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PS(1) PM(1) PX(1) PE(2) PC(2) PS(2) PM(2) PX(2) PE(3) PC(3) PS(3) PM(3) PX(3) PE(4) PC(4) PS(4) PM(4) PX(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 5) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PS(1) PMV(1) PX(1) PA(1) PE(2) PC(2) PB(2) PS(2) PMV(2) PX(2) PA(2) PE(3) PC(3) PB(3) PS(3) PMV(3) PX(3) PA(3) PE(4) PC(4) PB(4) PS(4) PMV(4) PX(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 6) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PS2(1) PL(1) PX2(1) PA(1) PE(2) PC(2) PB(2) PS2(2) PL(2) PX2(2) PA(2) PE(3) PC(3) PB(3) PS2(3) PL(3) PX2(3) PA(3) PE(4) PC(4) PB(4) PS2(4) PL(4) PX2(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 7) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PS(1) PN(1) PU(1) PX(1) PA(1) PE(2) PC(2) PB(2) PS(2) PN(2) PU(2) PX(2) PA(2) PE(3) PC(3) PB(3) PS(3) PN(3) PU(3) PX(3) PA(3) PE(4) PC(4) PB(4) PS(4) PN(4) PU(4) PX(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 8) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PS(1) PUF(1) PX(1) PA(1) PE(2) PC(2) PB(2) PS(2) PUF(2) PX(2) PA(2) PE(3) PC(3) PB(3) PS(3) PUF(3) PX(3) PA(3) PE(4) PC(4) PB(4) PS(4) PUF(4) PX(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 9) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PS(1) PX(1) PA(1) PE(2) PC(2) PB(2) PS(2) PX(2) PA(2) PE(3) PC(3) PB(3) PS(3) PX(3) PA(3) PE(4) PC(4) PB(4) PS(4) PX(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 10) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PC(1) PB(1) PA(1) PE(2) PC(2) PB(2) PA(2) PE(3) PC(3) PB(3) PA(3) PE(4) PC(4) PB(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 11) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PCE(1) PBE(1) PSE(1) PM(1) PX(1) PA(1) PE(2) PCE(2) PBE(2) PSE(2) PM(2) PX(2) PA(2) PE(3) PCE(3) PBE(3) PSE(3) PM(3) PX(3) PA(3) PE(4) PCE(4) PBE(4) PSE(4) PM(4) PX(4) PA(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 12) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PSUB(1) PASH(1) PNF(1) PBO(1) PN(1) PU(1) PX(1) PE(2) PSUB(2) PASH(2) PNF(2) PBO(2) PN(2) PU(2) PX(2) PE(3) PSUB(3) PASH(3) PNF(3) PBO(3) PN(3) PU(3) PX(3) PE(4) PSUB(4) PASH(4) PNF(4) PBO(4) PN(4) PU(4) PX(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 13) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PMIN(1) PM(1) PX(1) PE(2) PMIN(2) PM(2) PX(2) PE(3) PMIN(3) PM(3) PX(3) PE(4) PMIN(4) PM(4) PX(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
        else if constexpr (PAT == 14) asm volatile("v_mov_b32 v11, 0\n v_mov_b32 v21, 0\n v_mov_b32 v31, 0\n v_mov_b32 v41, 0\n .rept 16\n" PE(1) PMIN(1) PN(1) PU(1) PX(1) PE(2) PMIN(2) PN(2) PU(2) PX(2) PE(3) PMIN(3) PN(3) PU(3) PX(3) PE(4) PMIN(4) PN(4) PU(4) PX(4) ".endr\n s_mov_b64 exec, -1\n" OPS2);
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
        {0, "V0 shipped: E cmp cnd mad xor ds bcnt add", ub<0>, 64},
        {1, "V0 without the LDS atomic", ub<1>, 64},
        {2, "order B: E cmp bcnt cnd xor mad add (no LDS)", ub<2>, 64},
        {3, "no exec write: cmp cnd mad xor bcnt add", ub<3>, 64},
        {4, "no count: E cmp cnd mad xor", ub<4>, 64},
        {5, "mad with vcc as its carry-out: E cmp bcnt cnd mad(vcc) xor add", ub<5>, 64},
        {6, "lshl_add_u64 on a (t:0) pair: E cmp bcnt cnd lshladd xor add", ub<6>, 64},
        {7, "32-bit sum of 28-bit words: E cmp bcnt cnd and add xor add", ub<7>, 64},
        {8, "32-bit wrapping sum: E cmp bcnt cnd add xor add", ub<8>, 64},
        {9, "no sum at all: E cmp bcnt cnd xor add", ub<9>, 64},
        {10, "no select/digests: E cmp bcnt add", ub<10>, 64},
        {11, "cmp into s[20:21] (VOP3), cnd_e64: E cmpE bcnt cndE mad xor add", ub<11>, 64},
        {12, "arithmetic select (31-bit T): E sub ashr bitop3 nf-=m and add xor", ub<12>, 64},
        {13, "min only: E min mad xor", ub<13>, 64},
        {14, "min + full-rate sum: E min and add xor", ub<14>, 64},
    };
    struct Geo { int blocks, threads; } geos[] = {{256, 512}, {512, 512}, {512, 768}};
    printf("cycles per wave-instruction per SIMD (median over SIMDs; [waves per SIMD seen: min..max]); columns = launch geometry\n");
    printf("%-80s", "formulation (cycles per POSITION per SIMD)");
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
