// ubench.hip — VALU instruction issue-rate microbenchmark on gfx950 (which 64-bit / cross-lane ops are full rate?).
// Each kernel runs a long dependent-free stream of ONE instruction in 8 independent chains per lane; the reported
// number is wave-instructions per cycle per SIMD (1/2 = full rate for a wave64 on a SIMD32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

#define BENCH_KERNEL(NAME, BODY)                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)           \
    {                                                                                   \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;        \
        uint32_t b0 = a0 ^ 0x1234, b1 = a1 ^ 0x2345, b2 = a2 ^ 0x3456, b3 = a3 ^ 0x4567;\
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;                                    \
        uint32_t cnt = 0;                                                               \
        const uint64_t c0_ = clock64(), w0_ = wall_clock64();                           \
        for (int i = 0; i < ITERS; i++) { BODY }                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) { out[1 << 20] = (uint32_t)(clock64() - c0_); out[(1 << 20) + 1] = (uint32_t)(wall_clock64() - w0_); } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + cnt + (uint32_t)(q0 + q1 + q2 + q3); \
    }

// 8 instructions per iteration in each body
BENCH_KERNEL(k_add32, {
    asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n"
                 "v_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n v_add_u32 %5, %5, %4\n v_add_u32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_alignbit, {
    asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %4, %4, %5, 7\n v_alignbit_b32 %6, %6, %7, 7\n"
                 "v_alignbit_b32 %1, %1, %0, 9\n v_alignbit_b32 %3, %3, %2, 9\n v_alignbit_b32 %5, %5, %4, 9\n v_alignbit_b32 %7, %7, %6, 9\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_perm, {
    asm volatile("v_perm_b32 %0, %0, %1, %1\n v_perm_b32 %2, %2, %3, %3\n v_perm_b32 %4, %4, %5, %5\n v_perm_b32 %6, %6, %7, %7\n"
                 "v_perm_b32 %1, %1, %0, %0\n v_perm_b32 %3, %3, %2, %2\n v_perm_b32 %5, %5, %4, %4\n v_perm_b32 %7, %7, %6, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_bitop3, {
    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8\n v_bitop3_b32 %2, %2, %3, %4 bitop3:0xc8\n v_bitop3_b32 %4, %4, %5, %6 bitop3:0xc8\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0xc8\n"
                 "v_bitop3_b32 %1, %1, %0, %3 bitop3:0xc8\n v_bitop3_b32 %3, %3, %2, %5 bitop3:0xc8\n v_bitop3_b32 %5, %5, %4, %7 bitop3:0xc8\n v_bitop3_b32 %7, %7, %6, %1 bitop3:0xc8\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_dpp_shr, {
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_dpp_rowshr, {
    asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_lshr64, {
    asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3\n"
                 "v_lshrrev_b64 %0, 5, %0\n v_lshrrev_b64 %1, 5, %1\n v_lshrrev_b64 %2, 5, %2\n v_lshrrev_b64 %3, 5, %3\n"
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)); })
BENCH_KERNEL(k_lshladd64, {
    asm volatile("v_lshl_add_u64 %0, %1, 0, %0\n v_lshl_add_u64 %1, %2, 0, %1\n v_lshl_add_u64 %2, %3, 0, %2\n v_lshl_add_u64 %3, %0, 0, %3\n"
                 "v_lshl_add_u64 %0, %2, 0, %0\n v_lshl_add_u64 %1, %3, 0, %1\n v_lshl_add_u64 %2, %0, 0, %2\n v_lshl_add_u64 %3, %1, 0, %3\n"
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)); })
BENCH_KERNEL(k_cmp64, {
    asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0\n"
                 "v_cmp_lt_u64 vcc, %0, %2\n v_cmp_lt_u64 vcc, %1, %3\n v_cmp_lt_u64 vcc, %2, %0\n v_cmp_lt_u64 vcc, %3, %1\n"
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "vcc"); })
BENCH_KERNEL(k_cmp32, {
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0\n"
                 "v_cmp_lt_u32 vcc, %0, %2\n v_cmp_lt_u32 vcc, %1, %3\n v_cmp_lt_u32 vcc, %2, %0\n v_cmp_lt_u32 vcc, %3, %1\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc"); })
BENCH_KERNEL(k_cmp_cndmask, {
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %4, %0, %1, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %5, %2, %3, vcc\n"
                 "v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %6, %1, %2, vcc\n v_cmp_lt_u32 vcc, %3, %0\n v_cndmask_b32 %7, %3, %0, vcc\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : : "vcc"); })
BENCH_KERNEL(k_addco, {
    asm volatile("v_add_co_u32 %0, vcc, %0, %0\n v_add_co_u32 %1, vcc, %1, %1\n v_add_co_u32 %2, vcc, %2, %2\n v_add_co_u32 %3, vcc, %3, %3\n"
                 "v_add_co_u32 %4, vcc, %4, %4\n v_add_co_u32 %5, vcc, %5, %5\n v_add_co_u32 %6, vcc, %6, %6\n v_add_co_u32 %7, vcc, %7, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : : "vcc"); })
BENCH_KERNEL(k_saveexec, {
    asm volatile("v_add_co_u32 %0, vcc, %0, %0\n s_andn2_saveexec_b64 s[20:21], vcc\n v_add_u32 %1, %1, %2\n s_or_b64 exec, exec, s[20:21]\n"
                 "v_add_co_u32 %3, vcc, %3, %3\n s_andn2_saveexec_b64 s[20:21], vcc\n v_add_u32 %2, %2, %1\n s_or_b64 exec, exec, s[20:21]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc", "s20", "s21"); })
BENCH_KERNEL(k_mulu64u32, {
    asm volatile("v_mad_u64_u32 %0, vcc, %4, 1, %0\n v_mad_u64_u32 %1, vcc, %5, 1, %1\n v_mad_u64_u32 %2, vcc, %6, 1, %2\n v_mad_u64_u32 %3, vcc, %7, 1, %3\n"
                 "v_mad_u64_u32 %0, vcc, %5, 1, %0\n v_mad_u64_u32 %1, vcc, %6, 1, %1\n v_mad_u64_u32 %2, vcc, %7, 1, %2\n v_mad_u64_u32 %3, vcc, %4, 1, %3\n"
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc"); })


BENCH_KERNEL(k_cmp_sdwa, {
    asm volatile("v_cmp_ne_u32_sdwa s[20:21], %0, %1 src0_sel:BYTE_0 src1_sel:BYTE_0\n v_cmp_ne_u32_sdwa s[22:23], %0, %1 src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                 "v_cmp_ne_u32_sdwa s[24:25], %0, %1 src0_sel:BYTE_2 src1_sel:BYTE_2\n v_cmp_ne_u32_sdwa s[26:27], %0, %1 src0_sel:BYTE_3 src1_sel:BYTE_3\n"
                 "v_cmp_ne_u32_sdwa s[20:21], %2, %3 src0_sel:BYTE_0 src1_sel:BYTE_0\n v_cmp_ne_u32_sdwa s[22:23], %2, %3 src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                 "v_cmp_ne_u32_sdwa s[24:25], %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_2\n v_cmp_ne_u32_sdwa s[26:27], %2, %3 src0_sel:BYTE_3 src1_sel:BYTE_3\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); })
BENCH_KERNEL(k_salu_or64, {
    asm volatile("s_or_b64 s[20:21], s[20:21], s[22:23]\n s_lshl_b64 s[22:23], s[22:23], 1\n s_or_b64 s[24:25], s[24:25], s[26:27]\n s_lshl_b64 s[26:27], s[26:27], 1\n"
                 "s_or_b64 s[20:21], s[20:21], s[26:27]\n s_bcnt1_i32_b64 s28, s[22:23]\n s_or_b64 s[24:25], s[24:25], s[22:23]\n s_add_u32 s29, s29, s28\n"
                 : "+v"(a0) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc"); })
BENCH_KERNEL(k_mixed_valu_salu, {
    asm volatile("v_add_u32 %0, %0, %1\n s_or_b64 s[20:21], s[20:21], s[22:23]\n v_add_u32 %2, %2, %3\n s_lshl_b64 s[22:23], s[22:23], 1\n"
                 "v_add_u32 %1, %1, %0\n s_bcnt1_i32_b64 s28, s[22:23]\n v_add_u32 %3, %3, %2\n s_add_u32 s29, s29, s28\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23", "s28", "s29", "scc"); })
BENCH_KERNEL(k_mixed_valu_2salu, {
    asm volatile("v_add_u32 %0, %0, %1\n s_or_b64 s[20:21], s[20:21], s[22:23]\n s_lshl_b64 s[22:23], s[22:23], 1\n s_bcnt1_i32_b64 s28, s[22:23]\n"
                 "v_add_u32 %1, %1, %0\n s_add_u32 s29, s29, s28\n s_or_b64 s[24:25], s[24:25], s[22:23]\n s_andn2_b64 s[26:27], s[24:25], s[20:21]\n"
                 : "+v"(a0), "+v"(a1) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc"); })
BENCH_KERNEL(k_lshl_or, {
    asm volatile("v_lshl_or_b32 %0, %0, 2, %1\n v_lshl_or_b32 %2, %2, 2, %3\n v_lshl_or_b32 %4, %4, 2, %5\n v_lshl_or_b32 %6, %6, 2, %7\n"
                 "v_lshl_or_b32 %1, %1, 2, %0\n v_lshl_or_b32 %3, %3, 2, %2\n v_lshl_or_b32 %5, %5, 2, %4\n v_lshl_or_b32 %7, %7, 2, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_bfe, {
    asm volatile("v_bfe_u32 %0, %1, 3, 10\n v_bfe_u32 %2, %3, 3, 10\n v_bfe_u32 %4, %5, 3, 10\n v_bfe_u32 %6, %7, 3, 10\n"
                 "v_bfe_u32 %1, %0, 5, 10\n v_bfe_u32 %3, %2, 5, 10\n v_bfe_u32 %5, %4, 5, 10\n v_bfe_u32 %7, %6, 5, 10\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_cndmask, {
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                 "v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %7, %7, %6, vcc\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) : : ); })
BENCH_KERNEL(k_lshr32, {
    asm volatile("v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %4, 3, %5\n v_lshrrev_b32 %6, 3, %7\n"
                 "v_lshrrev_b32 %1, 5, %0\n v_lshrrev_b32 %3, 5, %2\n v_lshrrev_b32 %5, 5, %4\n v_lshrrev_b32 %7, 5, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_and_or, {
    asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %2, %2, %3, %4\n v_and_or_b32 %4, %4, %5, %6\n v_and_or_b32 %6, %6, %7, %0\n"
                 "v_and_or_b32 %1, %1, %0, %3\n v_and_or_b32 %3, %3, %2, %5\n v_and_or_b32 %5, %5, %4, %7\n v_and_or_b32 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })

BENCH_KERNEL(k_xor32, {
    asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7\n"
                 "v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n v_xor_b32 %5, %5, %4\n v_xor_b32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_min_u32, {
    asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %2, %2, %3\n v_min_u32 %4, %4, %5\n v_min_u32 %6, %6, %7\n"
                 "v_min_u32 %1, %1, %0\n v_min_u32 %3, %3, %2\n v_min_u32 %5, %5, %4\n v_min_u32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_mul_u24, {
    asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %6, %6, %7\n"
                 "v_mul_u32_u24 %1, %1, %0\n v_mul_u32_u24 %3, %3, %2\n v_mul_u32_u24 %5, %5, %4\n v_mul_u32_u24 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_lshlrev, {
    asm volatile("v_lshlrev_b32 %0, %0, %1\n v_lshlrev_b32 %2, %2, %3\n v_lshlrev_b32 %4, %4, %5\n v_lshlrev_b32 %6, %6, %7\n"
                 "v_lshlrev_b32 %1, %1, %0\n v_lshlrev_b32 %3, %3, %2\n v_lshlrev_b32 %5, %5, %4\n v_lshlrev_b32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_lshl_const, {
    asm volatile("v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %4, 3, %5\n v_lshlrev_b32 %6, 3, %7\n"
                 "v_lshlrev_b32 %1, 5, %0\n v_lshlrev_b32 %3, 5, %2\n v_lshlrev_b32 %5, 5, %4\n v_lshlrev_b32 %7, 5, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_lshr_vgpr, {
    asm volatile("v_lshrrev_b32 %0, %0, %1\n v_lshrrev_b32 %2, %2, %3\n v_lshrrev_b32 %4, %4, %5\n v_lshrrev_b32 %6, %6, %7\n"
                 "v_lshrrev_b32 %1, %1, %0\n v_lshrrev_b32 %3, %3, %2\n v_lshrrev_b32 %5, %5, %4\n v_lshrrev_b32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_brev, {
    asm volatile("v_bfrev_b32 %0, %1\n v_bfrev_b32 %2, %3\n v_bfrev_b32 %4, %5\n v_bfrev_b32 %6, %7\n"
                 "v_bfrev_b32 %1, %0\n v_bfrev_b32 %3, %2\n v_bfrev_b32 %5, %4\n v_bfrev_b32 %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_min_f64, {
    asm volatile("v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %2\n v_min_f64 %2, %2, %3\n v_min_f64 %3, %3, %0\n"
                 "v_min_f64 %0, %0, %2\n v_min_f64 %1, %1, %3\n v_min_f64 %2, %2, %0\n v_min_f64 %3, %3, %1\n"
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)); })
BENCH_KERNEL(k_mov, {
    asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n"
                 "v_mov_b32 %1, %0\n v_mov_b32 %3, %2\n v_mov_b32 %5, %4\n v_mov_b32 %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
#define UB_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
BENCH_KERNEL(k_min_dpp, {
    asm volatile("v_min_u32_dpp %0, %1, %0" UB_DPP "v_min_u32_dpp %2, %3, %2" UB_DPP "v_min_u32_dpp %4, %5, %4" UB_DPP "v_min_u32_dpp %6, %7, %6" UB_DPP
                 "v_min_u32_dpp %1, %0, %1" UB_DPP "v_min_u32_dpp %3, %2, %3" UB_DPP "v_min_u32_dpp %5, %4, %5" UB_DPP "v_min_u32_dpp %7, %6, %7" UB_DPP
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_cndmask_dpp, {
    asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc" UB_DPP "v_cndmask_b32_dpp %2, %3, %2, vcc" UB_DPP "v_cndmask_b32_dpp %4, %5, %4, vcc" UB_DPP "v_cndmask_b32_dpp %6, %7, %6, vcc" UB_DPP
                 "v_cndmask_b32_dpp %1, %0, %1, vcc" UB_DPP "v_cndmask_b32_dpp %3, %2, %3, vcc" UB_DPP "v_cndmask_b32_dpp %5, %4, %5, vcc" UB_DPP "v_cndmask_b32_dpp %7, %6, %7, vcc" UB_DPP
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_subb_dpp, {
    asm volatile("v_subb_co_u32_dpp %0, vcc, %1, %0, vcc" UB_DPP "v_subb_co_u32_dpp %2, vcc, %3, %2, vcc" UB_DPP "v_subb_co_u32_dpp %4, vcc, %5, %4, vcc" UB_DPP "v_subb_co_u32_dpp %6, vcc, %7, %6, vcc" UB_DPP
                 "v_subb_co_u32_dpp %1, vcc, %0, %1, vcc" UB_DPP "v_subb_co_u32_dpp %3, vcc, %2, %3, vcc" UB_DPP "v_subb_co_u32_dpp %5, vcc, %4, %5, vcc" UB_DPP "v_subb_co_u32_dpp %7, vcc, %6, %7, vcc" UB_DPP
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) : : "vcc"); })
// the strand pick of the wide builds as the kernel issues it (s_nop; borrow compare; min; select; mask to an SGPR pair), two per iteration
BENCH_KERNEL(k_pick_dpp, {
    asm volatile("s_mov_b64 vcc, -1\n v_subb_co_u32_dpp %4, vcc, %0, %1, vcc" UB_DPP "v_min_u32_dpp %5, %0, %1" UB_DPP "v_cndmask_b32_dpp %6, %2, %3, vcc" UB_DPP "s_mov_b64 s[20:21], vcc\n"
                 "s_mov_b64 vcc, -1\n v_subb_co_u32_dpp %4, vcc, %1, %0, vcc" UB_DPP "v_min_u32_dpp %7, %1, %0" UB_DPP "v_cndmask_b32_dpp %6, %3, %2, vcc" UB_DPP "s_mov_b64 s[22:23], vcc\n"
                 "v_xor_b32 %0, %0, %5\n v_xor_b32 %1, %1, %7\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) : : "vcc", "s20", "s21", "s22", "s23"); })
// ... and with plain instructions plus the two moves they save
BENCH_KERNEL(k_pick_mov, {
    asm volatile("v_mov_b32_dpp %4, %0" UB_DPP "v_mov_b32_dpp %6, %2" UB_DPP "v_cmp_le_u32 s[20:21], %4, %1\n v_min_u32 %5, %4, %1\n v_cndmask_b32 %6, %6, %3, s[20:21]\n"
                 "v_mov_b32_dpp %4, %1" UB_DPP "v_mov_b32_dpp %6, %3" UB_DPP "v_cmp_le_u32 s[22:23], %4, %0\n v_min_u32 %7, %4, %0\n v_cndmask_b32 %6, %6, %2, s[22:23]\n"
                 "v_xor_b32 %0, %0, %5\n v_xor_b32 %1, %1, %7\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) : : "vcc", "s20", "s21", "s22", "s23"); })
BENCH_KERNEL(k_sub, {
    asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %2, %2, %3\n v_sub_u32 %4, %4, %5\n v_sub_u32 %6, %6, %7\n"
                 "v_sub_u32 %1, %1, %0\n v_sub_u32 %3, %3, %2\n v_sub_u32 %5, %5, %4\n v_sub_u32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_dot4, {
    asm volatile("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %2, %2, %3, %4\n v_dot4_u32_u8 %4, %4, %5, %6\n v_dot4_u32_u8 %6, %6, %7, %0\n"
                 "v_dot4_u32_u8 %1, %1, %0, %3\n v_dot4_u32_u8 %3, %3, %2, %5\n v_dot4_u32_u8 %5, %5, %4, %7\n v_dot4_u32_u8 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_mad_u24, {
    asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %6, %6, %7, %0\n"
                 "v_mad_u32_u24 %1, %1, %0, %3\n v_mad_u32_u24 %3, %3, %2, %5\n v_mad_u32_u24 %5, %5, %4, %7\n v_mad_u32_u24 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_add3, {
    asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %6, %6, %7, %0\n"
                 "v_add3_u32 %1, %1, %0, %3\n v_add3_u32 %3, %3, %2, %5\n v_add3_u32 %5, %5, %4, %7\n v_add3_u32 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_or3, {
    asm volatile("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %2, %2, %3, %4\n v_or3_b32 %4, %4, %5, %6\n v_or3_b32 %6, %6, %7, %0\n"
                 "v_or3_b32 %1, %1, %0, %3\n v_or3_b32 %3, %3, %2, %5\n v_or3_b32 %5, %5, %4, %7\n v_or3_b32 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_xad, {
    asm volatile("v_xad_u32 %0, %0, %1, %2\n v_xad_u32 %2, %2, %3, %4\n v_xad_u32 %4, %4, %5, %6\n v_xad_u32 %6, %6, %7, %0\n"
                 "v_xad_u32 %1, %1, %0, %3\n v_xad_u32 %3, %3, %2, %5\n v_xad_u32 %5, %5, %4, %7\n v_xad_u32 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_pk_add_u16, {
    asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %6, %6, %7\n"
                 "v_pk_add_u16 %1, %1, %0\n v_pk_add_u16 %3, %3, %2\n v_pk_add_u16 %5, %5, %4\n v_pk_add_u16 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_pk_min_u16, {
    asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_min_u16 %2, %2, %3\n v_pk_min_u16 %4, %4, %5\n v_pk_min_u16 %6, %6, %7\n"
                 "v_pk_min_u16 %1, %1, %0\n v_pk_min_u16 %3, %3, %2\n v_pk_min_u16 %5, %5, %4\n v_pk_min_u16 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_sad_u8, {
    asm volatile("v_sad_u8 %0, %0, %1, %2\n v_sad_u8 %2, %2, %3, %4\n v_sad_u8 %4, %4, %5, %6\n v_sad_u8 %6, %6, %7, %0\n"
                 "v_sad_u8 %1, %1, %0, %3\n v_sad_u8 %3, %3, %2, %5\n v_sad_u8 %5, %5, %4, %7\n v_sad_u8 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_msad, {
    asm volatile("v_msad_u8 %0, %0, %1, %2\n v_msad_u8 %2, %2, %3, %4\n v_msad_u8 %4, %4, %5, %6\n v_msad_u8 %6, %6, %7, %0\n"
                 "v_msad_u8 %1, %1, %0, %3\n v_msad_u8 %3, %3, %2, %5\n v_msad_u8 %5, %5, %4, %7\n v_msad_u8 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_cvt_pk_u8, {
    asm volatile("v_lerp_u8 %0, %0, %1, %2\n v_lerp_u8 %2, %2, %3, %4\n v_lerp_u8 %4, %4, %5, %6\n v_lerp_u8 %6, %6, %7, %0\n"
                 "v_lerp_u8 %1, %1, %0, %3\n v_lerp_u8 %3, %3, %2, %5\n v_lerp_u8 %5, %5, %4, %7\n v_lerp_u8 %7, %7, %6, %1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })

__global__ __launch_bounds__(256) void k_ldsadd(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t h[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
    __syncthreads();
    uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            atomicAdd(&h[(x >> 20) & 4095], 1u);
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_ldsadd_lcg_only(uint32_t *out, uint32_t seed)
{
    uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x, acc = 0;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) { x = x * 1664525u + 1013904223u; acc ^= (x >> 20) & 4095; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}


BENCH_KERNEL(k_cndmask_e64, {
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[22:23]\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[22:23]\n"
                 "v_cndmask_b32_e64 %1, %1, %0, s[22:23]\n v_cndmask_b32_e64 %3, %3, %2, s[20:21]\n v_cndmask_b32_e64 %5, %5, %4, s[22:23]\n v_cndmask_b32_e64 %7, %7, %6, s[20:21]\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) : : "s20", "s21", "s22", "s23"); })
BENCH_KERNEL(k_min_sdwa, {
    asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_min_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                 "v_min_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_min_u32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                 "v_min_u32_sdwa %1, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_min_u32_sdwa %3, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                 "v_min_u32_sdwa %5, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_min_u32_sdwa %7, %7, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_and_literal, {
    asm volatile("v_and_b32 %0, 0x3ffc, %1\n v_and_b32 %2, 0x3ffc, %3\n v_and_b32 %4, 0x3ffc, %5\n v_and_b32 %6, 0x3ffc, %7\n"
                 "v_and_b32 %1, 0xdfdfdfdf, %0\n v_and_b32 %3, 0xdfdfdfdf, %2\n v_and_b32 %5, 0xdfdfdfdf, %4\n v_and_b32 %7, 0xdfdfdfdf, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_and_sgpr, {
    asm volatile("v_and_b32 %0, s20, %1\n v_and_b32 %2, s20, %3\n v_and_b32 %4, s20, %5\n v_and_b32 %6, s20, %7\n"
                 "v_and_b32 %1, s21, %0\n v_and_b32 %3, s21, %2\n v_and_b32 %5, s21, %4\n v_and_b32 %7, s21, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) : : "s20", "s21"); })
// LDS atomics: every thread on its own cell (conflict-free), 8 per iteration, non-returning
__global__ __launch_bounds__(256) void k_ds_add_own(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t cells[256];
    cells[threadIdx.x] = seed;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)&cells[threadIdx.x], one = 1;
    const uint64_t c0_ = clock64(), w0_ = wall_clock64();
    for (int i = 0; i < ITERS; i++)
        asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n ds_add_u32 %0, %1\n" : : "v"(addr), "v"(one) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1 << 20] = (uint32_t)(clock64() - c0_); out[(1 << 20) + 1] = (uint32_t)(wall_clock64() - w0_); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = cells[threadIdx.x];
}
// random cells of a 16 KiB histogram, addresses precomputed (8 per thread, reused)
__global__ __launch_bounds__(256) void k_ds_add_rand(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t cells[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) cells[i] = 0;
    __syncthreads();
    uint32_t ad[8]; uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int i = 0; i < 8; i++) { x = x * 1664525u + 1013904223u; ad[i] = (uint32_t)(uintptr_t)&cells[(x >> 12) & 4095]; }
    const uint32_t one = 1;
    const uint64_t c0_ = clock64(), w0_ = wall_clock64();
    for (int i = 0; i < ITERS; i++)
        asm volatile("ds_add_u32 %0, %8\n ds_add_u32 %1, %8\n ds_add_u32 %2, %8\n ds_add_u32 %3, %8\n ds_add_u32 %4, %8\n ds_add_u32 %5, %8\n ds_add_u32 %6, %8\n ds_add_u32 %7, %8\n"
                     : : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(one) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1 << 20] = (uint32_t)(clock64() - c0_); out[(1 << 20) + 1] = (uint32_t)(wall_clock64() - w0_); }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = cells[threadIdx.x];
}
// functional: what does a non-returning LDS atomic do with address bits 1:0 ?
__global__ void k_ds_misaligned(uint32_t *out)
{
    __shared__ uint32_t cells[64];
    cells[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t addr = (uint32_t)(uintptr_t)&cells[8 * threadIdx.x + 1] + threadIdx.x, v = 0x01010101u;  // byte offsets 0..3 inside a cell
        asm volatile("ds_add_u32 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
    }
    __syncthreads();
    out[threadIdx.x] = cells[threadIdx.x];
}

// ---- instruction MIXES (do full-rate ops keep their rate between half-rate ops / SALU / exec switches?) ----
BENCH_KERNEL(k_mix_align_xor, {
    asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_xor_b32 %2, %2, %3\n v_alignbit_b32 %4, %4, %5, 7\n v_xor_b32 %6, %6, %7\n"
                 "v_alignbit_b32 %1, %1, %0, 9\n v_xor_b32 %3, %3, %2\n v_alignbit_b32 %5, %5, %4, 9\n v_xor_b32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_mix_align2_xor2, {
    asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %4, %4, %5, 7\n v_xor_b32 %2, %2, %3\n v_xor_b32 %6, %6, %7\n"
                 "v_alignbit_b32 %1, %1, %0, 9\n v_alignbit_b32 %5, %5, %4, 9\n v_xor_b32 %3, %3, %2\n v_xor_b32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
BENCH_KERNEL(k_mix_xor_and_lshr, {
    asm volatile("v_xor_b32 %0, %0, %1\n v_and_b32 %2, 0x3ffc, %3\n v_lshrrev_b32 %4, 3, %5\n v_xor_b32 %6, %6, %7\n"
                 "v_and_b32 %1, 0xdfdfdfdf, %0\n v_lshrrev_b32 %3, 5, %2\n v_xor_b32 %5, %5, %4\n v_and_b32 %7, 0x6060606, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })

BENCH_KERNEL(k_mix_run8, {
    asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %4, %4, %5, 7\n v_alignbit_b32 %6, %6, %7, 7\n"
                 "v_alignbit_b32 %1, %1, %0, 9\n v_alignbit_b32 %3, %3, %2, 9\n v_alignbit_b32 %5, %5, %4, 9\n v_alignbit_b32 %7, %7, %6, 9\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
    if (i & 1) continue;
    asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7\n"
                 "v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n v_xor_b32 %5, %5, %4\n v_xor_b32 %7, %7, %6\n"
                 "v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7\n"
                 "v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n v_xor_b32 %5, %5, %4\n v_xor_b32 %7, %7, %6\n"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3)); })
// the masked region of scan2_kernel, 4 positions per iteration: exec <- V; hist atomic; digests; exec <- V & F; cell atomic
#define UB_LDS_PROLOGUE                                                                                         \
    __shared__ uint32_t cells[4096 + 256];                                                                      \
    for (int i = threadIdx.x; i < 4096 + 256; i += 256) cells[i] = 0;                                           \
    __syncthreads();                                                                                            \
    uint32_t ad[4]; uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;                        \
    for (int i = 0; i < 4; i++) { x = x * 1664525u + 1013904223u; ad[i] = (uint32_t)(uintptr_t)&cells[(x >> 12) & 4095]; } \
    const uint32_t cell = (uint32_t)(uintptr_t)&cells[4096 + threadIdx.x], one = 1;                             \
    uint64_t sum = seed; uint32_t xlo = seed, nf = 0, lo0 = x, lo1 = x * 3, lo2 = x * 5, lo3 = x * 7;            \
    const uint64_t V0 = 0xFFFFFFFF0FFFFFFFull ^ seed, V1 = 0xFFF0FFFFFFFFFFFFull ^ seed, V2 = ~0ull ^ seed, V3 = 0xFFFFFFFFFFFF00FFull ^ seed; \
    const uint64_t F0 = 0x5555555555555555ull * seed, F1 = 0x3333333333333333ull * seed, F2 = 0x0F0F0F0F0F0F0F0Full * seed, F3 = 0x00FF00FF00FF00FFull * seed; \
    const uint64_t c0_ = clock64(), w0_ = wall_clock64();
#define UB_LDS_EPILOGUE                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1 << 20] = (uint32_t)(clock64() - c0_); out[(1 << 20) + 1] = (uint32_t)(wall_clock64() - w0_); } \
    __syncthreads();                                                                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = cells[threadIdx.x] + (uint32_t)sum + (uint32_t)(sum >> 32) + xlo + nf + cells[4096 + threadIdx.x];
#define UB_OPS : "+v"(sum), "+v"(xlo), "+v"(nf) : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(lo0), "v"(lo1), "v"(lo2), "v"(lo3), \
                 "s"(V0), "s"(V1), "s"(V2), "s"(V3), "s"(F0), "s"(F1), "s"(F2), "s"(F3), "v"(cell), "v"(one) : "memory", "vcc", "scc"
// operands: 0 sum 1 xlo 2 nf 3-6 ad 7-10 lo 11-14 V 15-18 F 19 cell 20 one
#define UB_P6A(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %20\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n s_and_b64 exec, %" #V ", %" #F "\n ds_add_u32 %19, %20\n"
__global__ __launch_bounds__(256) void k_p6a(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6A(0, 3, 7, 11, 15) UB_P6A(1, 4, 8, 12, 16) UB_P6A(2, 5, 9, 13, 17) UB_P6A(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}
// b: no LDS at all (exec switches + VALU digests only)
#define UB_P6B(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n s_and_b64 exec, %" #V ", %" #F "\n v_add_u32 %2, %2, %20\n"
__global__ __launch_bounds__(256) void k_p6b(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6B(0, 3, 7, 11, 15) UB_P6B(1, 4, 8, 12, 16) UB_P6B(2, 5, 9, 13, 17) UB_P6B(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}
// c: hist atomic + VALU forward count (one LDS op per position)
#define UB_P6C(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %20\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n s_and_b64 exec, %" #V ", %" #F "\n v_add_u32 %2, %2, %20\n"
__global__ __launch_bounds__(256) void k_p6c(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6C(0, 3, 7, 11, 15) UB_P6C(1, 4, 8, 12, 16) UB_P6C(2, 5, 9, 13, 17) UB_P6C(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}
// d: only the exec switches and the LDS atomics (no VALU)
#define UB_P6D(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %20\n s_and_b64 exec, %" #V ", %" #F "\n ds_add_u32 %19, %20\n"
__global__ __launch_bounds__(256) void k_p6d(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6D(0, 3, 7, 11, 15) UB_P6D(1, 4, 8, 12, 16) UB_P6D(2, 5, 9, 13, 17) UB_P6D(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}
// e: no exec switching at all: values zeroed by v_cndmask with the SGPR mask, digests and counts unmasked, hist atomic to a dump cell... (VALU only)
#define UB_P6E(i, ad, lo, V, F) "v_cndmask_b32_e64 %2, 0, %" #lo ", %" #V "\n v_mad_u64_u32 %0, vcc, %2, 1, %0\n v_xor_b32 %1, %1, %2\n"
__global__ __launch_bounds__(256) void k_p6e(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6E(0, 3, 7, 11, 15) UB_P6E(1, 4, 8, 12, 16) UB_P6E(2, 5, 9, 13, 17) UB_P6E(3, 6, 10, 14, 18) UB_OPS);
    UB_LDS_EPILOGUE
}
// f: exec switch + VALU digests only, no strand part
#define UB_P6F(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n"
__global__ __launch_bounds__(256) void k_p6f(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6F(0, 3, 7, 11, 15) UB_P6F(1, 4, 8, 12, 16) UB_P6F(2, 5, 9, 13, 17) UB_P6F(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}
// g: like f with add_co/addc instead of the mad
#define UB_P6G(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n v_lshl_add_u64 %0, %0, 0, %0\n v_xor_b32 %1, %1, %" #lo "\n"
__global__ __launch_bounds__(256) void k_p6g(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6G(0, 3, 7, 11, 15) UB_P6G(1, 4, 8, 12, 16) UB_P6G(2, 5, 9, 13, 17) UB_P6G(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}
// h: only exec switches + xor (one full-rate VALU per exec switch)
#define UB_P6H(i, ad, lo, V, F) "s_mov_b64 exec, %" #V "\n v_xor_b32 %1, %1, %" #lo "\n"
__global__ __launch_bounds__(256) void k_p6h(uint32_t *out, uint32_t seed)
{
    UB_LDS_PROLOGUE
    for (int i = 0; i < ITERS; i++)
        asm volatile(UB_P6H(0, 3, 7, 11, 15) UB_P6H(1, 4, 8, 12, 16) UB_P6H(2, 5, 9, 13, 17) UB_P6H(3, 6, 10, 14, 18) "s_mov_b64 exec, -1\n" UB_OPS);
    UB_LDS_EPILOGUE
}

int main()
{
    uint32_t *d;
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    CHK(hipMalloc(&d, (size_t)blocks * 256 * 4 + (8 << 20)));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    struct B { const char *name; void (*fn)(uint32_t *, uint32_t); };
    B list[] = {{"v_add_u32", k_add32}, {"v_alignbit_b32", k_alignbit}, {"v_perm_b32", k_perm}, {"v_bitop3_b32", k_bitop3},
                {"v_mov_dpp wave_shr", k_dpp_shr}, {"v_mov_dpp row_shr", k_dpp_rowshr}, {"v_lshrrev_b64", k_lshr64},
                {"v_lshl_add_u64", k_lshladd64}, {"v_cmp_lt_u64", k_cmp64}, {"v_cmp_lt_u32", k_cmp32},
                {"cmp32+cndmask pair", k_cmp_cndmask}, {"v_add_co_u32", k_addco}, {"addco+saveexec+add+restore (per 4)", k_saveexec},
                {"v_mad_u64_u32", k_mulu64u32}, {"v_cmp_ne_u32_sdwa (byte sel)", k_cmp_sdwa}, {"SALU only (or/lshl/bcnt/add b64)", k_salu_or64},
                {"VALU add + SALU 1:1", k_mixed_valu_salu}, {"VALU add + SALU 1:3", k_mixed_valu_2salu}, {"v_lshl_or_b32", k_lshl_or}, {"v_bfe_u32", k_bfe},
                {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32_e64 sgpr", k_cndmask_e64}, {"v_min_u32_sdwa word1", k_min_sdwa}, {"v_and_b32 literal", k_and_literal}, {"v_and_b32 sgpr", k_and_sgpr}, {"ds_add_u32 own cell", k_ds_add_own}, {"ds_add_u32 random cell", k_ds_add_rand}, {"mix alignbit/xor alternating", k_mix_align_xor}, {"mix alignbit x2 / xor x2", k_mix_align2_xor2}, {"mix xor/and-literal/lshr", k_mix_xor_and_lshr}, {"mix runs: 8 alignbit then 8 xor (16 per 8 counted -> x0.5)", k_mix_run8},
                {"P6a masked region x4 (x2 = cycles per position)", k_p6a}, {"P6b no LDS, v_add count (x2/pos)", k_p6b}, {"P6c hist LDS + v_add count (x2/pos)", k_p6c}, {"P6d exec + 2 LDS only (x2/pos)", k_p6d}, {"P6e cndmask-zero, no exec (x2/pos)", k_p6e}, {"P6f exec + mad + xor (x2/pos)", k_p6f}, {"P6g exec + lshl_add_u64 + xor (x2/pos)", k_p6g}, {"P6h exec + xor (x2/pos)", k_p6h}, {"v_lshrrev_b32", k_lshr32}, {"v_and_or_b32", k_and_or}, {"v_xor_b32", k_xor32}, {"v_min_u32", k_min_u32}, {"v_mul_u32_u24", k_mul_u24}, {"v_lshlrev_b32", k_lshlrev}, {"v_lshlrev_b32 by a constant", k_lshl_const}, {"v_lshrrev_b32 by a VGPR", k_lshr_vgpr}, {"v_bfrev_b32", k_brev}, {"v_min_f64", k_min_f64}, {"v_mov_b32", k_mov}, {"v_min_u32_dpp", k_min_dpp}, {"v_cndmask_b32_dpp", k_cndmask_dpp}, {"v_subb_co_u32_dpp", k_subb_dpp}, {"strand pick x2, DPP-fused (8 VALU counted; x4 = cycles per pick)", k_pick_dpp}, {"strand pick x2, moves + plain (8 counted of 12; x4 = cycles per pick)", k_pick_mov}, {"v_sub_u32", k_sub}, {"v_dot4_u32_u8", k_dot4}, {"v_mad_u32_u24", k_mad_u24}, {"v_add3_u32", k_add3}, {"v_or3_b32", k_or3}, {"v_xad_u32", k_xad}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_min_u16", k_pk_min_u16}, {"v_sad_u8", k_sad_u8}, {"v_msad_u8", k_msad}, {"v_lerp_u8", k_cvt_pk_u8}, {"lds atomicAdd random bin (+lcg)", k_ldsadd}, {"lcg only", k_ldsadd_lcg_only}};
    // (k_ds_misaligned is not run: on gfx950 a ds_add_u32 whose address is not 4-byte aligned raises a memory violation - measured, r02a)
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (auto &b : list) {
        hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
        CHK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD: blocks*4 waves / (CUs*4 SIMDs) * ITERS * 8
        double winstr = (double)blocks * 4 / (prop.multiProcessorCount * 4) * ITERS * 8;
        double ns_per = best * 1e6 / winstr;
        uint32_t ck[2] = {0, 1}; hipMemcpy(ck, d + (1 << 20), 8, hipMemcpyDeviceToHost);
        const double ghz = ck[0] / (ck[1] * 10.0);
        printf("%-42s %8.3f ms  %7.3f ns per wave-instr per SIMD  clock %.3f GHz -> %5.2f cycles\n", b.name, best, ns_per, ghz, ns_per * ghz);
    }
    return 0;
}
