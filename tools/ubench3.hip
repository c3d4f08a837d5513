// ubench3.hip (round 3) - LDS-side micro-benchmarks for the scan2 masked region on gfx950:
//   * ds_append: what it does (LDS[M0[15:0] + offset] += popcount(exec)) and what it costs next to a ds_add_u32 on the thread's own cell;
//   * the histogram atomic with random cells (32 lanes of a lane group over 32 banks) against lane-privatised cells
//     (cell = (random 12 bits : lane & 3): four disjoint bank sets);
//   * the whole masked region (exec <- V; hist atomic; mad; xor; exec <- V & F; forward count) in the shipped form and with either change.
// Launch shape of the kernel: 768-thread blocks, 68 KiB of LDS each, two per CU = 6 waves per SIMD.  Prints SIMD-cycles per position.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

// functional: 4 waves, lanes 0..39 of each active; every wave appends 3 times to its own counter, placed beyond 64 KiB
__global__ __launch_bounds__(256) void k_append_check(uint32_t *out)
{
    __shared__ uint32_t big[17 * 1024 + 64];   // 68 KiB: the counters sit at byte offsets >= 65536 like the kernel's
    for (int i = threadIdx.x; i < 17 * 1024 + 64; i += 256) big[i] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t addr = (uint32_t)(uintptr_t)&big[16 * 1024 + 16 + wave];
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(addr - 32768u);
    uint32_t ret = 0xDEAD;
    if (lane < 40) {
        asm volatile("s_mov_b32 m0, %1\n ds_append %0 offset:32768\n ds_append %0 offset:32768\n ds_append %0 offset:32768\n s_waitcnt lgkmcnt(0)"
                     : "+v"(ret) : "s"(m0v) : "memory");
    }
    __syncthreads();
    if (lane == 0) { out[wave * 4] = big[16 * 1024 + 16 + wave]; out[wave * 4 + 1] = ret; out[wave * 4 + 2] = addr; }
    if (threadIdx.x == 0) { uint32_t nz = 0; for (int i = 0; i < 17 * 1024 + 64; i++) nz += big[i] != 0; out[32] = nz; }
}

// which LDS word does ds_append touch?  mode 0: M0 = addr, offset 0; 1: M0 = addr << 16, offset 0; 2: M0 = (addr << 16) | 0xFFFF, offset 0;
// 3: M0 = 0, offset = 4096; 4: M0 = (addr >> 2) << 16, offset 0; 5: M0 = addr << 16 | size 0x1000, offset 8
__global__ __launch_bounds__(64) void k_append_probe(uint32_t *out, int mode)
{
    __shared__ uint32_t big[17 * 1024 + 64];
    for (int i = threadIdx.x; i < 17 * 1024 + 64; i += 64) big[i] = 0;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)&big[1000];   // byte address 4000 (+ the array's base)
    uint32_t m0v = mode == 0 ? addr : mode == 1 ? addr << 16 : mode == 2 ? ((addr << 16) | 0xFFFFu) : mode == 3 ? 0u : mode == 4 ? ((addr >> 2) << 16) : ((addr << 16) | 0x1000u);
    m0v = __builtin_amdgcn_readfirstlane(m0v);
    uint32_t ret = 0xDEAD;
    if (mode == 3) asm volatile("s_mov_b32 m0, %1\n ds_append %0 offset:4096\n s_waitcnt lgkmcnt(0)" : "+v"(ret) : "s"(m0v) : "memory");
    else if (mode == 5) asm volatile("s_mov_b32 m0, %1\n ds_append %0 offset:8\n s_waitcnt lgkmcnt(0)" : "+v"(ret) : "s"(m0v) : "memory");
    else asm volatile("s_mov_b32 m0, %1\n ds_append %0\n s_waitcnt lgkmcnt(0)" : "+v"(ret) : "s"(m0v) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int idx = -1; uint32_t val = 0;
        for (int i = 0; i < 17 * 1024 + 64; i++) if (big[i]) { idx = i; val = big[i]; }
        out[0] = (uint32_t)idx; out[1] = val; out[2] = addr; out[3] = (uint32_t)(uintptr_t)&big[0];
    }
}

#define UB_PROLOGUE(PRIV)                                                                                        \
    __shared__ uint32_t cells[16384 + 768 + 16];                                                                 \
    for (int i = threadIdx.x; i < 16384 + 768 + 16; i += 768) cells[i] = 0;                                     \
    __syncthreads();                                                                                            \
    uint32_t ad[4]; uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;                        \
    for (int i = 0; i < 4; i++) { x = x * 1664525u + 1013904223u;                                               \
        const uint32_t c = PRIV ? ((((x >> 12) & 4095) << 2) | (threadIdx.x & 3)) : ((x >> 12) & 16383);       \
        ad[i] = (uint32_t)(uintptr_t)&cells[c]; }                                                               \
    const uint32_t cell = (uint32_t)(uintptr_t)&cells[16384 + threadIdx.x], one = 1;                            \
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&cells[16384 + 768 + (threadIdx.x >> 6)] - 32768u); \
    uint32_t sink = 0;                                                                                          \
    uint64_t sum = seed; uint32_t xlo = seed, nf = 0, lo0 = x, lo1 = x * 3, lo2 = x * 5, lo3 = x * 7;            \
    const uint64_t V0 = 0xFFFFFFFF0FFFFFFFull ^ seed, V1 = 0xFFF0FFFFFFFFFFFFull ^ seed, V2 = ~0ull ^ seed, V3 = 0xFFFFFFFFFFFF00FFull ^ seed; \
    const uint64_t F0 = 0x5555555555555555ull * seed, F1 = 0x3333333333333333ull * seed, F2 = 0x0F0F0F0F0F0F0F0Full * seed, F3 = 0x00FF00FF00FF00FFull * seed; \
    const uint64_t c0_ = clock64(), w0_ = wall_clock64();
#define UB_EPILOGUE                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1 << 20] = (uint32_t)(clock64() - c0_); out[(1 << 20) + 1] = (uint32_t)(wall_clock64() - w0_); } \
    __syncthreads();                                                                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = cells[threadIdx.x] + (uint32_t)sum + (uint32_t)(sum >> 32) + xlo + nf + sink + cells[16384 + threadIdx.x] + cells[16384 + 768 + (threadIdx.x & 3)];
#define UB_OPS : "+v"(sum), "+v"(xlo), "+v"(nf), "+v"(sink) : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(lo0), "v"(lo1), "v"(lo2), "v"(lo3), \
                 "s"(V0), "s"(V1), "s"(V2), "s"(V3), "s"(F0), "s"(F1), "s"(F2), "s"(F3), "v"(cell), "v"(one), "s"(m0v) : "memory", "vcc", "scc"
// operands: 0 sum 1 xlo 2 nf 3 sink 4-7 ad 8-11 lo 12-15 V 16-19 F 20 cell 21 one 22 m0v
#define R_SHIP(ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %21\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n s_and_b64 exec, %" #V ", %" #F "\n ds_add_u32 %20, %21\n"
#define R_APP(ad, lo, V, F)  "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %21\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n s_and_b64 exec, %" #V ", %" #F "\n ds_append %3 offset:32768\n"
#define R_SCNT(ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %21\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n s_and_b64 vcc, %" #V ", %" #F "\n s_bcnt1_i32_b64 vcc_lo, vcc\n s_add_u32 s40, s40, vcc_lo\n"
#define R_HIST(ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %21\n"
#define R_CELL(ad, lo, V, F) "s_and_b64 exec, %" #V ", %" #F "\n ds_add_u32 %20, %21\n"
#define R_APPO(ad, lo, V, F) "s_and_b64 exec, %" #V ", %" #F "\n ds_append %3 offset:32768\n"
#define R_NOFW(ad, lo, V, F) "s_mov_b64 exec, %" #V "\n ds_add_u32 %" #ad ", %21\n v_mad_u64_u32 %0, vcc, %" #lo ", 1, %0\n v_xor_b32 %1, %1, %" #lo "\n"
#define REGION_KERNEL(NAME, PRIV, R)                                                                              \
    __global__ __launch_bounds__(768) void NAME(uint32_t *out, uint32_t seed)                                   \
    {                                                                                                           \
        UB_PROLOGUE(PRIV)                                                                                       \
        for (int i = 0; i < ITERS; i++)                                                                         \
            asm volatile("s_mov_b32 m0, %22\n" R(4, 8, 12, 16) R(5, 9, 13, 17) R(6, 10, 14, 18) R(7, 11, 15, 19) "s_mov_b64 exec, -1\n" UB_OPS, "s40"); \
        UB_EPILOGUE                                                                                             \
    }
REGION_KERNEL(k_ship, 0, R_SHIP)
REGION_KERNEL(k_ship_priv, 1, R_SHIP)
REGION_KERNEL(k_app, 0, R_APP)
REGION_KERNEL(k_app_priv, 1, R_APP)
REGION_KERNEL(k_scnt, 0, R_SCNT)
REGION_KERNEL(k_scnt_priv, 1, R_SCNT)
REGION_KERNEL(k_hist, 0, R_HIST)
REGION_KERNEL(k_hist_priv, 1, R_HIST)
REGION_KERNEL(k_cell, 0, R_CELL)
REGION_KERNEL(k_appo, 0, R_APPO)
REGION_KERNEL(k_nofw, 0, R_NOFW)
REGION_KERNEL(k_nofw_priv, 1, R_NOFW)

int main()
{
    uint32_t *d;
    const int blocks = 256 * 2;   // 68 KiB of LDS per block: two 768-thread blocks per CU, 6 waves per SIMD like the kernel
    CHK(hipMalloc(&d, (size_t)(8 << 20)));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
    CHK(hipMemset(d, 0, 4096));
    hipLaunchKernelGGL(k_append_check, dim3(1), dim3(256), 0, 0, d);
    CHK(hipDeviceSynchronize());
    uint32_t h[40]; CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("ds_append check (want 120 per wave, last return 80, exactly 4 non-zero LDS words):");
    for (int w = 0; w < 4; w++) printf("  wave %d: counter %u ret %u @%u", w, h[w * 4], h[w * 4 + 1], h[w * 4 + 2]);
    printf("  non-zero words %u\n", h[32]);
    for (int mode = 0; mode < 6; mode++) {
        CHK(hipMemset(d, 0xFF, 64));
        hipLaunchKernelGGL(k_append_probe, dim3(1), dim3(64), 0, 0, d, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("ds_append probe mode %d: fault\n", mode); return 1; }
        CHK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("ds_append probe mode %d: LDS word %d (byte %d) = %u   [target word 1000 at byte %u, array base %u]\n", mode, (int)h[0], (int)h[0] * 4 + (int)h[3], h[1], h[2], h[3]);
    }
    struct B { const char *name; void (*fn)(uint32_t *, uint32_t); int per; };
    B list[] = {{"region shipped (hist rand + own cell)      /4 pos", k_ship, 4}, {"region shipped, hist privatised            /4 pos", k_ship_priv, 4},
                {"region with ds_append                      /4 pos", k_app, 4}, {"region with ds_append, hist privatised     /4 pos", k_app_priv, 4},
                {"region with scalar popcount                /4 pos", k_scnt, 4}, {"region with scalar popcount, privatised    /4 pos", k_scnt_priv, 4},
                {"region without the forward count           /4 pos", k_nofw, 4}, {"region without the forward count, priv     /4 pos", k_nofw_priv, 4},
                {"exec + hist atomic only, random cells      /4 pos", k_hist, 4}, {"exec + hist atomic only, privatised        /4 pos", k_hist_priv, 4},
                {"exec + own-cell atomic only                /4 pos", k_cell, 4}, {"exec + ds_append only                      /4 pos", k_appo, 4}};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (auto &b : list) {
        hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(768), 0, 0, d, 1u);
        CHK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(768), 0, 0, d, 2u + r);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // positions per SIMD: blocks*4 waves / (CUs*4 SIMDs) * ITERS * 4
        const double pos = (double)blocks * 12 / (prop.multiProcessorCount * 4) * ITERS * b.per;
        uint32_t ck[2] = {0, 1}; hipMemcpy(ck, d + (1 << 20), 8, hipMemcpyDeviceToHost);
        const double ghz = ck[0] / (ck[1] * 10.0);
        printf("%-52s %8.3f ms  clock %.3f GHz -> %6.2f SIMD-cycles per position (6 waves per SIMD)\n", b.name, best, ghz, best * 1e6 / pos * ghz);
    }
    return 0;
}
