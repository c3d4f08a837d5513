// ntk_kernels.hpp — hand-written HIP kernels for gfx950 (CDNA4, wave64).  No CUDA path, no MFMA:
// the work is byte/integer scanning bounded by HBM reads and VALU issue.
//
// The scan engine restates the reference's per-record chain as one streaming pass over a batch:
//   normalize/strip_returns (alphabet part)  reference src/sequence.rs:19-62,165-191
//   reverse_complement                       reference src/sequence.rs:68-105,202-208
//   CanonicalKmers / BitNuclKmer             reference src/kmer.rs:48-130, src/bitkmer.rs:26-143
// Per raw byte only a 2-way class matters on device: "base" (extends the window, 2-bit
// code A0 C1 G2 T3) or "break" (resets it).  Bytes of the pre-step's "deleted" class never reach the
// device (the host packer drops them while copying into the pinned batch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "ntk_tile.hpp"

// Ablation switches (NTK_ABL_*: loads only, floor kernel, no LDS atomics, no digests, no exec writes, no SDWA compares, no mask algebra) and
// the per-wave clock census (NTK_V_CLOCKS) exist for tools/kbench.hip alone, which is built with -DNTK_KBENCH: they produce WRONG results
// by design and measure what a class of the tile loop costs (profiles/r03c/ablation.txt, r04i/ablation.txt).  No product object may see one.
#if !defined(NTK_KBENCH) && (defined(NTK_ABL_LOADSONLY) || defined(NTK_ABL_FLOOR) || defined(NTK_ABL_NOLDS) || defined(NTK_ABL_NODIGEST) || \
                             defined(NTK_ABL_NOEXEC) || defined(NTK_ABL_NOSDWA) || defined(NTK_ABL_NOMASKALG) || defined(NTK_V_CLOCKS) || \
                             defined(NTK_X_CMPFIRST) || defined(NTK_X_TWOPHASE) || defined(NTK_X_MFMASUM) || defined(NTK_X_SELOUT) || defined(NTK_X_PREFETCH2) || defined(NTK_X_FASTSTART) || defined(NTK_ABL_HALFIMPORTS))
#error "NTK_ABL_* / NTK_X_* / NTK_V_CLOCKS are kernel-bench switches: build with -DNTK_KBENCH (tools/build_kbench.sh), never into the library"
#endif

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Cross-lane "value the previous lane holds": DPP wave_shr:1 (lane 0 reads 0), a single VALU move.
constexpr int kDppWaveShr1 = 0x138;
struct DevXL {
    __device__ __forceinline__ uint32_t prev_and(int s, uint32_t x, uint32_t mask) const { return prev(s, x) & mask; }
    __device__ __forceinline__ uint32_t prev_auto(uint32_t x) const { return prev(0, x); }   // (the host emulation numbers these call sites itself)
    __device__ __forceinline__ uint32_t prev(int, uint32_t x) const
    {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, kDppWaveShr1, 0xf, 0xf, true);
    }
    // (the previous lane's x) + c in one DPP add
    __device__ __forceinline__ uint32_t prev_add(int, uint32_t x, uint32_t c) const
    {
        uint32_t r;
        asm("s_nop 1\n v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(c));
        return r;
    }
    // take ? a : (the previous lane's b) - the cross-lane move rides on the select (v_cndmask_b32_dpp, src0 = DPP operand).
    // s_nop 1: a VGPR written by the preceding VALU instruction may not be read through DPP for two wait states, and the
    // compiler does not look inside the asm.
    __device__ __forceinline__ uint32_t select_prev(int, bool take, uint32_t a, uint32_t b) const
    {
        uint32_t r;
        const uint64_t m = __builtin_amdgcn_ballot_w64(take);
        asm("s_nop 1\n s_mov_b64 vcc, %3\n v_cndmask_b32_dpp %0, %2, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "=v"(r) : "v"(a), "v"(b), "s"(m) : "vcc");
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// sinks
// ---------------------------------------------------------------------------------------------
template <int KW>
struct ReduceSink {
    uint64_t sum = 0, xr = 0;
    uint32_t n_fwd = 0, n_valid = 0;
    uint32_t *hist;
    uint32_t bin_shift;

    __device__ __forceinline__ void begin_tile(int64_t, uint32_t inval16, bool) { n_valid += __popc(~inval16 & 0xFFFFu); }
    __device__ __forceinline__ void emit(int, bool valid, bool take_fwd, uint32_t hi, uint32_t lo)
    {
        if (valid) {
            const uint64_t v = KW == 2 ? (((uint64_t)hi << 32) | lo) : (uint64_t)lo;
            sum += v;
            xr ^= v;
            n_fwd += take_fwd ? 1u : 0u;  // v_addc_co_u32 off the compare's carry mask
            atomicAdd(&hist[KW == 2 ? (uint32_t)(v >> bin_shift) : (lo >> bin_shift)], 1u);  // no 64-bit shift for 32-bit values
        }
    }
    __device__ __forceinline__ void end_tile() {}
};

// Values are staged through LDS so that the global stores are whole contiguous 1-KiB pieces (a lane owns 16
// CONSECUTIVE positions, so direct stores would be 64 scattered 8-byte pieces per instruction: measured 1.7x slower).
// Stage layout: row = emitting lane (0..61), 18 u64 per row (16 values + 2 pad: rows stay 16-byte aligned and the
// 16 lanes of a ds_write_b64 group spread over the banks).
constexpr int kStageRow = 18;
constexpr int kStageWaveU64 = kTileSlots * kStageRow;  // 1116 u64 = 8928 B per wave

template <int KW>
struct MaterializeSink {
    uint64_t *values;
    uint16_t *valid16, *rc16;
    uint64_t *stage;      // this wave's LDS staging area
    uint32_t lane = 0;
    int64_t base = 0;     // global byte index of this lane's first base in the current tile
    uint64_t n_bytes = 0;
    uint32_t inval = 0, rcbits = 0;
    bool halo = true;     // halo lanes own no output slot

    __device__ __forceinline__ void begin_tile(int64_t lane_base, uint32_t inval16, bool halo_lane) { base = lane_base; inval = inval16; rcbits = 0; halo = halo_lane; }
    __device__ __forceinline__ void emit(int j, bool, bool take_fwd, uint32_t hi, uint32_t lo)
    {
        if (values && !halo) stage[(lane - kHaloLanes) * kStageRow + j] = KW == 2 ? (((uint64_t)hi << 32) | lo) : (uint64_t)lo;
        rcbits |= (take_fwd ? 0u : 1u) << (15 - j);
    }
    __device__ __forceinline__ void end_tile()
    {
        if (!halo && base < (int64_t)n_bytes) {
            const uint32_t v = ~inval & 0xFFFFu;
            valid16[base >> 4] = (uint16_t)v;
            rc16[base >> 4] = (uint16_t)(rcbits & v);
        }
        if (!values) return;
        // copy the tile's 992 values out: instruction s moves positions [128 s, 128 s + 128), 16 bytes per lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes above are complete and ordered
        const int64_t tile_pos0 = base - (int64_t)lane * 16 + kHaloLanes * 16;  // position of emitting lane 0, slot 0
        const int64_t limit = (int64_t)((n_bytes + 15) & ~(uint64_t)15);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int q = 128 * s + 2 * (int)lane;
            if (q < kTileSlots * 16 && tile_pos0 + q < limit) {
                const ulonglong2 two = *reinterpret_cast<const ulonglong2 *>(&stage[(q >> 4) * kStageRow + (q & 15)]);
                // streaming (non-temporal) store: the 8 B per position written here are not read again by this kernel and
                // would only evict the sequence bytes from L2 (-13 % kernel time at config-2 size)
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2 vv; vv.x = two.x; vv.y = two.y;
                __builtin_nontemporal_store(vv, reinterpret_cast<u64x2 *>(&values[tile_pos0 + q]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next tile overwrites the stage
    }
};

// 64-bit lane mask "byte BSEL of a == byte BSEL of b" in one SDWA compare, written straight to an SGPR pair.
#define NTK_SDWA_EQ(dst, a, b, BSEL) \
    asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:" #BSEL " src1_sel:" #BSEL : "=s"(dst) : "v"(a), "v"(b))

// ---------------------------------------------------------------------------------------------
// the scan kernel (template flags: see lane_tile in ntk_tile.hpp; ACCEPT_U: U/u is a base coding T because the
// records went through normalize).  Work split: every wave streams its own contiguous run of tiles (992 emitting
// bytes each) with the next tile's load in flight while the current one is processed; no byte is fetched from HBM
// twice (the 32 halo bytes of a tile come back from L2).
// ---------------------------------------------------------------------------------------------
template <int KW, bool CANON, bool TIE_RC, bool ACCEPT_U, bool REDUCE, int KFIX = 0, bool QM = false>
__global__ __launch_bounds__(1024) void scan_kernel(ScanArgs a)
{
    using Sink = typename std::conditional<REDUCE, ReduceSink<KW>, MaterializeSink<KW>>::type;
    __shared__ uint32_t s_hist[REDUCE ? kHistBins : 1];
    __shared__ uint64_t s_red[REDUCE ? 16 * 4 : 1];
    extern __shared__ __attribute__((aligned(16))) uint64_t s_stage[];  // materialise mode: kStageWaveU64 u64 per wave (dynamic)

    Sink sink;
    if constexpr (REDUCE) {
        for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
        sink.hist = s_hist;
        sink.bin_shift = a.bin_shift;
    } else {
        sink.values = a.values; sink.valid16 = a.valid16; sink.rc16 = a.rc16; sink.n_bytes = a.n_bytes;
        sink.stage = s_stage + (threadIdx.x >> 6) * kStageWaveU64; sink.lane = threadIdx.x & 63u;
    }

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // Dynamic work distribution.  The CU arbitrates VALU issue between its resident waves by age, so with a static
    // split the oldest waves race ahead and the youngest finish alone on an under-filled SIMD (measured: wave end
    // times spread over 0.25..0.92 ms of a 0.92 ms kernel).  Instead every wave pulls chunks of tiles from a
    // per-shard counter (8 shards ~ one per XCD, <= ~20 pulls/us each) until the shard is empty; tiles carry no
    // state from their predecessor (halo lanes), so any wave can take any chunk.
    // All tile indices inside the loop are 32-bit and relative to the launch (a launch covers <= 2^25 tiles) so that the
    // loop control stays on the scalar unit (there are no 64-bit ordered scalar compares).
    const uint32_t shard = blockIdx.x % a.n_shards;
    const uint32_t launch_tiles = (uint32_t)(a.tile_end - a.tile_begin);
    const uint32_t shard_begin = shard * a.tiles_per_shard;
    uint32_t shard_end = shard_begin + a.tiles_per_shard;
    if (shard_end > launch_tiles) shard_end = launch_tiles;
    uint32_t *ctr = a.work_counters + shard * 16;
    const uint32_t shard_tiles = shard_begin < shard_end ? shard_end - shard_begin : 0u;
    DevXL xl;
    const bool halo_lane = lane < (uint32_t)kHaloLanes;

    uint32_t next = 0;
    if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);
    next = __builtin_amdgcn_readfirstlane(next);
    while (next < shard_tiles) {
        const uint32_t r0 = shard_begin + next;                 // first tile of the chunk, launch-relative
        uint32_t r1 = r0 + a.chunk_tiles;
        if (r1 > shard_end) r1 = shard_end;
        if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);  // in flight while this chunk is processed
        // wave-uniform buffer descriptor starting 32 bytes (the halo) before the chunk; hardware bounds checking
        // returns 0 (a break byte) past the padded end, and for the "negative" halo offsets of tile 0.
        const uint64_t t0 = a.tile_begin + r0;
        const uint64_t run_byte = t0 * kTileStride;
        const uint32_t halo = t0 ? 32u : 0u;
        const uint64_t cbase = (uint64_t)a.seq + run_byte - halo;
        uint64_t rem = ((a.n_bytes + 15) & ~(uint64_t)15) - (run_byte - halo);
        if (rem > 0xFFFFFF00ull) rem = 0xFFFFFF00ull;
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)cbase);
        const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(cbase >> 32));
        const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)rem);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(((uint64_t)bhi << 32) | blo), 0, nrec, 0x00020000);
        // QM builds: the quality bytes travel as a second stream with the same geometry (out of range -> quality 0 on
        // bytes that are breaks already), prefetched one tile ahead like the sequence stream.
        __amdgpu_buffer_rsrc_t rq = rs;
        if constexpr (QM) {
            const uint64_t qbase = (uint64_t)a.qual + run_byte - halo;
            const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)qbase);  // u32 temporaries: the builtin returns int
            const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)(qbase >> 32));
            rq = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)qhi << 32) | qlo), 0, nrec, 0x00020000);
        }

        uint32_t voff = lane * 16u - (32u - halo);  // wraps (out of range -> 0) for the halo lanes of tile 0
        uint64_t tile_byte = run_byte;              // wave-uniform: first emitting byte of the current tile
        u32x4 cur = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        u32x4 curq = cur;
        if constexpr (QM) curq = __builtin_amdgcn_raw_buffer_load_b128(rq, voff, 0, 0);
        for (uint32_t r = r0; r < r1; r++) {
            u32x4 nxt = cur, nxtq = curq;
            if (r + 1 < r1) {  // wave-uniform
                nxt = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + kTileStride, 0, 0);
                if constexpr (QM) nxtq = __builtin_amdgcn_raw_buffer_load_b128(rq, voff + kTileStride, 0, 0);
            }
            const bool tail = r >= a.tail_tile_rel;  // this tile reaches the end of the input
            Raw16 raw{cur.x, cur.y, cur.z, cur.w};
            if constexpr (QM) raw = quality_break16(raw, Raw16{curq.x, curq.y, curq.z, curq.w}, a.q_add, a.q_sel);
            lane_tile<KW, CANON, TIE_RC, ACCEPT_U, KFIX>(a, sink, xl, raw, (int64_t)tile_byte - 32 + lane * 16, halo_lane, tail);
            cur = nxt; curq = nxtq; voff += kTileStride; tile_byte += kTileStride;
        }
        next = __builtin_amdgcn_readfirstlane(next);
    }

    if constexpr (REDUCE) {
        // wave -> block -> per-block partials (plain stores; the fold kernel sums them)
        uint64_t sum = sink.sum, xr = sink.xr, nf, nv;
        uint32_t *ph = a.part_hist + (size_t)blockIdx.x * kHistBins;
        nf = sink.n_fwd; nv = sink.n_valid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o, 64);
            xr ^= __shfl_xor(xr, o, 64);
            nf += __shfl_xor(nf, o, 64);
            nv += __shfl_xor(nv, o, 64);
        }
        if (lane == 0) { s_red[wave * 4 + 0] = nv; s_red[wave * 4 + 1] = nf; s_red[wave * 4 + 2] = sum; s_red[wave * 4 + 3] = xr; }
        __syncthreads();
        for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) ph[i] = s_hist[i];
        if (threadIdx.x == 0) {
            uint64_t tv = 0, tf = 0, ts = 0, tx = 0;
            for (uint32_t w = 0; w < (blockDim.x >> 6); w++) {
                tv += s_red[w * 4 + 0]; tf += s_red[w * 4 + 1]; ts += s_red[w * 4 + 2]; tx ^= s_red[w * 4 + 3];
            }
            uint64_t *ps = a.part_scalars + (size_t)blockIdx.x * 4;
            ps[0] = tv; ps[1] = tf; ps[2] = ts; ps[3] = tx;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// "sv2" reduce kernel (17 <= K <= 32, canonical): see ntk_tile.hpp (lane_tile_sv2) for the scheme.
//   HB     bits of the value's prefix the LDS histogram is indexed by: 12 (16 KiB, the bins of the result) or 14 (64 KiB,
//          four cells per result bin, folded when the block writes its partials; the address is then one AND of the top half)
// ---------------------------------------------------------------------------------------------
template <int K, int HB>
struct DevMasks2 {
    static constexpr bool kLight = Sv2Light<K, HB>::value;
    // Lane masks: the window ending at byte j is emitted where VA[j] & VB[j] is set.  The last AND of the mask algebra is left
    // to the masked region, which forms exec with it (s_and_b64 exec, VA, VB): one scalar op instead of an AND and a move.
    uint64_t VA[16], VB[16];
    uint64_t sum = 0, sum2 = 0;   // per-lane sum of the lo words (of the whole word in the word builds): two accumulators, used alternately
    uint32_t sumh = 0;            // builds that are not kLight (K >= 24): sum of the hi words mod 2^32 (all that 2^32 * sum needs mod 2^64: a full-rate add)
    uint32_t xh = 0, xlo = 0;     // xor of the lo words; not kLight: xor of the hi words
    uint32_t rep = 0;             // word builds, K <= 6: this lane's histogram copy (see emit_word)
    static constexpr int kWordCopies = K <= 6 ? (((1 << HB) >> (2 * (K <= 6 ? K : 0))) < 64 ? ((1 << HB) >> (2 * (K <= 6 ? K : 0))) : 64) : 1;
    uint32_t one = 1;
    uint32_t nf_s = 0;            // forward-strand count of the WAVE (scalar: s_bcnt1 of the compare mask, no LDS op, no VALU op)
    uint32_t nf_bits = 0;         // fused minimizers: per-lane sum of the chosen keys' strand bits
#ifdef NTK_X_MFMASUM
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 macc = {0, 0, 0, 0}, msel = {0, 0, 0, 0};   // kbench experiment: byte sums on the matrix pipe (see emit_canon)
#endif

    template <int KM, class Enc>   // KM: good bases a window needs (K, or K + W - 1 for windowed minimizers)
    __device__ __forceinline__ void compute(const Enc &en, bool tail_tile, int64_t lane_base, uint64_t n_bytes)
    {
        uint64_t B[16];
#ifdef NTK_X_CLEANTILE   // kbench experiment (profiles/r06v): a tile that holds no break at all (long contigs) needs no byte compares and no mask algebra
        {
            const uint32_t dif = (en.ex[0] ^ en.uu[0]) | (en.ex[1] ^ en.uu[1]) | (en.ex[2] ^ en.uu[2]) | (en.ex[3] ^ en.uu[3]);
            if (!tail_tile && __builtin_amdgcn_ballot_w64(dif != 0u) == 0ull) {
#pragma unroll
                for (int i = 0; i < 16; i++) { VA[i] = Sv2Geom<KM>::kKeep; VB[i] = ~0ull; }
                asm volatile("" ::: "memory");
                return;
            }
        }
#endif
#ifdef NTK_ABL_NOSDWA
#pragma unroll
        for (int i = 0; i < 16; i++) B[i] = __builtin_amdgcn_ballot_w64(en.ex[i & 3] != (uint32_t)i);
        if (false) {
#endif
        NTK_SDWA_EQ(B[0], en.ex[0], en.uu[0], BYTE_0);  NTK_SDWA_EQ(B[1], en.ex[0], en.uu[0], BYTE_1);
        NTK_SDWA_EQ(B[2], en.ex[0], en.uu[0], BYTE_2);  NTK_SDWA_EQ(B[3], en.ex[0], en.uu[0], BYTE_3);
        NTK_SDWA_EQ(B[4], en.ex[1], en.uu[1], BYTE_0);  NTK_SDWA_EQ(B[5], en.ex[1], en.uu[1], BYTE_1);
        NTK_SDWA_EQ(B[6], en.ex[1], en.uu[1], BYTE_2);  NTK_SDWA_EQ(B[7], en.ex[1], en.uu[1], BYTE_3);
        NTK_SDWA_EQ(B[8], en.ex[2], en.uu[2], BYTE_0);  NTK_SDWA_EQ(B[9], en.ex[2], en.uu[2], BYTE_1);
        NTK_SDWA_EQ(B[10], en.ex[2], en.uu[2], BYTE_2); NTK_SDWA_EQ(B[11], en.ex[2], en.uu[2], BYTE_3);
        NTK_SDWA_EQ(B[12], en.ex[3], en.uu[3], BYTE_0); NTK_SDWA_EQ(B[13], en.ex[3], en.uu[3], BYTE_1);
        NTK_SDWA_EQ(B[14], en.ex[3], en.uu[3], BYTE_2); NTK_SDWA_EQ(B[15], en.ex[3], en.uu[3], BYTE_3);
#ifdef NTK_ABL_NOSDWA
        }
#endif
        if (tail_tile) {  // wave-uniform: bytes at or beyond n_bytes are breaks (the last 16-B line may carry padding)
#pragma unroll
            for (int i = 0; i < 16; i++) B[i] &= __builtin_amdgcn_ballot_w64(lane_base + i < (int64_t)n_bytes);
        }
#ifdef NTK_ABL_NOMASKALG
#pragma unroll
        for (int i = 0; i < 16; i++) { VA[i] = B[i]; VB[i] = ~0ull; }
#else
        window_masks_ab_any<KM>(B, VA, VB);
#endif
    }

    // unsigned minimum of two keys that are positive doubles (bit 63 clear; not NaN / infinity: the exponent field is never all ones): one v_min_f64
    __device__ __forceinline__ uint64_t min64(uint64_t a, uint64_t b) const
    {
        uint64_t r;
        asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    // (min(a.hi16, b.lo16) : min(a.lo16, b.hi16)): the T words of positions j and j+8 share registers with crossed halves
    __device__ __forceinline__ uint32_t pk_min16_crossed(uint32_t a, uint32_t b) const
    {
        uint32_t r;
        asm("v_pk_min_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    // Byte offset of a histogram cell = the value's top HB bits, * 4.  (LDS atomics at an address that is not 4-byte aligned raise a
    // memory violation on gfx950 - measured - so the low two bits have to be cleared.)  _hi: the prefix sits in bits 31:16 of T,
    // _lo: in bits 15:0 (the second position of a packed minimum).
    __device__ __forceinline__ uint32_t cell_offset_hi(uint32_t T) const
    {
        if (HB == 14) {
            uint32_t off;
            const uint32_t kMask = 0xFFFCu;   // (T >> 16) & 0xFFFC in one SDWA op
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(off) : "s"(kMask), "v"(T));
            return off;
        }
        return (T >> 18) & 0x3FFCu;
    }
    __device__ __forceinline__ uint32_t cell_offset_lo(uint32_t T) const
    {
        return HB == 14 ? (T & 0xFFFCu) : ((T >> 2) & 0x3FFCu);
    }

    // ---- the masked regions ---------------------------------------------------------------------------------------------------
    // One asm block holds the side effects of FOUR positions.  Per position: exec <- VA & VB (the validity of the window ending
    // there); canonical builds: the strand compare - under that exec its mask is already "valid and forward", so the forward
    // count is s_bcnt1 + s_add on the scalar unit - and the select of the lo word; the digests; the histogram atomic last.  exec
    // is restored once per block.  Everything that does not depend on validity (window words, packed minimum, cell offsets) is
    // computed outside under the full exec mask.  Operand roles: ft / rt = T words (first 16 bases) of the forward / reverse-
    // complement value, fl / rl = their lo words, o = histogram cell offset, t = the chosen lo word.
    // Measured against the round-2 region (exec move; atomic; mad; xor; exec AND; own-cell atomic): -6.5 % kernel time at k = 21
    // (profiles/r03a): every exec write stalls the VALU for ~3 cycles and LDS ops are the most expensive instructions of the loop.
#ifdef NTK_ABL_NOEXEC    // ablation: exec is never narrowed (what a tile without breaks could run, were it not for the halo lanes)
#define NTK_R_EXEC(i) ""
#else
#define NTK_R_EXEC(i) "s_and_b64 exec, %[A" #i "], %[B" #i "]\n"
#endif
#define NTK_R_MASKS(i) [A##i] "s"(VA[pos[i]]), [B##i] "s"(VB[pos[i]])
#ifdef NTK_ABL_NOLDS      // ablations of tools/kbench.hip (wrong results, same instruction stream otherwise)
#define NTK_R_HIST(i) ""
#else
#define NTK_R_HIST(i) "ds_add_u32 %[o" #i "], %[one]\n"
#endif
#define NTK_R_CNT_FIRST "s_bcnt1_i32_b64 %[nf], vcc\n"                                   /* the block's first position starts its count */
#define NTK_R_CNT "s_bcnt1_i32_b64 %[cn], vcc\n s_add_u32 %[nf], %[nf], %[cn]\n"
#ifdef NTK_ABL_NODIGEST
#define NTK_R_SUM_0(x) ""
#define NTK_R_SUM_1(x) ""
#define NTK_R_XOR(x) ""
#else
#define NTK_R_SUM_0(x) "v_mad_u64_u32 %[sumA], %[sd], " x ", 1, %[sumA]\n"
#define NTK_R_SUM_1(x) "v_mad_u64_u32 %[sumB], %[sd], " x ", 1, %[sumB]\n"
#define NTK_R_XOR(x) "v_xor_b32 %[xlo], %[xlo], " x "\n"
#endif
#define NTK_R_SUM_2(x) NTK_R_SUM_0(x)
#define NTK_R_SUM_3(x) NTK_R_SUM_1(x)

    // canonical, kLight (17 <= K <= 23 with the 14-bit histogram)
    template <bool TIE_RC_, class S>
    __device__ __forceinline__ void emit_canon(S &, const int (&pos)[4], const uint32_t (&ft)[4], const uint32_t (&rt)[4], const uint32_t (&fl)[4],
                                               const uint32_t (&rl)[4], uint32_t Tm0, uint32_t Tm1)
    {
        static_assert(kLight, "light builds");
        const uint32_t off[4] = {cell_offset_hi(Tm0), cell_offset_hi(Tm1), cell_offset_lo(Tm0), cell_offset_lo(Tm1)};
        uint32_t t0, t1, t2, t3, cn, nf_grp;
        uint64_t sd;
#define NTK_R_POS(i, CMP, CNT)                                              \
        NTK_R_EXEC(i)                                                       \
        CMP " vcc, %[ft" #i "], %[rt" #i "]\n"                              \
        "v_cndmask_b32 %[t" #i "], %[rl" #i "], %[fl" #i "], vcc\n"          \
        NTK_R_SUM_##i("%[t" #i "]")                                         \
        NTK_R_XOR("%[t" #i "]")                                             \
        NTK_R_HIST(i)                                                       \
        CNT
#define NTK_R_IN(i) [o##i] "v"(off[i]), [ft##i] "v"(ft[i]), [rt##i] "v"(rt[i]), [fl##i] "v"(fl[i]), [rl##i] "v"(rl[i]), NTK_R_MASKS(i)
#define NTK_R_BODY(CMP) NTK_R_POS(0, CMP, NTK_R_CNT_FIRST) NTK_R_POS(1, CMP, NTK_R_CNT) NTK_R_POS(2, CMP, NTK_R_CNT) NTK_R_POS(3, CMP, NTK_R_CNT) "s_mov_b64 exec, -1\n"
#define NTK_R_OPS                                                                                                                        \
        : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),           \
          [sd] "=&s"(sd), [nf] "=&s"(nf_grp), [cn] "=&s"(cn)                                                                             \
        : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one)                                                            \
        : "memory", "vcc", "scc"
#if defined(NTK_X_CMPFIRST)
        // round-6 experiment (a), profiles/r06a: the four strand compares leave the region - under the full exec mask, each into its own
        // SGPR pair, scheduled by the compiler among the window words - and the region keeps exec, select, digests, atomic, count
        uint64_t F[4];
#pragma unroll
        for (int i = 0; i < 4; i++) F[i] = __builtin_amdgcn_ballot_w64(TIE_RC_ ? ft[i] < rt[i] : ft[i] <= rt[i]);
#define NTK_X_POS(i, CNT)                                                   \
        NTK_R_EXEC(i)                                                       \
        "s_and_b64 vcc, exec, %[F" #i "]\n"                                 \
        "v_cndmask_b32 %[t" #i "], %[rl" #i "], %[fl" #i "], vcc\n"          \
        NTK_R_SUM_##i("%[t" #i "]")                                         \
        NTK_R_XOR("%[t" #i "]")                                             \
        NTK_R_HIST(i)                                                       \
        CNT
#define NTK_X_IN(i) [o##i] "v"(off[i]), [fl##i] "v"(fl[i]), [rl##i] "v"(rl[i]), [F##i] "s"(F[i]), NTK_R_MASKS(i)
        asm volatile(NTK_X_POS(0, NTK_R_CNT_FIRST) NTK_X_POS(1, NTK_R_CNT) NTK_X_POS(2, NTK_R_CNT) NTK_X_POS(3, NTK_R_CNT) "s_mov_b64 exec, -1\n"
                     : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
                       [sd] "=&s"(sd), [nf] "=&s"(nf_grp), [cn] "=&s"(cn)
                     : NTK_X_IN(0), NTK_X_IN(1), NTK_X_IN(2), NTK_X_IN(3), [one] "v"(one)
                     : "memory", "vcc", "scc");
#undef NTK_X_IN
#undef NTK_X_POS
#elif defined(NTK_X_SELOUT)
        // round-6 experiment (a''), profiles/r06j: compare AND select leave the region, as in the K >= 24 builds (emit_canon_wide) - under the
        // full exec mask, scheduled by the compiler among the window words; the region keeps exec, the two digests, the atomic and the count
        // (s_and of the compare's mask with exec + s_bcnt1 + s_add).  The k = 31 build, which has this shape, idles 41 cycles per tile where
        // the k = 21 build idles 101.
        uint64_t F[4], fm;
        uint32_t ts[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool fwd = TIE_RC_ ? ft[i] < rt[i] : ft[i] <= rt[i];
            F[i] = __builtin_amdgcn_ballot_w64(fwd);
            ts[i] = fwd ? fl[i] : rl[i];
        }
#define NTK_X_POS(i, CNT)                                                   \
        NTK_R_EXEC(i)                                                       \
        NTK_R_SUM_##i("%[t" #i "]")                                         \
        NTK_R_XOR("%[t" #i "]")                                             \
        NTK_R_HIST(i)                                                       \
        "s_and_b64 %[fm], exec, %[F" #i "]\n"                               \
        CNT
#define NTK_X_CNT_FIRST "s_bcnt1_i32_b64 %[nf], %[fm]\n"
#define NTK_X_CNT "s_bcnt1_i32_b64 %[cn], %[fm]\n s_add_u32 %[nf], %[nf], %[cn]\n"
#define NTK_X_IN(i) [o##i] "v"(off[i]), [t##i] "v"(ts[i]), [F##i] "s"(F[i]), NTK_R_MASKS(i)
        asm volatile(NTK_X_POS(0, NTK_X_CNT_FIRST) NTK_X_POS(1, NTK_X_CNT) NTK_X_POS(2, NTK_X_CNT) NTK_X_POS(3, NTK_X_CNT) "s_mov_b64 exec, -1\n"
                     : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [sd] "=&s"(sd), [nf] "=&s"(nf_grp), [cn] "=&s"(cn), [fm] "=&s"(fm)
                     : NTK_X_IN(0), NTK_X_IN(1), NTK_X_IN(2), NTK_X_IN(3), [one] "v"(one)
                     : "memory", "scc");
        (void)t0; (void)t1; (void)t2; (void)t3;
#undef NTK_X_IN
#undef NTK_X_CNT
#undef NTK_X_CNT_FIRST
#undef NTK_X_POS
#elif defined(NTK_X_TWOPHASE)
        // round-6 experiment (a'), profiles/r06a: the chain cut in two - first exec / compare / select / count of all four positions, then
        // exec / digests / atomic of all four: the select's result is not needed for ~10 instructions, at the price of four more exec writes
#define NTK_X_P1(i, CMP, CNT) NTK_R_EXEC(i) CMP " vcc, %[ft" #i "], %[rt" #i "]\n" "v_cndmask_b32 %[t" #i "], %[rl" #i "], %[fl" #i "], vcc\n" CNT
#define NTK_X_P2(i) NTK_R_EXEC(i) NTK_R_SUM_##i("%[t" #i "]") NTK_R_XOR("%[t" #i "]") NTK_R_HIST(i)
#define NTK_X_BODY(CMP) NTK_X_P1(0, CMP, NTK_R_CNT_FIRST) NTK_X_P1(1, CMP, NTK_R_CNT) NTK_X_P1(2, CMP, NTK_R_CNT) NTK_X_P1(3, CMP, NTK_R_CNT) \
                        NTK_X_P2(0) NTK_X_P2(1) NTK_X_P2(2) NTK_X_P2(3) "s_mov_b64 exec, -1\n"
        if constexpr (TIE_RC_) asm volatile(NTK_X_BODY("v_cmp_lt_u32") NTK_R_OPS);
        else asm volatile(NTK_X_BODY("v_cmp_le_u32") NTK_R_OPS);
#undef NTK_X_BODY
#undef NTK_X_P2
#undef NTK_X_P1
#elif defined(NTK_X_MFMASUM)
        // round-6 experiment (b), profiles/r06a: the sum of the lo words leaves the VALU - the chosen words are zero outside the window's
        // lanes (four full-rate moves before the region) and ONE v_mfma_i32_16x16x64_i8 adds the sixteen bytes a lane holds, by significance,
        // against a 0/1 selector.  TIMING PROXY: i8 is signed, so the byte sums are off by 256 x (bytes with the top bit set), which an exact
        // version would have to count with more VALU work (tools/gen_ubench14.py prices that); n_total / n_fwd / xor / histogram stay exact.
        t0 = t1 = t2 = t3 = 0;
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
#define NTK_X_POS(i, CMP, CNT) NTK_R_EXEC(i) CMP " vcc, %[ft" #i "], %[rt" #i "]\n" "v_cndmask_b32 %[t" #i "], %[rl" #i "], %[fl" #i "], vcc\n" NTK_R_XOR("%[t" #i "]") NTK_R_HIST(i) CNT
#define NTK_X_BODY(CMP) NTK_X_POS(0, CMP, NTK_R_CNT_FIRST) NTK_X_POS(1, CMP, NTK_R_CNT) NTK_X_POS(2, CMP, NTK_R_CNT) NTK_X_POS(3, CMP, NTK_R_CNT) "s_mov_b64 exec, -1\n"
#define NTK_X_OPS : [xlo] "+v"(xlo), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [nf] "=&s"(nf_grp), [cn] "=&s"(cn) \
                  : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one) : "memory", "vcc", "scc"
        if constexpr (TIE_RC_) asm volatile(NTK_X_BODY("v_cmp_lt_u32") NTK_X_OPS);
        else asm volatile(NTK_X_BODY("v_cmp_le_u32") NTK_X_OPS);
        (void)sd;
        macc = __builtin_amdgcn_mfma_i32_16x16x64_i8(i32x4{(int)t0, (int)t1, (int)t2, (int)t3}, msel, macc, 0, 0, 0);
#undef NTK_X_OPS
#undef NTK_X_BODY
#undef NTK_X_POS
#else
        if constexpr (TIE_RC_) asm volatile(NTK_R_BODY("v_cmp_lt_u32") NTK_R_OPS);   // byte path: ties report the reverse complement (src/kmer.rs:124-128)
        else asm volatile(NTK_R_BODY("v_cmp_le_u32") NTK_R_OPS);                     // bit path: ties stay forward (src/bitkmer.rs:136-143)
#endif
        nf_s += nf_grp;
#undef NTK_R_OPS
#undef NTK_R_BODY
#undef NTK_R_IN
#undef NTK_R_POS
    }

    // canonical, K >= 24: the hi word of the chosen value does not follow from the histogram cell, so the whole 64-bit value is summed
    // (v_lshl_add_u64 on a register pair).  The pair has to be built by the compiler, i.e. OUTSIDE the asm block: strand compare
    // (its mask F in an SGPR pair), T = min(ft, rt) - the chosen T word: equal T words mean equal values, ntk_tile.hpp -, the lo
    // select, the hi word T >> (64 - 2K) and the cell all run under the full exec mask; the region keeps the side effects, and the
    // forward count is s_and (F with the window's validity, which is exec) + s_bcnt1 + s_add.  (Measured, k = 31, profiles/r03a/wide_ab.txt: the
    // select inside the region with separate sums of the lo and hi words costs 16 more VALU instructions per tile and is 4-8 % slower;
    // these builds are VALU-bound at 235 instructions per tile - scalar count, LDS count and the round-2 region all run within 1 %.)
    // (Round 4 measured the whole position inside the region with vcc selects and a pinned (lo : hi) pair - 4 issue cycles fewer per
    // position by the class costs, 2.5 % SLOWER: one dependent chain under its own exec mask - and the same selects as a snippet outside:
    // 1.4 % slower; profiles/r04b/wide_region_ab.txt.  Neither form is kept in the source.)
    template <bool TIE_RC_, class S>
    __device__ __forceinline__ void emit_canon_wide(S &, const int (&pos)[4], const uint32_t (&ft)[4], const uint32_t (&rt)[4], const uint32_t (&fl)[4],
                                                       const uint32_t (&rl)[4])
    {
        static_assert(!kLight && K >= 17, "wide builds");
        constexpr int SH = 64 - 2 * K;
        uint32_t hi[4], lo[4], off[4], cn, nf_grp;
        uint64_t F[4], val[4], fm;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool fwd = TIE_RC_ ? ft[i] < rt[i] : ft[i] <= rt[i];
            F[i] = __builtin_amdgcn_ballot_w64(fwd);   // the compare's own SGPR pair
            const uint32_t T = ft[i] < rt[i] ? ft[i] : rt[i];
            lo[i] = fwd ? fl[i] : rl[i];
            hi[i] = SH ? T >> SH : T;
            off[i] = cell_offset_hi(T);
            val[i] = ((uint64_t)hi[i] << 32) | lo[i];
        }
#define NTK_R_POS(i, CNT)                                                   \
        NTK_R_EXEC(i)                                                       \
        NTK_R_WSUM_##i("%[v" #i "]")                                        \
        "v_xor_b32 %[xh], %[xh], %[h" #i "]\n"                              \
        NTK_R_XOR("%[l" #i "]")                                             \
        NTK_R_HIST(i)                                                       \
        NTK_R_WFWD(i)                                                       \
        CNT
#define NTK_R_WSUM_0(x) "v_lshl_add_u64 %[sumA], " x ", 0, %[sumA]\n"
#define NTK_R_WSUM_1(x) "v_lshl_add_u64 %[sumB], " x ", 0, %[sumB]\n"
#define NTK_R_WSUM_2(x) NTK_R_WSUM_0(x)
#define NTK_R_WSUM_3(x) NTK_R_WSUM_1(x)
#define NTK_R_WFWD(i) "s_and_b64 %[fm], exec, %[F" #i "]\n"
#define NTK_R_WCNT_FIRST "s_bcnt1_i32_b64 %[nf], %[fm]\n"
#define NTK_R_WCNT "s_bcnt1_i32_b64 %[cn], %[fm]\n s_add_u32 %[nf], %[nf], %[cn]\n"
#define NTK_R_IN(i) [o##i] "v"(off[i]), [v##i] "v"(val[i]), [h##i] "v"(hi[i]), [l##i] "v"(lo[i]), [F##i] "s"(F[i]), NTK_R_MASKS(i)
        asm volatile(NTK_R_POS(0, NTK_R_WCNT_FIRST) NTK_R_POS(1, NTK_R_WCNT) NTK_R_POS(2, NTK_R_WCNT) NTK_R_POS(3, NTK_R_WCNT) "s_mov_b64 exec, -1\n"
                     : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [xh] "+v"(xh), [nf] "=&s"(nf_grp), [cn] "=&s"(cn), [fm] "=&s"(fm)
                     : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one)
                     : "memory", "scc");
        nf_s += nf_grp;
#undef NTK_R_IN
#undef NTK_R_WFWD
#undef NTK_R_WCNT
#undef NTK_R_WCNT_FIRST
#undef NTK_R_WSUM_3
#undef NTK_R_WSUM_2
#undef NTK_R_WSUM_1
#undef NTK_R_WSUM_0
#undef NTK_R_POS
    }

    // Word builds (K <= 16, lane_tile_sv2w), canonical: f / r = the two candidate values, left-aligned; the chosen value is their
    // minimum (formed outside), its top HB bits the cell; the compare inside the region only feeds the forward count.
    // K <= 6: 4^K bins do not fill the cells: the lane's copy number (rep = lane mod kWordCopies) goes on top of the value, so that
    // the lanes of a wave do not queue up on a handful of cells.  The left-aligned word has zeros below the value: one funnel
    // shift yields (rep : value : 00), the 4-byte-aligned offset of cell rep * 4^K + value.
    __device__ __forceinline__ uint32_t word_cell(uint32_t v) const
    {
        if constexpr (K <= 6) return alignbit(rep, v, 30 - 2 * K);
        else return cell_offset_hi(v);
    }
    template <bool TIE_RC_, uint32_t VMASK, class S>   // VMASK != ~0: the candidates carry junk below the value (odd K, lane_tile_sv2w): the chosen word is masked
    __device__ __forceinline__ void emit_word(S &, const int (&pos)[4], const uint32_t (&f)[4], const uint32_t (&r)[4])
    {
        uint32_t v[4], off[4], cn, nf_grp;
        uint64_t sd;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = (f[i] < r[i] ? f[i] : r[i]) & VMASK; off[i] = word_cell(v[i]); }
#define NTK_R_POS(i, CMP, CNT)                                              \
        NTK_R_EXEC(i)                                                       \
        CMP " vcc, %[f" #i "], %[r" #i "]\n"                                \
        NTK_R_SUM_##i("%[v" #i "]")                                         \
        NTK_R_XOR("%[v" #i "]")                                             \
        NTK_R_HIST(i)                                                       \
        CNT
#define NTK_R_IN(i) [o##i] "v"(off[i]), [f##i] "v"(f[i]), [r##i] "v"(r[i]), [v##i] "v"(v[i]), NTK_R_MASKS(i)
#define NTK_R_BODY(CMP) NTK_R_POS(0, CMP, NTK_R_CNT_FIRST) NTK_R_POS(1, CMP, NTK_R_CNT) NTK_R_POS(2, CMP, NTK_R_CNT) NTK_R_POS(3, CMP, NTK_R_CNT) "s_mov_b64 exec, -1\n"
#define NTK_R_OPS                                                                                                                        \
        : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [sd] "=&s"(sd), [nf] "=&s"(nf_grp), [cn] "=&s"(cn)                       \
        : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one)                                                            \
        : "memory", "vcc", "scc"
        if constexpr (TIE_RC_) asm volatile(NTK_R_BODY("v_cmp_lt_u32") NTK_R_OPS);
        else asm volatile(NTK_R_BODY("v_cmp_le_u32") NTK_R_OPS);
        nf_s += nf_grp;
#undef NTK_R_OPS
#undef NTK_R_BODY
#undef NTK_R_IN
#undef NTK_R_POS
    }

    // One word per position, no strand: the forward-only word builds (v = the value, left-aligned), the forward-only kLight builds
    // (lo word; T carries the prefixes of positions j and j + 8 in its halves) and the fused minimizers (lo word of the window's
    // minimizer, fb = its strand bit, summed per lane).
    template <bool WITH_BITS>
    __device__ __forceinline__ void region_plain(const int (&pos)[4], const uint32_t (&off)[4], const uint32_t (&w)[4], const uint32_t (&fb)[4])
    {
        uint64_t sd;
#define NTK_R_POS(i, BITS)                                                  \
        NTK_R_EXEC(i)                                                       \
        NTK_R_SUM_##i("%[w" #i "]")                                         \
        NTK_R_XOR("%[w" #i "]")                                             \
        BITS                                                                \
        NTK_R_HIST(i)
#define NTK_R_IN(i) [o##i] "v"(off[i]), [w##i] "v"(w[i]), [f##i] "v"(fb[i]), NTK_R_MASKS(i)
        if constexpr (WITH_BITS)
            asm volatile(NTK_R_POS(0, "v_add_u32 %[nfb], %[nfb], %[f0]\n") NTK_R_POS(1, "v_add_u32 %[nfb], %[nfb], %[f1]\n")
                         NTK_R_POS(2, "v_add_u32 %[nfb], %[nfb], %[f2]\n") NTK_R_POS(3, "v_add_u32 %[nfb], %[nfb], %[f3]\n") "s_mov_b64 exec, -1\n"
                         : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [nfb] "+v"(nf_bits), [sd] "=&s"(sd)
                         : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one) : "memory", "scc");   // (s_and_b64 writes SCC)
        else
            asm volatile(NTK_R_POS(0, "") NTK_R_POS(1, "") NTK_R_POS(2, "") NTK_R_POS(3, "") "s_mov_b64 exec, -1\n"
                         : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [sd] "=&s"(sd)
                         : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one) : "memory", "scc");   // (s_and_b64 writes SCC)
#undef NTK_R_IN
#undef NTK_R_POS
    }
    template <class S>
    __device__ __forceinline__ void emit_word_fwd(S &, const int (&pos)[4], const uint32_t (&v)[4])
    {
        const uint32_t off[4] = {word_cell(v[0]), word_cell(v[1]), word_cell(v[2]), word_cell(v[3])};
        region_plain<false>(pos, off, v, v);
    }
    template <class S>
    __device__ __forceinline__ void emit_min4(S &, const int (&pos)[4], const uint32_t (&cell4)[4], const uint32_t (&lo)[4], const uint32_t (&fb)[4])
    {
        uint32_t off[4];   // cell4 = the value's top 14 bits, times four
#pragma unroll
        for (int i = 0; i < 4; i++) off[i] = HB == 14 ? cell4[i] : (cell4[i] >> 4) << 2;
        region_plain<true>(pos, off, lo, fb);
    }
    // Forward-only builds, 17 <= K <= 32 (lane_tile_sv2_fwd).  kLight: T[0], T[1] carry their position's prefix in bits 31:16 and
    // that of T[2], T[3] (the same registers) in bits 15:0.
    template <class S>
    __device__ __forceinline__ void emit_fwd(S &, const int (&pos)[4], const uint32_t (&T)[4], const uint32_t (&lo)[4])
    {
        if constexpr (kLight) {
            const uint32_t off[4] = {cell_offset_hi(T[0]), cell_offset_hi(T[1]), cell_offset_lo(T[2]), cell_offset_lo(T[3])};
            region_plain<false>(pos, off, lo, lo);
        } else {
            constexpr int SH = 64 - 2 * K;
            uint32_t hi[4], off[4];
            uint64_t sd;
#pragma unroll
            for (int i = 0; i < 4; i++) { hi[i] = SH ? T[i] >> SH : T[i]; off[i] = cell_offset_hi(T[i]); }
#define NTK_R_POS(i)                                                        \
        NTK_R_EXEC(i)                                                       \
        NTK_R_SUM_##i("%[l" #i "]")                                         \
        NTK_R_XOR("%[l" #i "]")                                             \
        "v_add_u32 %[sumh], %[sumh], %[h" #i "]\n"                           \
        "v_xor_b32 %[xh], %[xh], %[h" #i "]\n"                              \
        NTK_R_HIST(i)
#define NTK_R_IN(i) [o##i] "v"(off[i]), [l##i] "v"(lo[i]), [h##i] "v"(hi[i]), NTK_R_MASKS(i)
            asm volatile(NTK_R_POS(0) NTK_R_POS(1) NTK_R_POS(2) NTK_R_POS(3) "s_mov_b64 exec, -1\n"
                         : [sumA] "+v"(sum), [sumB] "+v"(sum2), [sumh] "+v"(sumh), [xlo] "+v"(xlo), [xh] "+v"(xh), [sd] "=&s"(sd)
                         : NTK_R_IN(0), NTK_R_IN(1), NTK_R_IN(2), NTK_R_IN(3), [one] "v"(one) : "memory", "scc");   // (s_and_b64 writes SCC)
#undef NTK_R_IN
#undef NTK_R_POS
        }
    }
#undef NTK_R_XOR
#undef NTK_R_SUM_3
#undef NTK_R_SUM_2
#undef NTK_R_SUM_1
#undef NTK_R_SUM_0
#undef NTK_R_CNT
#undef NTK_R_CNT_FIRST
#undef NTK_R_HIST
#undef NTK_R_MASKS
#undef NTK_R_EXEC
};

struct NoSink {};

#ifndef NTK_SV2_MINWAVES
#define NTK_SV2_MINWAVES 1
#endif
// W > 0: windowed minimizers fused into the scan (lane_tile_sv2_min) instead of every k-mer
template <int K, bool TIE_RC, bool ACCEPT_U, bool QM = false, int HB = 12, int W = 0, bool FWD = false>
__global__ __launch_bounds__(1024, NTK_SV2_MINWAVES) void scan2_kernel(ScanArgs a)
{
    static_assert(K >= 1 && K <= 32 && (HB == 12 || HB == 14), "sv2 covers 1 <= k <= 32");
    static_assert(W == 0 || Sv2MinFused<K, W>::value, "fused minimizers: see ntk_tile.hpp");
    static_assert(W == 0 || K <= 16 || Sv2Light<K, HB>::value, "fused minimizers sum the lo word only: the cell must cover the bits above it");
    constexpr bool WORD = K <= 16;                        // one-word values (lane_tile_sv2w): digests kept left-aligned
    constexpr bool LIGHT = Sv2Light<K, HB>::value && !WORD;
    constexpr int kCells = 1 << HB;
    using Geo = Sv2Geom<(W ? K + W - 1 : K)>;   // halo lanes / stride: 2 / 992 bytes; fused minimizers whose windows need more than 32 bytes: 3 / 976
    constexpr uint32_t kStride = Geo::kStride, kHaloB = Geo::kHaloBytes;
    // One LDS object, histogram first: the masked regions address the histogram with the cell's byte offset alone, which
    // is only right while the histogram sits at LDS address 0 (checked below; the kernel has no other LDS object).
    struct Lds { uint32_t hist[kCells]; uint64_t red[16 * 6]; };
    __shared__ Lds L;
    uint32_t *const s_hist = L.hist;
    uint64_t *const s_red = L.red;
    if ((uint32_t)(uintptr_t)&L.hist[0] != 0u) __builtin_trap();

#ifdef NTK_V_CLOCKS
    const uint64_t dbg_c0 = clock64(), dbg_w0 = wall_clock64();
    uint32_t dbg_tiles = 0;
#endif
#ifndef NTK_X_FASTSTART
    if (a.zero_acc && blockIdx.x == 0)   // NTK_FLAG_RESET: the accumulators start from zero (see ScanArgs)
        for (uint32_t i = threadIdx.x; i < a.zero_words; i += blockDim.x) a.zero_acc[i] = 0;
    for (int i = threadIdx.x; i < kCells; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();

#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t shard = blockIdx.x % a.n_shards;
    const uint32_t launch_tiles = (uint32_t)(a.tile_end - a.tile_begin);
    const uint32_t shard_begin = shard * a.tiles_per_shard;
    uint32_t shard_end = shard_begin + a.tiles_per_shard;
    if (shard_end > launch_tiles) shard_end = launch_tiles;
    uint32_t *ctr = a.work_counters + shard * 16;
    const uint32_t shard_tiles = shard_begin < shard_end ? shard_end - shard_begin : 0u;
#ifdef NTK_X_FASTSTART
    // round-6 experiment (profiles/r06o): the wave's first pull is in flight while the block zeroes its histogram (128-bit stores)
    uint32_t next = 0;
    if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);
    if (a.zero_acc && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < a.zero_words; i += blockDim.x) a.zero_acc[i] = 0;
    for (int i = threadIdx.x; i < kCells / 4; i += blockDim.x) reinterpret_cast<u32x4 *>(s_hist)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#endif
    // SPEC: the byte path's tie rule on bytes nobody normalised (NTK_PATH_BYTES_CANONICAL with pre < NORMALIZE).  The reference compares RAW
    // bytes (src/kmer.rs:121-128), which is the 2-bit order as long as no base is lower case - what Sequence::normalize reports by returning
    // None on a clean read (src/sequence.rs:57-61).  The build ORs every byte it loads into `lc` (two full-rate ops per tile); a wave that saw
    // bit 5 anywhere raises a.lower_flag and the host has queued canonical_bytes_reduce_kernel behind this launch, which then redoes it.
    constexpr bool SPEC = TIE_RC && !ACCEPT_U && !QM && W == 0 && !FWD;
    uint32_t lc = 0;
    if (SPEC && a.lower_flag_next && blockIdx.x == 0 && threadIdx.x == 0) *a.lower_flag_next = 0;   // the next launch's flag (a ring, see run_scan)
    DevXL xl;
    DevMasks2<K, HB> mp;
    NoSink sink;
    mp.rep = lane & (uint32_t)(DevMasks2<K, HB>::kWordCopies - 1);
#ifdef NTK_X_MFMASUM
    { const int sel = 1 << (8 * (lane & 3)); mp.msel = {sel, sel, sel, sel}; }   // B[k][j] = [k % 4 == j % 4]: column j sums the bytes of significance j % 4
#endif

#ifndef NTK_X_FASTSTART
    uint32_t next = 0;
    if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);
#endif
    next = __builtin_amdgcn_readfirstlane(next);
#ifdef NTK_X_GUIDED   // kbench experiment (profiles/r06q): the pulls shrink over a shard's last stretch (remaining / (2 x the shard's waves), at least 2 tiles)
    uint32_t c_cur = a.chunk_tiles;
    const uint32_t shard_waves2 = 2u * ((gridDim.x + a.n_shards - 1) / a.n_shards) * (blockDim.x >> 6);
#endif
    while (next < shard_tiles) {
        const uint32_t r0 = shard_begin + next;
#ifdef NTK_X_GUIDED
        uint32_t r1 = r0 + c_cur;
        if (r1 > shard_end) r1 = shard_end;
        {
            const uint32_t done = next + c_cur < shard_tiles ? next + c_cur : shard_tiles;
            uint32_t c = (shard_tiles - done) / shard_waves2;
            c = c < 2u ? 2u : (c > a.chunk_tiles ? a.chunk_tiles : c);
            c_cur = c;
            if (lane == 0) next = atomicAdd(ctr, c);
        }
#else
        uint32_t r1 = r0 + a.chunk_tiles;
        if (r1 > shard_end) r1 = shard_end;
        if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);
#endif
        const uint64_t t0 = a.tile_begin + r0;
        const uint64_t run_byte = t0 * kStride;
        const uint32_t halo = t0 ? kHaloB : 0u;
        const uint64_t cbase = (uint64_t)a.seq + run_byte - halo;
        uint64_t rem = ((a.n_bytes + 15) & ~(uint64_t)15) - (run_byte - halo);
        if (rem > 0xFFFFFF00ull) rem = 0xFFFFFF00ull;
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)cbase);
        const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(cbase >> 32));
        const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)rem);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)bhi << 32) | blo), 0, nrec, 0x00020000);
        __amdgpu_buffer_rsrc_t rq = rs;
        if constexpr (QM) {
            const uint64_t qbase = (uint64_t)a.qual + run_byte - halo;
            const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)qbase);
            const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)(qbase >> 32));
            rq = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)qhi << 32) | qlo), 0, nrec, 0x00020000);
        }
        uint32_t voff = lane * 16u - (kHaloB - halo);
        uint64_t tile_byte = run_byte;
#ifdef NTK_ABL_FLOOR
        auto load_tile = [&](uint32_t off) { (void)rs; return u32x4{off, off, off, off}; };
#else
        auto load_tile = [&](uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); };
#endif
        auto load_qual = [&](uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0); };
#ifdef NTK_ABL_FLOOR
        uint32_t fl_code = lane * 0x9E3779B9u + r0, fl_rcode = ~fl_code, fl_step = 0x85EBCA6Bu + lane;
        asm volatile("" : "+v"(fl_step));
#pragma unroll
        for (int i = 0; i < 16; i++) { mp.VA[i] = ~3ull; mp.VB[i] = ~0ull; }
#endif
        auto process = [&](const u32x4 &t, const u32x4 &q, uint32_t r, auto &&after_encode) {
            const bool tail = r >= a.tail_tile_rel;
            Raw16 raw{t.x, t.y, t.z, t.w};
            if constexpr (SPEC) {
                if (tail) lc |= or_of_input_bytes(raw, (int64_t)a.n_bytes - ((int64_t)tile_byte - (int64_t)kHaloB + lane * 16));   // (wave-uniform: the last 16-byte line's padding is nobody's base)
                else lc = bitop3<0xFE>(bitop3<0xFE>(lc, t.x, t.y), t.z, t.w);
            }
            if constexpr (QM) raw = quality_break16(raw, Raw16{q.x, q.y, q.z, q.w}, a.q_add, a.q_sel);
#ifdef NTK_ABL_LOADSONLY
            mp.xlo ^= raw.x ^ raw.y ^ raw.z ^ raw.w; (void)tail;
            after_encode();
#elif defined(NTK_ABL_FLOOR)
            // floor kernel (tools/kbench.hip, profiles/r03*/floor.txt): no load, no encode, no validity - only the window words and
            // the per-position work, on synthetic register-resident stream words (two adds keep them changing from tile to tile)
            (void)tail; (void)raw;
            after_encode();
            fl_code += fl_step; fl_rcode += fl_code;
            lane_tile_sv2<TIE_RC, K>(sink, xl, mp, fl_code, fl_rcode);
#else
            const EncSV2 en = encode16_sv2<ACCEPT_U>(raw);
            mp.template compute<(W ? K + W - 1 : K)>(en, tail, (int64_t)tile_byte - (int64_t)kHaloB + lane * 16, a.n_bytes);
            after_encode();
            if constexpr (W > 0) lane_tile_sv2_min<TIE_RC, K, W>(sink, xl, mp, en.code, en.rcode);
            else if constexpr (WORD) lane_tile_sv2w<TIE_RC, K, FWD>(sink, xl, mp, en.code, en.rcode);
            else if constexpr (FWD) lane_tile_sv2_fwd<K>(sink, xl, mp, en.code);
            else lane_tile_sv2<TIE_RC, K>(sink, xl, mp, en.code, en.rcode);
#endif
            voff += kStride; tile_byte += kStride;
#ifdef NTK_V_CLOCKS
            dbg_tiles++;
#endif
        };
        // the next tile's load is in flight while the current one is processed
        u32x4 ta = load_tile(voff), qa = ta;
        if constexpr (QM) qa = load_qual(voff);
        // One tile per trip.  The next tile is loaded into the SAME registers as soon as the encode and the validity compares have
        // consumed the current one - the rest of the tile's work (most of it) hides the latency, and no rotation moves are needed
        // (a separate next-tile buffer loaded at the top of the trip and moved at its end: +2 v_mov_b64, about 1 % slower at k = 21, 23
        // and 31, profiles/r03d/lateload_ab2.txt; two tiles per trip: 5 - 25 % slower, profiles/r02b, with the tile offset in the scalar
        // operand no gain, profiles/r03a/pp2_ab.txt; the first tile of the next chunk loaded during the last tile of this one: no gain,
        // profiles/r04b).
#ifdef NTK_X_PREFETCH2
        // round-6 experiment (profiles/r06k): TWO tiles in flight per wave (two register sets, the loop unrolled by two) - for the builds that
        // are not held by VALU issue (forward-only: 110 VALU per tile, 27 % of the tile's cycles idle)
        u32x4 tb = ta;
        if (r0 + 1 < r1) tb = load_tile(voff + kStride);
        for (uint32_t r = r0; r < r1; r += 2) {
            process(ta, qa, r, [&] { if (r + 2 < r1) ta = load_tile(voff + 2 * kStride); });
            if (r + 1 < r1) process(tb, qa, r + 1, [&] { if (r + 3 < r1) tb = load_tile(voff + 2 * kStride); });
        }
#else
        for (uint32_t r = r0; r < r1; r++)
            process(ta, qa, r, [&] {
                if (r + 1 < r1) { ta = load_tile(voff + kStride); if constexpr (QM) qa = load_qual(voff + kStride); }
            });
#endif
        next = __builtin_amdgcn_readfirstlane(next);
    }

#ifdef NTK_V_CLOCKS
    if ((threadIdx.x & 63) == 0 && a.values) {  // per-wave census (tools/kbench.hip): start, end of the tile loop, shader cycles | tiles << 40, placement
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        a.values[w * 4 + 0] = dbg_w0; a.values[w * 4 + 1] = wall_clock64();
        a.values[w * 4 + 2] = ((clock64() - dbg_c0) & 0xFFFFFFFFFFull) | ((uint64_t)dbg_tiles << 40); a.values[w * 4 + 3] = ((uint64_t)xcc << 32) | hwid;
    }
#endif
    if constexpr (SPEC)
        if (a.lower_flag && __builtin_amdgcn_ballot_w64((lc & 0x20202020u) != 0u) != 0ull && lane == 0) atomicOr(a.lower_flag, 1u);
    // wave -> block -> per-block partials (plain stores; the fold kernel sums them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm blocks' LDS atomics are not tracked by the compiler
    // sum: the two alternating accumulators of the lo words, plus (K >= 24) the hi words' sum; xor: lo words, and (K >= 24) T words
    uint64_t sum = mp.sum + mp.sum2 + ((uint64_t)mp.sumh << 32);
#ifdef NTK_X_MFMASUM
    if ((lane & 15) < 4) sum += (uint64_t)(int64_t)(mp.macc.x + mp.macc.y + mp.macc.z + mp.macc.w) << (8 * (lane & 3));   // columns 0..3, all row groups
#endif
    uint64_t xr = (WORD || LIGHT) ? (uint64_t)mp.xlo : ((uint64_t)mp.xh << 32) | mp.xlo, nf, nv = 0;
    uint64_t shi = 0, xf = 0;   // LIGHT: high parts of the digests, from the histogram
    uint32_t *ph = a.part_hist + (size_t)blockIdx.x * kHistBins;
    nf = lane == 0 ? mp.nf_s : 0u;          // the wave's forward-strand count lives in a scalar register
    if constexpr (W > 0) nf = mp.nf_bits;   // strand bits of the chosen keys: forward count (TIE_RC) or rc count
    __syncthreads();
    constexpr int HBE = HB;   // bits of the cell index that are value bits
    for (int c = threadIdx.x; c < kHistBins; c += blockDim.x) {
        uint32_t tot = 0;
        if constexpr (K <= 6) {   // word builds up to 6 bases: cell = copy * 4^K + value, bin = value
            if (c < (1 << (2 * K)))
                for (int r = 0; r < DevMasks2<K, HB>::kWordCopies; r++) tot += s_hist[(r << (2 * K)) + c];
        } else
#pragma unroll
        for (int q = 0; q < kCells / kHistBins; q++) {
            const uint32_t f = c * (kCells / kHistBins) + q, h = s_hist[f];
            tot += h;
            if constexpr (LIGHT) {   // cell f >> (HB - HBE) = the value's top HBE bits p: hi word = p >> (32 + HBE - 2K), xor bits 2K-HBE and up = p
                const uint32_t pfx = f >> (HB - HBE);
                shi += (uint64_t)(pfx >> (32 + HBE - 2 * K)) * h;
                xf ^= (h & 1u) ? pfx : 0u;
            }
        }
        ph[c] = tot; nv += tot;
    }
    if constexpr (LIGHT) {
        // the lo words and the histogram cells overlap in bits [2K-HB, 32) of the value: those bits are taken from the cells
        constexpr uint32_t low_mask = 2 * K - HBE >= 32 ? 0xFFFFFFFFu : ((1u << ((2 * K - HBE) & 31)) - 1u);
        sum += shi << 32;
        xr = (xf << (2 * K - HBE)) | (uint64_t)(mp.xlo & low_mask);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        xr ^= __shfl_xor(xr, o, 64);
        nf += __shfl_xor(nf, o, 64);
        nv += __shfl_xor(nv, o, 64);
    }
    if (lane == 0) { s_red[wave * 4 + 0] = nv; s_red[wave * 4 + 1] = nf; s_red[wave * 4 + 2] = sum; s_red[wave * 4 + 3] = xr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tv = 0, tf = 0, ts = 0, tx = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); w++) {
            tv += s_red[w * 4 + 0]; tf += s_red[w * 4 + 1]; ts += s_red[w * 4 + 2]; tx ^= s_red[w * 4 + 3];
        }
        uint64_t *ps = a.part_scalars + (size_t)blockIdx.x * 4;
        if constexpr (W > 0 && !TIE_RC) tf = tv - tf;   // the bits counted the reverse-complement choices
        if constexpr (FWD) tf = tv;                     // forward-only builds keep no strand counter
        if constexpr (WORD && K < 16 && W == 0) { ts >>= 32 - 2 * K; tx >>= 32 - 2 * K; }   // left-aligned digests (exact: < 2^32 words per block; the fused minimizers keep plain values)
        ps[0] = tv; ps[1] = tf; ps[2] = ts; ps[3] = tx;
    }
}

#ifndef NTK_SCAN_TEMPLATES_ONLY   // ntk_scan2.hip instantiates scan2_kernel only; the plain kernels below belong to ntk_api.hip
// ---------------------------------------------------------------------------------------------
// Generic fused windowed minimizers (round 4): ANY (k <= 31, w <= 49) of the canonical paths in one pass, nothing written to HBM -
// what every (k, w) without a register-fused scan2 build ran as materialise + window-min (8.8 ms per 1.5 Gbases, profiles/r04f).
// Semantics as lane_tile_sv2_min / window_min_reduce_kernel (reference sequence::minimizer, src/sequence.rs:139-152, applied to every
// window of w + k - 1 good bases): the window ending at byte e holds the w k-mers ending at e-w+1 .. e; its minimizer is the smallest
// canonical value, the LEFTMOST on ties, reported with that k-mer's strand flag.
//   * k <= 25 (F64): key = bit 62 | value << 11 | tile position << 1 | strand bit, built for both strands straight from the code streams
//     (ntk_tile.hpp minimizer_keys_f64); ONE v_min_f64 is the strand choice and every leftmost minimum after it.
//   * 26 <= k <= 31: the run-time-k tile logic of the round-1 kernel (lane_tile: value, "window of k contains a break" bit, strand per
//     position), key = (value << 1) | strand flag and a minimum that prefers its LEFT operand on ties and ignores the strand bit:
//         take L  <=>  key_L <= (key_R | 1)          (floor(key_L / 2) <= floor(key_R / 2))
//   * sliding minimum over w for any run-time w, positions x = 16 * lane + j of the wave's tile, with FIXED shifts: doubling
//         M_1 = key;  M_2q[x] = min(M_q[x - q], M_q[x])  while 2q <= w;      window[x] = min(M_q[x - (w - q)], M_q[x])
//     (two overlapping windows of q make w: ceil(log2 w) array minima).  A shift by s < 16 reads own registers (j >= s) or the previous
//     lane's (DPP wave_shr:1, s imports of two registers), 16 <= s < 32 the previous lane's / the lane before that; the doubling rounds sit
//     behind wave-uniform branches on w, the last step is a switch over the shift (ScanArgs::min_overlap, 0..17).
//   * window validity: the k-mer invalid bits of the lane and of the three lanes before it, OR-smeared over the w window ends each k-mer
//     is part of (w <= 49 keeps that in 64 bits).
// Tile geometry at run time: ScanArgs::min_halo_lanes non-emitting lanes, stride (64 - that) * 16 bytes.
// ---------------------------------------------------------------------------------------------
// waves per SIMD the register allocation has to allow, per key form: 4 = two 512-thread blocks per CU (128 VGPRs), 3 = one 768-thread block
// per CU (168 VGPRs); each block has its 64 KiB histogram
#ifndef NTK_MINGEN_WAVES_F64
#define NTK_MINGEN_WAVES_F64 4
#endif
#ifndef NTK_MINGEN_WAVES_G
#define NTK_MINGEN_WAVES_G 4
#endif
constexpr int min_gen_waves(bool f64) { return f64 ? NTK_MINGEN_WAVES_F64 : NTK_MINGEN_WAVES_G; }
constexpr int min_gen_threads(bool f64) { return min_gen_waves(f64) == 3 ? 768 : 512; }
// The output stage of the generic fused minimizer kernel: per window the side effects on the accumulators, four positions per asm block under
// the window's validity mask (as the scan2 regions: exec write, v_mad_u64_u32 on the lo word, xor, strand bit, LDS atomic; WIDE: the bits above
// the lo word summed mod 2^32 and xor-ed as well).  Everything that does not depend on validity is computed outside under the full exec mask.
struct DevMinOut {
    uint64_t sum = 0, sum2 = 0;   // lo words, two accumulators used alternately
    uint32_t sumh = 0, xh = 0;    // WIDE: hi words (mod 2^32: all that 2^32 * sum needs mod 2^64), their xor
    uint32_t xlo = 0, nfb = 0, one = 1;
    template <bool WIDE>
    __device__ __forceinline__ void region(const uint64_t *V, const uint32_t (&off)[4], const uint32_t (&lo)[4], const uint32_t (&hi)[4],
                                           const uint32_t (&sb)[4])
    {
        uint64_t sd;
#define NTK_M_POS(i, ACC, W)                                                \
        "s_mov_b64 exec, %[m" #i "]\n"                                      \
        "v_mad_u64_u32 %[" ACC "], %[sd], %[l" #i "], 1, %[" ACC "]\n"       \
        "v_xor_b32 %[xlo], %[xlo], %[l" #i "]\n"                            \
        "v_add_u32 %[nfb], %[nfb], %[f" #i "]\n"                            \
        W                                                                   \
        "ds_add_u32 %[o" #i "], %[one]\n"
#define NTK_M_WIDE(i) "v_add_u32 %[sumh], %[sumh], %[h" #i "]\n v_xor_b32 %[xh], %[xh], %[h" #i "]\n"
#define NTK_M_IN_N(i) [o##i] "v"(off[i]), [l##i] "v"(lo[i]), [f##i] "v"(sb[i]), [m##i] "s"(V[i])
#define NTK_M_IN(i) NTK_M_IN_N(i), [h##i] "v"(hi[i])
        if constexpr (WIDE)
            asm volatile(NTK_M_POS(0, "sumA", NTK_M_WIDE(0)) NTK_M_POS(1, "sumB", NTK_M_WIDE(1)) NTK_M_POS(2, "sumA", NTK_M_WIDE(2)) NTK_M_POS(3, "sumB", NTK_M_WIDE(3))
                         "s_mov_b64 exec, -1\n"
                         : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [nfb] "+v"(nfb), [sumh] "+v"(sumh), [xh] "+v"(xh), [sd] "=&s"(sd)
                         : NTK_M_IN(0), NTK_M_IN(1), NTK_M_IN(2), NTK_M_IN(3), [one] "v"(one) : "memory");
        else
            asm volatile(NTK_M_POS(0, "sumA", "") NTK_M_POS(1, "sumB", "") NTK_M_POS(2, "sumA", "") NTK_M_POS(3, "sumB", "")
                         "s_mov_b64 exec, -1\n"
                         : [sumA] "+v"(sum), [sumB] "+v"(sum2), [xlo] "+v"(xlo), [nfb] "+v"(nfb), [sd] "=&s"(sd)
                         : NTK_M_IN_N(0), NTK_M_IN_N(1), NTK_M_IN_N(2), NTK_M_IN_N(3), [one] "v"(one) : "memory");
#undef NTK_M_IN
#undef NTK_M_IN_N
#undef NTK_M_WIDE
#undef NTK_M_POS
    }
};

// What minimizer_lane hands its windows to (ntk_tile.hpp): begin() takes the lane's 16 validity flags, emit4() turns four of them into lane
// masks, forms the accumulator operands of the four windows and runs their region.  M = where the cell's 14 bits come from (minimizer_scan_kernel).
template <class Key, bool F64, int M>
struct DevMinSink {
    DevMinOut &out; uint32_t &nv_lane; uint32_t cell_sh;
    uint32_t vb = 0;   // the lane's validity flags, next position's on top
    __device__ __forceinline__ void begin(uint32_t invw)
    {
        nv_lane += __popc(~invw & 0xFFFFu);
        vb = ~invw << 16;
    }
    __device__ __forceinline__ void emit4(int, const Key (&g)[4])   // (groups arrive in position order)
    {
        // the four validity masks: one flag is shifted out per position (v_add_co_u32 with the mask's SGPR pair as its carry output); made
        // here, group by group - all 16 at once would hold 32 SGPRs across the sliding minimum and spill
        uint64_t V[4];
#pragma unroll
        for (int i = 0; i < 4; i++) asm("v_add_co_u32 %0, %1, %0, %0" : "+v"(vb), "=s"(V[i]));
        uint32_t lo[4], hi[4], sb[4], off[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            key_fields(g[i], lo[i], hi[i], sb[i]);
            if constexpr (M == 2) off[i] = (hi[i] >> cell_sh) & 0xFFFCu;
            else if constexpr (M == 3) {
                if constexpr (F64) off[i] = ((uint32_t)(g[i].k >> 32) >> cell_sh) & 0xFFFCu;
                else off[i] = 0;
            }
            else if constexpr (M == 1) off[i] = alignbit(hi[i], lo[i], cell_sh) & 0xFFFCu;
            else off[i] = (lo[i] << cell_sh) & 0xFFFCu;
        }
        out.template region<M == 2>(V, off, lo, hi, sb);
    }
};

// (keys, minima and window validity: ntk_tile.hpp, minimizer_lane - shared with the host emulation)
// LDS: a 14-bit histogram (64 KiB, cell = the value's top 14 bits, left-aligned for k < 7) at address 0 - the regions address it with the
// cell's byte offset alone.  Digests: the lo word of every window's minimizer is summed / xor-ed; k <= 16: that is the value; 17 <= k <= 23:
// the bits above follow from the histogram when the block writes out (cell and lo word cover every bit, as in the LIGHT scan2 builds);
// k >= 24: the hi words are accumulated as well (WIDE).
// MODE = where the cell's 14 bits come from: 0: k <= 7, the value shifted up; 1: 8 <= k <= 23, a funnel shift of (hi : lo); 2: k >= 24, the hi
// word alone (these also sum the hi words: WIDE); 3: f64 keys with 19 <= k <= 23, the key's high word alone (min_gen_mode, host side too).
constexpr int min_gen_mode(uint32_t k, bool f64) { return k >= 24 ? 2 : ((f64 && k >= 19) ? 3 : (k >= 8 ? 1 : 0)); }
template <int KW, bool TIE_RC, bool ACCEPT_U, bool QM, bool F64, int MODE>
__global__ __launch_bounds__(min_gen_threads(F64), min_gen_waves(F64)) void minimizer_scan_kernel(ScanArgs a)
{
    constexpr int HB = 14, kCells = 1 << HB;
    struct Lds { uint32_t hist[kCells]; uint64_t red[12 * 4]; };
    __shared__ Lds L;
    uint32_t *const s_hist = L.hist;
    uint64_t *const s_red = L.red;
    if ((uint32_t)(uintptr_t)&L.hist[0] != 0u) __builtin_trap();
    typedef typename MinKey<F64>::type Key;
    if (a.zero_acc && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < a.zero_words; i += blockDim.x) a.zero_acc[i] = 0;
    for (int i = threadIdx.x; i < kCells; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t shard = blockIdx.x % a.n_shards;
    const uint32_t launch_tiles = (uint32_t)(a.tile_end - a.tile_begin);
    const uint32_t shard_begin = shard * a.tiles_per_shard;
    uint32_t shard_end = shard_begin + a.tiles_per_shard;
    if (shard_end > launch_tiles) shard_end = launch_tiles;
    uint32_t *ctr = a.work_counters + shard * 16;
    const uint32_t shard_tiles = shard_begin < shard_end ? shard_end - shard_begin : 0u;
    const uint32_t HL = a.min_halo_lanes, stride = (64u - HL) * 16u, halo_bytes = HL * 16u;
    DevXL xl;
    DevMinOut out;
    uint32_t nv_lane = 0;
    const uint32_t kk = a.k;
    uint32_t cell_sh;   // the shift of the mode, in a VGPR (an SGPR operand would make the full-rate shifts half-rate)
    {
        const uint32_t s_ = MODE == 2 ? 2 * kk - 48 : (MODE == 3 ? 2 * kk - 37 : (MODE == 1 ? 2 * kk - 16 : 16 - 2 * kk));
        asm volatile("v_mov_b32 %0, %1" : "=v"(cell_sh) : "s"(s_));
    }

    uint32_t next = 0;
    if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);
    next = __builtin_amdgcn_readfirstlane(next);
    while (next < shard_tiles) {
        const uint32_t r0 = shard_begin + next;
        uint32_t r1 = r0 + a.chunk_tiles;
        if (r1 > shard_end) r1 = shard_end;
        if (lane == 0) next = atomicAdd(ctr, a.chunk_tiles);
        const uint64_t t0 = a.tile_begin + r0;
        const uint64_t run_byte = t0 * stride;
        const uint32_t halo = t0 ? halo_bytes : 0u;
        const uint64_t cbase = (uint64_t)a.seq + run_byte - halo;
        uint64_t rem = ((a.n_bytes + 15) & ~(uint64_t)15) - (run_byte - halo);
        if (rem > 0xFFFFFF00ull) rem = 0xFFFFFF00ull;
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)cbase);
        const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(cbase >> 32));
        const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)rem);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)bhi << 32) | blo), 0, nrec, 0x00020000);
        __amdgpu_buffer_rsrc_t rq = rs;
        if constexpr (QM) {
            const uint64_t qbase = (uint64_t)a.qual + run_byte - halo;
            const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)qbase);
            const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)(qbase >> 32));
            rq = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)qhi << 32) | qlo), 0, nrec, 0x00020000);
        }
        uint32_t voff = lane * 16u - (halo_bytes - halo);   // wraps (out of range -> 0) for the halo lanes of tile 0
        uint64_t tile_byte = run_byte;                      // first emitting byte of the current tile
        u32x4 cur = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0), curq = cur;
        if constexpr (QM) curq = __builtin_amdgcn_raw_buffer_load_b128(rq, voff, 0, 0);
        for (uint32_t r = r0; r < r1; r++) {
            u32x4 nxt = cur, nxtq = curq;
            if (r + 1 < r1) {
                nxt = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + stride, 0, 0);
                if constexpr (QM) nxtq = __builtin_amdgcn_raw_buffer_load_b128(rq, voff + stride, 0, 0);
            }
            const bool tail = r >= a.tail_tile_rel;
            Raw16 raw{cur.x, cur.y, cur.z, cur.w};
            if constexpr (QM) raw = quality_break16(raw, Raw16{curq.x, curq.y, curq.z, curq.w}, a.q_add, a.q_sel);
            // keys of the 16 own k-mers, window validity and the window minima (ntk_tile.hpp); the windows arrive four at a time
            // keys of the 16 own k-mers, window validity and the window minima (ntk_tile.hpp); the windows arrive four at a time
            DevMinSink<Key, F64, MODE> sink{out, nv_lane, cell_sh};
            minimizer_lane<KW, TIE_RC, ACCEPT_U, F64>(a, xl, sink, raw, (int64_t)tile_byte - halo_bytes + lane * 16, lane, tail);
            cur = nxt; curq = nxtq; voff += stride; tile_byte += stride;
        }
        next = __builtin_amdgcn_readfirstlane(next);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm blocks' LDS atomics are not tracked by the compiler
    const bool wide = kk >= 24, light = kk >= 17 && kk <= 23;
    uint64_t sum = out.sum + out.sum2, xr = out.xlo;
    if (wide) {
        uint32_t sumh = out.sumh, xh = out.xh;
        if constexpr (F64) { sumh -= nv_lane << 19; xh ^= (nv_lane & 1u) << 19; }   // the keys' marker bit rode along in every hi word
        sum += (uint64_t)sumh << 32;
        xr |= (uint64_t)xh << 32;
    }
    // the strand bits counted: "reverse complement" - except for the f64 keys under TIE_RC, where the tie-winning strand (rc) carries 0
    uint64_t nf = (F64 && TIE_RC) ? out.nfb : nv_lane - out.nfb, nv = nv_lane;
    __syncthreads();
    // cells -> the result's bins (the leading min(k, 6) bases): four cells per bin for k >= 7, 4^(7 - k) for k <= 6
    uint32_t *ph = a.part_hist + (size_t)blockIdx.x * kHistBins;
    const uint32_t cshift = kk >= 7 ? 2u : 14u - 2u * kk, nbins = kk >= 6 ? (uint32_t)kHistBins : 1u << (2u * kk);
    uint64_t shi = 0, xf = 0;
    for (uint32_t c = threadIdx.x; c < (uint32_t)kHistBins; c += blockDim.x) {
        uint32_t tot = 0;
        if (c < nbins)
            for (uint32_t q = 0; q < (1u << cshift); q++) {
                const uint32_t f = (c << cshift) + q, h = s_hist[f];
                tot += h;
                if (light) { shi += (uint64_t)(f >> (46u - 2u * kk)) * h; xf ^= (h & 1u) ? f : 0u; }
            }
        ph[c] = tot;
    }
    if (light) {
        const uint32_t low_bits = 2u * kk - 14u;   // 20 .. 32: the lo word below the cell's bits
        sum += shi << 32;
        xr = (xf << low_bits) | (low_bits >= 32u ? xr : (xr & ((1ull << low_bits) - 1ull)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64); xr ^= __shfl_xor(xr, o, 64); nf += __shfl_xor(nf, o, 64); nv += __shfl_xor(nv, o, 64);
    }
    if (lane == 0) { s_red[wave * 4 + 0] = nv; s_red[wave * 4 + 1] = nf; s_red[wave * 4 + 2] = sum; s_red[wave * 4 + 3] = xr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tv = 0, tf = 0, ts = 0, tx = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); w++) { tv += s_red[w * 4 + 0]; tf += s_red[w * 4 + 1]; ts += s_red[w * 4 + 2]; tx ^= s_red[w * 4 + 3]; }
        uint64_t *ps = a.part_scalars + (size_t)blockIdx.x * 4;
        ps[0] = tv; ps[1] = tf; ps[2] = ts; ps[3] = tx;
    }
}

// Sums the per-block partials into the ctx accumulators (same stream, after the scan kernel).
// Grid: kFoldBinGroups x kFoldRowGroups blocks sum disjoint (bin range, row subset) pieces and add them with one
// u64 atomic per bin; one extra block reduces the scalar partials.  (A single pass over <= 8 MiB, a few microseconds.)
constexpr int kFoldThreads = 256, kFoldBinGroups = kHistBins / kFoldThreads, kFoldRowGroups = 32;
constexpr int kFoldBlocks = kFoldBinGroups * kFoldRowGroups + 1;
// select != nullptr (speculative scans of un-normalised byte-path input, scan2_kernel SPEC): a set flag means the raw-byte kernel re-did
// the launch and left nblocks_alt rows of partials in place of the scan's.  undigested: the partials come from a k > 32 scan - its k-mers
// are counted in acc[NTK_ACC_UNDIGESTED] as well (they are in the counters and the histogram, not in sum / xor).
__global__ __launch_bounds__(kFoldThreads) void fold_kernel(const uint32_t *part_hist, const uint64_t *part_scalars, int nblocks,
                                                            uint64_t *acc, uint32_t *work_counters = nullptr, int n_counters = 8,
                                                            const uint32_t *select = nullptr, int nblocks_alt = 0, int undigested = 0)
{
    if (select && *select) nblocks = nblocks_alt;
    if (blockIdx.x < kFoldBinGroups * kFoldRowGroups) {
        const int bin = (blockIdx.x % kFoldBinGroups) * kFoldThreads + threadIdx.x;
        uint64_t s = 0;
#pragma unroll 8
        for (int b = blockIdx.x / kFoldBinGroups; b < nblocks; b += kFoldRowGroups) s += part_hist[(size_t)b * kHistBins + bin];
        if (s) atomicAdd((unsigned long long *)&acc[8 + bin], (unsigned long long)s);
        return;
    }
    __shared__ uint64_t s_red[kFoldThreads / 64][4];
    // the scan that filled these partials is complete: re-arm its work counters for the next scan (saves a memset launch)
    if (work_counters)
        for (int i = threadIdx.x; i < n_counters; i += kFoldThreads) work_counters[i * 16] = 0;
    uint64_t tv = 0, tf = 0, ts = 0, tx = 0;
    for (int b = threadIdx.x; b < nblocks; b += kFoldThreads) {
        tv += part_scalars[b * 4 + 0]; tf += part_scalars[b * 4 + 1];
        ts += part_scalars[b * 4 + 2]; tx ^= part_scalars[b * 4 + 3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tv += __shfl_xor(tv, o, 64); tf += __shfl_xor(tf, o, 64); ts += __shfl_xor(ts, o, 64); tx ^= __shfl_xor(tx, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = tv; s_red[threadIdx.x >> 6][1] = tf; s_red[threadIdx.x >> 6][2] = ts; s_red[threadIdx.x >> 6][3] = tx; }
    __syncthreads();
    if (threadIdx.x < 64) {
        tv = tf = ts = tx = 0;
        for (int w = 0; w < kFoldThreads / 64; w++) { tv += s_red[w][0]; tf += s_red[w][1]; ts += s_red[w][2]; tx ^= s_red[w][3]; }
        if (threadIdx.x == 0) {
            acc[0] += tv; acc[1] += tf; acc[2] += tv - tf; acc[3] += ts; acc[4] ^= tx;
            if (undigested) acc[5] += tv;
            if (select && *select) acc[6] += 1;   // NTK_ACC_REDONE: this launch's result is the byte-walking kernel's
        }
        acc[8 + kHistBins + threadIdx.x] += (tx >> threadIdx.x) & 1;  // summable form of the xor (one bit counter per lane)
    }
}

// ---------------------------------------------------------------------------------------------
// compat-face kernels (per-sequence API parity; not the throughput path)
// ---------------------------------------------------------------------------------------------

// Exclusive prefix sum of one value per thread across a block (wave scan by shuffles, wave totals through LDS).
// s_wave: THREADS/64 entries of LDS; *total (optional, every thread gets it) = the block's sum.
template <int THREADS, class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *s_wave, T *total = nullptr)
{
    T x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T y = __shfl_up(x, o, 64);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    __syncthreads();   // s_wave may still be read by the previous call's stragglers
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    T wave_off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        const T t = s_wave[w];
        if (w < (int)(threadIdx.x >> 6)) wave_off += t;
        all += t;
    }
    if (total) *total = all;
    return wave_off + x - v;
}

// byte LUT map (normalize / complement): out[i] = lut[in[i]] & 0xFF
__global__ void map_reverse_kernel(const uint8_t *in, uint8_t *out, uint64_t n, const uint16_t *lut)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint8_t)lut[in[n - 1 - i]];
}

// Fixed-stride records, reverse-complemented record by record; bytes outside records are copied.
__global__ void revcomp_records_kernel(const uint8_t *in, uint8_t *out, uint64_t n_records, uint32_t len,
                                       uint32_t stride, const uint16_t *lut)
{
    const uint64_t total = n_records * stride;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = g / stride;
        const uint32_t j = (uint32_t)(g - r * stride);
        out[g] = j < len ? (uint8_t)lut[in[r * stride + (len - 1 - j)]] : in[g];
    }
}

// Stream compaction with a byte map: lut low byte = mapped char, bit 8 = "changed", bit 9 = "deleted".
constexpr int kCompactThreads = 256, kCompactPerThread = 16, kCompactBlockBytes = kCompactThreads * kCompactPerThread;
__global__ __launch_bounds__(kCompactThreads) void compact_count_kernel(const uint8_t *in, uint64_t n, const uint16_t *lut,
                                                                        uint32_t *block_kept, uint32_t *flags)
{
    __shared__ uint32_t s_cnt[kCompactThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kCompactBlockBytes + (uint64_t)threadIdx.x * kCompactPerThread;
    uint32_t kept = 0, changed = 0, deleted = 0;
    for (int i = 0; i < kCompactPerThread; i++) {
        if (base + i < n) {
            const uint16_t m = lut[in[base + i]];
            kept += (m & 0x200) ? 0u : 1u;
            changed |= (m >> 8) & 1u;
            deleted |= (m >> 9) & 1u;
        }
    }
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o, 64);
    if (__any(changed) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1u);
    if (__any(deleted) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1u);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kCompactThreads / 64; w++) t += s_cnt[w];
        block_kept[blockIdx.x] = t;
    }
}

// exclusive scan of block_kept -> block_off (u64), single block
__global__ __launch_bounds__(1024) void compact_scan_kernel(const uint32_t *block_kept, uint64_t *block_off, uint32_t nblocks,
                                                            uint64_t *total)
{
    __shared__ uint64_t s_wave[1024 / 64];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    uint64_t t = 0;
    for (uint32_t i = 0; i < per; i++) if (b0 + i < nblocks) t += block_kept[b0 + i];
    uint64_t all = 0;
    uint64_t run = block_exclusive_scan<1024, uint64_t>(t, s_wave, &all);
    if (threadIdx.x == 0) *total = all;
    for (uint32_t i = 0; i < per; i++) if (b0 + i < nblocks) { block_off[b0 + i] = run; run += block_kept[b0 + i]; }
}

__global__ __launch_bounds__(kCompactThreads) void compact_write_kernel(const uint8_t *in, uint64_t n, const uint16_t *lut,
                                                                        const uint64_t *block_off, uint8_t *out)
{
    __shared__ uint32_t s_wave[kCompactThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kCompactBlockBytes + (uint64_t)threadIdx.x * kCompactPerThread;
    uint16_t m[kCompactPerThread];
    uint32_t kept = 0;
#pragma unroll
    for (int i = 0; i < kCompactPerThread; i++) {
        m[i] = base + i < n ? lut[in[base + i]] : (uint16_t)0x200;
        kept += (m[i] & 0x200) ? 0u : 1u;
    }
    uint64_t w = block_off[blockIdx.x] + block_exclusive_scan<kCompactThreads, uint32_t>(kept, s_wave);
#pragma unroll
    for (int i = 0; i < kCompactPerThread; i++)
        if (!(m[i] & 0x200)) out[w++] = (uint8_t)m[i];
}

// Record starts as one bit per byte position (LSB-first u32 words, zeroed before): one thread per record.  The records of a batch
// then need no break byte between them (ntk_canonical_kmers_batch_planes uploads the caller's bytes as they lie).
__global__ void mark_record_starts_kernel(const uint64_t *offsets, uint64_t n_records, uint64_t n, uint32_t *startbits)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = offsets[r] - offsets[0];
        if (p < n) atomicOr(&startbits[p >> 5], 1u << (p & 31));
    }
}
// any record start among the byte positions [a, b)?  (b - a <= 255)
__device__ __forceinline__ bool any_record_start(const uint32_t *startbits, uint64_t a, uint64_t b)
{
    while (a < b) {
        const uint32_t off = (uint32_t)(a & 31), room = 32u - off, left = (uint32_t)(b - a), take = left < room ? left : room;
        const uint32_t mask = (take == 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << off;
        if (startbits[a >> 5] & mask) return true;
        a += take;
    }
    return false;
}

// CanonicalKmers with the reference's raw-byte comparison (reference src/kmer.rs:84-129), any k <= 255.
// One thread per window start; flags8[p]: bit0 = emitted, bit1 = is_rc.  cls: 1 = good base; comp LUT.
// startbits (optional): a window must not reach across a record start (no break bytes in the buffer then).
__global__ void canonical_bytes_kernel(const uint8_t *seq, uint64_t n, uint32_t k, const uint16_t *comp_lut, uint8_t *flags8,
                                       const uint32_t *startbits = nullptr)
{
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    uint8_t f = 0;
    if (p + k <= n && !(startbits && k > 1 && any_record_start(startbits, p + 1, p + k))) {
        bool good = true;
        for (uint32_t i = 0; i < k; i++) {
            const uint8_t c = seq[p + i] & 0xDF;
            good = good && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        }
        if (good) {
            bool is_rc = true;  // equal slices -> rc (reference src/kmer.rs:124-128)
            for (uint32_t i = 0; i < k; i++) {
                const uint8_t a = seq[p + i], b = (uint8_t)comp_lut[seq[p + k - 1 - i]];
                if (a != b) { is_rc = !(a < b); break; }
            }
            f = (uint8_t)(1u | (is_rc ? 2u : 0u));
        }
    }
    flags8[p] = f;
  }
}

// The same decision for a whole chunk of records lying back to back WITHOUT break bytes, written as the two bit planes of
// ntk_canonical_kmers_batch_planes (bit 15 - p % 16 of word p / 16, p = window START): a block stages 2048 + k - 1 bytes and the
// record-start bits in LDS once; a thread takes 8 consecutive starts and walks their 8 + k - 1 bytes ONCE, keeping the length of the
// run of bases that ends at each byte (a record start resets it to 1): the window that ends there is emitted iff the run is >= k.
// (The one-thread-per-start kernel above re-reads k bytes per start from global memory: 40 Gbases/s; this one: see DESIGN.md.)
constexpr int kPlThreads = 256, kPlPer = 8, kPlTile = kPlThreads * kPlPer;
__global__ __launch_bounds__(kPlThreads) void canonical_bytes_planes_kernel(const uint8_t *seq, uint64_t n, uint64_t n_readable, uint32_t k,
                                                                            const uint16_t *comp_lut, const uint32_t *startbits, uint64_t sb_words,
                                                                            uint16_t *valid16, uint16_t *rc16, unsigned long long *total)
{
    __shared__ __align__(16) uint8_t s_b[kPlTile + 256 + 16];
    __shared__ uint32_t s_sb[(kPlTile + 256) / 32 + 2];
    __shared__ uint8_t s_v[kPlThreads], s_r[kPlThreads], s_comp[256];
    __shared__ uint32_t s_cnt[kPlThreads / 64];
    s_comp[threadIdx.x] = (uint8_t)comp_lut[threadIdx.x];
    const uint64_t n_tiles = (n + kPlTile - 1) / kPlTile, n_words = (n + 15) >> 4;
    const uint32_t need = kPlTile + k - 1;
    uint32_t mine = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * kPlTile;
        __syncthreads();   // the previous tile's readers are done (and s_comp is written)
        for (uint32_t v = threadIdx.x; v * 16 < need; v += kPlThreads) {
            const uint64_t p = t0 + (uint64_t)v * 16;
            u32x4 x = {0u, 0u, 0u, 0u};
            if (p + 16 <= n_readable) x = *reinterpret_cast<const u32x4 *>(seq + p);
            *reinterpret_cast<u32x4 *>(&s_b[v * 16]) = x;
        }
        for (uint32_t w = threadIdx.x; w < need / 32 + 2; w += kPlThreads) {
            const uint64_t gw = (t0 >> 5) + w;
            s_sb[w] = gw < sb_words ? startbits[gw] : 0u;
        }
        __syncthreads();
        const uint32_t s = threadIdx.x * kPlPer;
        uint32_t run = 0, vmask = 0, rmask = 0;
        for (uint32_t i = 0; i < kPlPer + k - 1; i++) {
            const uint32_t idx = s + i;
            const uint8_t c = s_b[idx], cu = c & 0xDF;
            const bool good = t0 + idx < n && (cu == 'A' || cu == 'C' || cu == 'G' || cu == 'T');
            const bool start = (s_sb[idx >> 5] >> (idx & 31)) & 1u;
            run = good ? (start ? 1u : run + 1u) : 0u;
            if (i + 1 >= k && run >= k) {
                const uint32_t j = i + 1 - k, a0 = s + j;   // the window [a0, a0 + k) is emitted; its strand: raw-byte compare, ties -> rc
                bool is_rc = true;
                for (uint32_t m = 0; m < k; m++) {
                    const uint8_t a = s_b[a0 + m], b = s_comp[s_b[a0 + k - 1 - m]];
                    if (a != b) { is_rc = !(a < b); break; }
                }
                vmask |= 0x80u >> j;
                if (is_rc) rmask |= 0x80u >> j;
            }
        }
        s_v[threadIdx.x] = (uint8_t)vmask; s_r[threadIdx.x] = (uint8_t)rmask;
        mine += __popc(vmask);
        __syncthreads();
        if (threadIdx.x < kPlThreads / 2) {
            const uint64_t w = (t0 >> 4) + threadIdx.x;
            if (w < n_words) {
                valid16[w] = (uint16_t)(((uint32_t)s_v[2 * threadIdx.x] << 8) | s_v[2 * threadIdx.x + 1]);
                rc16[w] = (uint16_t)(((uint32_t)s_r[2 * threadIdx.x] << 8) | s_r[2 * threadIdx.x + 1]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kPlThreads / 64; w++) t += s_cnt[w];
        if (t) atomicAdd(total, t);
    }
}

// CanonicalKmers on bytes the caller did NOT normalise, folded into the accumulators (ntk_reduce_device with NTK_PATH_BYTES_CANONICAL and
// pre = NONE / STRIP_RETURNS): the reference compares the RAW bytes of the window with the raw bytes of the reverse complement's window
// (src/kmer.rs:121-128) and lower case sorts above upper case, so on mixed-case input the strand is not the smaller 2-bit value and the
// packed-value scan cannot serve it.  Records lie back to back with a break byte each (the batch layout: any byte that is not acgtACGT ends a
// run).  Same staging as the plane kernels; a thread walks the 8 + k - 1 bytes of its 8 starts once with the run length of bases and the
// rolling 2-bit values of both strands; where a window is emitted the strand is the byte-wise compare (ties -> rc), the value the chosen
// strand's.  Per-block partials in the layout fold_kernel sums.  k <= 32 (the digests are defined on 2-bit values).
// run_if != nullptr: the kernel is the second half of a speculative launch pair and returns at once unless the scan before it found a byte
// with bit 5 set (lower case: the only input on which the raw-byte order and the 2-bit order differ).
// k > 32 (any k <= 255, reference src/kmer.rs:48-82: k is a u8): the items' positions and strands are as well defined as for small k, their
// 2-bit values are not (more than 64 bits) - counters and the histogram of the leading 6 bases are produced, sum / xor stay 0 and fold_kernel
// counts the k-mers as undigested.  normalized: the batch is to be read as Sequence::normalize would have left it (U / u are T, case
// folded: src/sequence.rs:24-51), i.e. the strand compare runs on the 2-bit codes; only k > 32 comes here in that form.
// (WIDE = false: the k <= 32 raw-byte build, whose inner loop carries none of the k > 32 / normalised cases - they cost it 30 %, profiles/r06b.)
template <bool WIDE>
__global__ __launch_bounds__(kPlThreads) void canonical_bytes_reduce_kernel(const uint8_t *seq, uint64_t n, uint64_t n_readable, uint32_t k, uint32_t bin_shift,
                                                                            const uint16_t *comp_lut, uint32_t *part_hist, uint64_t *part_scalars,
                                                                            const uint32_t *run_if, uint32_t normalized_arg)
{
    if (run_if && *run_if == 0u) return;
    const bool normalized = WIDE && normalized_arg;
    // a thread walks PER starts + k - 1 bytes: 32 starts per thread (the plane kernels' 8 would be 3.5 byte steps per start at k = 21 and 9 at
    // k = 64; 32: 1.6 and 3 - round 6: 5.0 -> 4.2 ms per 1.51 GB at k = 21 on mixed-case input, 13.6 -> 9.5 ms at k = 64, profiles/r06l)
    constexpr uint32_t PER = 32u, TILE = (uint32_t)kPlThreads * PER;
    __shared__ __align__(16) uint8_t s_b[TILE + 256 + 16];
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint8_t s_comp[256];
    __shared__ uint64_t s_red[kPlThreads / 64][4];
    s_comp[threadIdx.x] = (uint8_t)comp_lut[threadIdx.x];
    for (int i = threadIdx.x; i < kHistBins; i += kPlThreads) s_hist[i] = 0;
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    const uint32_t need = TILE + k - 1;
    constexpr bool wide = WIDE;
    const uint64_t vmask_k = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t top = wide ? 0u : 2 * k - 2;
    uint64_t nv = 0, nf = 0, sum = 0, xr = 0;
    auto code_of = [](uint8_t c) -> uint32_t { const uint32_t x = (c >> 1) & 3u; return x ^ (x >> 1); };   // A0 C1 G2 T3 (U3), either case
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * TILE;
        __syncthreads();   // the previous tile's readers are done (and s_comp / s_hist are set up)
        for (uint32_t v = threadIdx.x; v * 16 < need; v += kPlThreads) {
            const uint64_t p = t0 + (uint64_t)v * 16;
            u32x4 x = {0u, 0u, 0u, 0u};
            if (p + 16 <= n_readable) x = *reinterpret_cast<const u32x4 *>(seq + p);
            *reinterpret_cast<u32x4 *>(&s_b[v * 16]) = x;
        }
        __syncthreads();
        const uint32_t s = threadIdx.x * PER;
        uint32_t run = 0;
        uint64_t fwd = 0, rc = 0;
        for (uint32_t i = 0; i < PER + k - 1; i++) {
            const uint32_t idx = s + i;
            const uint8_t c = s_b[idx], cu = c & 0xDF;
            const bool good = t0 + idx < n && (cu == 'A' || cu == 'C' || cu == 'G' || cu == 'T' || (normalized && cu == 'U'));
            const uint32_t code = code_of(c);
            run = good ? run + 1u : 0u;
            fwd = ((fwd << 2) | code) & vmask_k;
            rc = (rc >> 2) | ((uint64_t)(3u - code) << top);
            if (i + 1 >= k && run >= k) {
                const uint32_t a0 = idx + 1 - k;
                bool is_rc = true;   // equal slices -> rc (src/kmer.rs:124-128)
                if (normalized) {
                    for (uint32_t m = 0; m < k; m++) {
                        const uint32_t a = code_of(s_b[a0 + m]), b = 3u - code_of(s_b[a0 + k - 1 - m]);
                        if (a != b) { is_rc = !(a < b); break; }
                    }
                } else {
                    for (uint32_t m = 0; m < k; m++) {
                        const uint8_t a = s_b[a0 + m], b = s_comp[s_b[a0 + k - 1 - m]];
                        if (a != b) { is_rc = !(a < b); break; }
                    }
                }
                nv++; nf += is_rc ? 0u : 1u;
                if (!wide) {
                    const uint64_t v = is_rc ? rc : fwd;
                    sum += v; xr ^= v;
                    atomicAdd(&s_hist[(uint32_t)(v >> bin_shift)], 1u);
                } else {   // the leading six bases of the chosen strand
                    uint32_t bin = 0;
                    for (uint32_t m = 0; m < 6; m++) bin = (bin << 2) | (is_rc ? 3u - code_of(s_b[a0 + k - 1 - m]) : code_of(s_b[a0 + m]));
                    atomicAdd(&s_hist[bin], 1u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t *ph = part_hist + (size_t)blockIdx.x * kHistBins;
    for (int i = threadIdx.x; i < kHistBins; i += kPlThreads) ph[i] = s_hist[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nv += __shfl_xor(nv, o, 64); nf += __shfl_xor(nf, o, 64); sum += __shfl_xor(sum, o, 64); xr ^= __shfl_xor(xr, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = nv; s_red[threadIdx.x >> 6][1] = nf; s_red[threadIdx.x >> 6][2] = sum; s_red[threadIdx.x >> 6][3] = xr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tv = 0, tf = 0, ts = 0, tx = 0;
        for (int w = 0; w < kPlThreads / 64; w++) { tv += s_red[w][0]; tf += s_red[w][1]; ts += s_red[w][2]; tx ^= s_red[w][3]; }
        uint64_t *ps = part_scalars + (size_t)blockIdx.x * 4;
        ps[0] = tv; ps[1] = tf; ps[2] = ts; ps[3] = tx;
    }
}

// ---------------------------------------------------------------------------------------------
// CanonicalKmers with 33 <= k <= 255 (reference src/kmer.rs:48-82: k is a u8) on the reduce face, from the packed 2-bit streams: what
// canonical_bytes_reduce_kernel<true> produces (counters + histogram of the chosen strand's leading six bases; such k-mers have no 64-bit
// value, so no sum / xor) at a twelfth of its time - that kernel walks PER + k - 1 bytes per thread and compares byte by byte.
//   * A block stages 4096 window ENDS + the 256 bytes before them (k - 1 <= 254) as 272 slots of 16 bases: the scan2 encode gives each
//     slot its forward code word, its reverse-complement word and its break mask (ntk_tile.hpp encode16_sv2, bad16_from_letters).
//   * Validity: a window is emitted iff no break lies in its k bytes.  Per slot the position of its last break; an inclusive max-scan
//     over the slots (wave shuffles + one LDS round) hands every thread the last break before its own slot, and with the own mask that
//     is a 16-bit window mask in a dozen scalar-free ops - no per-position lane masks (the window may span sixteen lanes).
//   * Strand: the k-mer's first 32 bases against its reverse complement's first 32 (the complement of its LAST 32, reversed): two
//     funnel shifts each from three realigned forward words (the window starts k - 1 bases back: a wave-uniform word offset and bit
//     offset) and from the own and the two previous reverse-complement words; `<` on the (hi, lo) pair as the reference's slice compare
//     gives it (src/kmer.rs:124-128: ties report the reverse complement).  k >= 33, so 32 bases are a proper prefix; two k-mers that
//     agree on them (4^-32 per position on random text, but every window of a long inverted repeat) raise *redo_flag, and so does a byte
//     with bit 5 set when the input was not normalised (lower case: the raw-byte order is then not the 2-bit order) - the host has queued
//     canonical_bytes_reduce_kernel<true> behind this launch, which then redoes it (run_wide_reduce in ntk_api.hip; the fold takes
//     whichever partials are valid).  ACCEPT_U: the batch is read as Sequence::normalize leaves it (U / u are T, src/sequence.rs:24-51).
// ---------------------------------------------------------------------------------------------
// max of x and the value a DPP pattern brings from another lane (lanes the pattern does not reach keep x; every x here is >= -1)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int32_t wk_dpp_max(int32_t x)
{
    const int32_t y = __builtin_amdgcn_update_dpp(-1, x, CTRL, ROW_MASK, 0xF, false);
    return y > x ? y : x;
}
// inclusive max-scan over the rows of 16 lanes (row_shr:1, 2, 4, 8), and on to the whole wave (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int32_t wk_row_scan_max(int32_t x)
{
    x = wk_dpp_max<0x111, 0xF>(x); x = wk_dpp_max<0x112, 0xF>(x); x = wk_dpp_max<0x114, 0xF>(x); return wk_dpp_max<0x118, 0xF>(x);
}
__device__ __forceinline__ int32_t wk_wave_scan_max(int32_t x)
{
    x = wk_row_scan_max(x);
    x = wk_dpp_max<0x142, 0xA>(x);
    return wk_dpp_max<0x143, 0xC>(x);
}
template <bool ACCEPT_U>
__global__ __launch_bounds__(kWkThreads) void wide_canonical_reduce_kernel(const uint8_t *seq, uint64_t n, uint32_t k, uint32_t *part_hist,
                                                                           uint64_t *part_scalars, uint32_t *redo_flag, uint32_t *redo_flag_next)
{
    __shared__ uint32_t s_code[kWkSlots + 2], s_rcode[kWkSlots];
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ int32_t s_wmax[kWkThreads / 64 + 1];
    __shared__ uint64_t s_red[kWkThreads / 64][2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int i = tid; i < kHistBins; i += kWkThreads) s_hist[i] = 0;
    if (tid < 2) s_code[kWkSlots + tid] = 0;   // read as the far half of a funnel shift whose bits are not used (see R2 below)
    if (redo_flag_next && blockIdx.x == 0 && tid == 0) *redo_flag_next = 0;   // the next launch's flag (a ring, as the scan's lower_flag)
    const uint64_t n_tiles = (n + kWkTile - 1) / kWkTile, n_readable = (n + 15) & ~(uint64_t)15;
    // the window ending at byte 0 of slot s starts k - 1 bases earlier: in code word s - back, base_off bases below its top
    const uint32_t back = wk_back_words(k), base_bits = wk_base_bits(k);
    const int32_t s_own = (int32_t)tid + kWkHaloSlots;
    uint32_t nv = 0, nf = 0, redo = 0, lc = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t t0 = (int64_t)(tile * kWkTile) - 16 * kWkHaloSlots;   // byte of staged slot 0 (negative in the first tile)
        __syncthreads();   // the previous tile's readers are done (first trip: s_hist is zero)
        auto stage = [&](int32_t s, uint32_t &bad) -> int32_t {   // encodes slot s; returns the staged position of its last break (-1: none)
            const int64_t p = t0 + 16 * (int64_t)s;
            u32x4 x = {0u, 0u, 0u, 0u};
            if (p >= 0 && (uint64_t)p + 16 <= n_readable) x = *reinterpret_cast<const u32x4 *>(seq + p);
            const WkSlot sl = wk_stage_slot<ACCEPT_U>(Raw16{x.x, x.y, x.z, x.w}, s, p < 0 ? 0 : (int64_t)n - p);   // bytes before the input or at / beyond n are breaks
            if (!ACCEPT_U) lc |= sl.or_bytes;
            bad = sl.bad;
            s_code[s] = sl.code; s_rcode[s] = sl.rcode;
            return sl.last_break;
        };
        uint32_t bad_own, bad_halo;
        int32_t inc = stage(s_own, bad_own);
        int32_t halo = -1;
        if (tid < kWkHaloSlots) halo = stage((int32_t)tid, bad_halo);
        if (wave == 0) {   // last break of the halo slots (lanes 0..15 of wave 0: the scan of row 0 ends in lane 15)
            halo = wk_row_scan_max(halo);
            if (lane == 15) s_wmax[kWkThreads / 64] = halo;
        }
        inc = wk_wave_scan_max(inc);   // inclusive max-scan over the wave's slots (DPP: the LDS pipe is the histogram's)
        if (lane == 63) s_wmax[wave] = inc;
        int32_t before = __builtin_amdgcn_update_dpp(-1, inc, 0x138, 0xF, 0xF, false);   // wave_shr:1 - the last break before the own slot, within the wave
        __syncthreads();
        {
            const int32_t h = s_wmax[kWkThreads / 64];
            before = h > before ? h : before;
            for (uint32_t w = 0; w < wave; w++) { const int32_t y = s_wmax[w]; before = y > before ? y : before; }
        }
        const uint32_t valid = wk_valid16(bad_own, before, k, s_own);   // position j at bit 15 - j
        const uint32_t i0 = (uint32_t)s_own - back;
        const uint32_t w0 = s_code[i0], w1 = s_code[i0 + 1], w2 = s_code[i0 + 2], w3 = s_code[i0 + 3];
        // three words realigned to the first window's first base (bases a0 .. a0 + 47; the last one is never used, so w3 may be a word nobody staged)
        const WkWords ww = {wk_take32(w0, w1, base_bits), wk_take32(w1, w2, base_bits), wk_take32(w2, w3, base_bits),
                            s_rcode[s_own], s_rcode[s_own - 1], s_rcode[s_own - 2]};
        if (valid) {
            auto position = [&](auto jc) {   // (the strand bits gathered branch-free and counted once per slot: 10 % slower, profiles/r06r)
                constexpr int J = decltype(jc)::value;
                if ((valid >> (15 - J)) & 1u) {
                    bool lt, tie; uint32_t top;
                    wk_strand<J>(ww, lt, tie, top);
                    redo |= tie ? 1u : 0u;
                    atomicAdd(&s_hist[top >> 20], 1u);
                    nv++; nf += lt ? 1u : 0u;
                }
            };
#define NTK_WK_POS(J) position(std::integral_constant<int, J>{});
            NTK_WK_POS(0) NTK_WK_POS(1) NTK_WK_POS(2) NTK_WK_POS(3) NTK_WK_POS(4) NTK_WK_POS(5) NTK_WK_POS(6) NTK_WK_POS(7)
            NTK_WK_POS(8) NTK_WK_POS(9) NTK_WK_POS(10) NTK_WK_POS(11) NTK_WK_POS(12) NTK_WK_POS(13) NTK_WK_POS(14) NTK_WK_POS(15)
#undef NTK_WK_POS
        }
    }
    if (!ACCEPT_U && (lc & 0x20202020u)) redo = 1;
    if (redo_flag && __builtin_amdgcn_ballot_w64(redo != 0u) != 0ull && lane == 0) atomicOr(redo_flag, 1u);
    __syncthreads();
    uint32_t *ph = part_hist + (size_t)blockIdx.x * kHistBins;
    for (int i = tid; i < kHistBins; i += kWkThreads) ph[i] = s_hist[i];
    uint64_t nv64 = nv, nf64 = nf;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nv64 += __shfl_xor(nv64, o, 64); nf64 += __shfl_xor(nf64, o, 64); }
    if (lane == 0) { s_red[wave][0] = nv64; s_red[wave][1] = nf64; }
    __syncthreads();
    if (tid == 0) {
        uint64_t tv = 0, tf = 0;
        for (int w = 0; w < kWkThreads / 64; w++) { tv += s_red[w][0]; tf += s_red[w][1]; }
        uint64_t *ps = part_scalars + (size_t)blockIdx.x * 4;
        ps[0] = tv; ps[1] = tf; ps[2] = 0; ps[3] = 0;
    }
}

// BitNuclKmer (reference src/bitkmer.rs:39-109, Sequence::bit_kmers src/sequence.rs:250-252) in the same bit-plane form, for
// ntk_bit_kmers_batch_planes: per window START "emitted" and "was_rc", plus (optionally) the item's packed value as a dense u64 per window
// start (0 where nothing is emitted).  Same staging as above; a thread walks the 8 + k - 1 bytes of its 8 starts once with the run length of
// bases (acgtACGT only: src/bitkmer.rs:8-15), the rolling forward value and the rolling reverse-complement value (extend_kmer, :26-36, and
// reverse_complement, :112-132, one base at a time); CANON: the smaller of the two, ties keep the forward k-mer (:136-143).
template <bool CANON>
__global__ __launch_bounds__(kPlThreads) void bit_kmers_planes_kernel(const uint8_t *seq, uint64_t n, uint64_t n_readable, uint32_t k, const uint32_t *startbits,
                                                                      uint64_t sb_words, uint16_t *valid16, uint16_t *rc16, uint64_t *values,
                                                                      unsigned long long *total)
{
    __shared__ __align__(16) uint8_t s_b[kPlTile + 256 + 16];
    __shared__ uint32_t s_sb[(kPlTile + 256) / 32 + 2];
    __shared__ uint8_t s_v[kPlThreads], s_r[kPlThreads];
    __shared__ uint32_t s_cnt[kPlThreads / 64];
    const uint64_t n_tiles = (n + kPlTile - 1) / kPlTile, n_words = (n + 15) >> 4;
    const uint32_t need = kPlTile + k - 1;
    const uint64_t vmask_k = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t top = 2 * k - 2;
    uint32_t mine = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * kPlTile;
        __syncthreads();   // the previous tile's readers are done
        for (uint32_t v = threadIdx.x; v * 16 < need; v += kPlThreads) {
            const uint64_t p = t0 + (uint64_t)v * 16;
            u32x4 x = {0u, 0u, 0u, 0u};
            if (p + 16 <= n_readable) x = *reinterpret_cast<const u32x4 *>(seq + p);
            *reinterpret_cast<u32x4 *>(&s_b[v * 16]) = x;
        }
        for (uint32_t w = threadIdx.x; w < need / 32 + 2; w += kPlThreads) {
            const uint64_t gw = (t0 >> 5) + w;
            s_sb[w] = gw < sb_words ? startbits[gw] : 0u;
        }
        __syncthreads();
        const uint32_t s = threadIdx.x * kPlPer;
        uint32_t run = 0, vmask = 0, rmask = 0;
        uint64_t fwd = 0, rc = 0, out[kPlPer];
#pragma unroll
        for (int j = 0; j < kPlPer; j++) out[j] = 0;
        for (uint32_t i = 0; i < kPlPer + k - 1; i++) {
            const uint32_t idx = s + i;
            const uint8_t c = s_b[idx], cu = c & 0xDF;
            const bool good = t0 + idx < n && (cu == 'A' || cu == 'C' || cu == 'G' || cu == 'T');
            const bool start = (s_sb[idx >> 5] >> (idx & 31)) & 1u;
            const uint32_t x = (c >> 1) & 3u, code = x ^ (x >> 1);   // A0 C1 G2 T3
            run = good ? (start ? 1u : run + 1u) : 0u;
            fwd = ((fwd << 2) | code) & vmask_k;
            rc = (rc >> 2) | ((uint64_t)(3u - code) << top);
            if (i + 1 >= k && run >= k) {
                const uint32_t j = i + 1 - k;
                const bool was_rc = CANON && fwd > rc;
                vmask |= 0x80u >> j;
                if (was_rc) rmask |= 0x80u >> j;
#pragma unroll
                for (int q = 0; q < kPlPer; q++) if ((uint32_t)q == j) out[q] = was_rc ? rc : fwd;
            }
        }
        s_v[threadIdx.x] = (uint8_t)vmask; s_r[threadIdx.x] = (uint8_t)rmask;
        mine += __popc(vmask);
        if (values) {
#pragma unroll
            for (int q = 0; q < kPlPer; q++) if (t0 + s + q < ((n + 15) & ~(uint64_t)15)) values[t0 + s + q] = out[q];
        }
        __syncthreads();
        if (threadIdx.x < kPlThreads / 2) {
            const uint64_t w = (t0 >> 4) + threadIdx.x;
            if (w < n_words) {
                valid16[w] = (uint16_t)(((uint32_t)s_v[2 * threadIdx.x] << 8) | s_v[2 * threadIdx.x + 1]);
                rc16[w] = (uint16_t)(((uint32_t)s_r[2 * threadIdx.x] << 8) | s_r[2 * threadIdx.x + 1]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kPlThreads / 64; w++) t += s_cnt[w];
        if (t) atomicAdd(total, t);
    }
}

// ---------------------------------------------------------------------------------------------
// batched compat face: the items of a whole batch of records, compacted on the device
// (Sequence::canonical_kmers / bit_kmers for every record of a FastxReader batch in one call, reference
// src/sequence.rs:237-252).  Records are packed back to back with one break byte each; window flags come as bit planes
// (bit 15 - i%16 of word i/16): indexed by the window's END for the packed-value scan, by its START for the raw-byte kernel.
// ---------------------------------------------------------------------------------------------
// flags8 (canonical_bytes_kernel: bit 0 emitted, bit 1 is_rc, by window start) -> the two bit planes; one thread per word
__global__ void pack_flags8_kernel(const uint8_t *flags8, uint64_t n, uint16_t *valid16, uint16_t *rc16)
{
    const uint64_t n_words = (n + 15) >> 4;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0, r = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint64_t p = w * 16 + i;
            const uint32_t f = p < n ? flags8[p] : 0u;
            v |= (f & 1u) << (15 - i);
            r |= ((f >> 1) & 1u) << (15 - i);
        }
        valid16[w] = (uint16_t)v; rc16[w] = (uint16_t)(r & v);
    }
}

constexpr int kCpThreads = 256, kCpWords = 4, kCpBlockWords = kCpThreads * kCpWords;   // a block covers 16 384 positions
__global__ __launch_bounds__(kCpThreads) void cp_count_kernel(const uint16_t *valid16, uint64_t n_words, uint32_t *block_items)
{
    __shared__ uint32_t s_wave[kCpThreads / 64];
    const uint64_t w0 = (uint64_t)blockIdx.x * kCpBlockWords + (uint64_t)threadIdx.x * kCpWords;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kCpWords; i++) c += w0 + i < n_words ? __popc((uint32_t)valid16[w0 + i]) : 0u;
    uint32_t all = 0;
    (void)block_exclusive_scan<kCpThreads, uint32_t>(c, s_wave, &all);
    if (threadIdx.x == 0) block_items[blockIdx.x] = all;
}

// exclusive scan of the per-block item counts (any number of blocks: tiles of 1024 with a running carry), single block
__global__ __launch_bounds__(1024) void cp_scan_kernel(const uint32_t *block_items, uint64_t *block_off, uint64_t nblocks, uint64_t *total)
{
    __shared__ uint64_t s_wave[1024 / 64];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const uint64_t b = b0 + threadIdx.x;
        const uint64_t v = b < nblocks ? block_items[b] : 0;
        uint64_t all = 0;
        const uint64_t ex = block_exclusive_scan<1024, uint64_t>(v, s_wave, &all);
        if (b < nblocks) block_off[b] = carry + ex;
        carry += all;
    }
    if (threadIdx.x == 0) *total = carry;
}

// rec_start[r] = packed offset of record r's first byte (n_records + 1 entries; record r ends one byte before rec_start[r+1]).
// index_shift: window start = flag index - index_shift (k - 1 for end-indexed planes, 0 for start-indexed ones).
__global__ __launch_bounds__(kCpThreads) void cp_scatter_kernel(const uint16_t *valid16, const uint16_t *rc16, const uint64_t *values,
                                                                uint64_t n_words, const uint64_t *block_off, const uint64_t *rec_start,
                                                                uint64_t n_records, uint32_t index_shift, uint64_t cap,
                                                                uint64_t *pos_out, uint64_t *val_out, uint8_t *flag_out,
                                                                unsigned long long *counts)
{
    __shared__ uint32_t s_wave[kCpThreads / 64];
    const uint64_t w0 = (uint64_t)blockIdx.x * kCpBlockWords + (uint64_t)threadIdx.x * kCpWords;
    uint32_t vw[kCpWords], rw[kCpWords], c = 0;
#pragma unroll
    for (int i = 0; i < kCpWords; i++) {
        vw[i] = w0 + i < n_words ? valid16[w0 + i] : 0u;
        rw[i] = w0 + i < n_words ? rc16[w0 + i] : 0u;
        c += __popc(vw[i]);
    }
    uint64_t idx = block_off[blockIdx.x] + block_exclusive_scan<kCpThreads, uint32_t>(c, s_wave);
    if (c == 0) return;
    // record of the thread's first item: the last r with rec_start[r] <= position (binary search), then it only advances
    uint64_t rec = 0, run = 0;
    bool have = false;
#pragma unroll
    for (int i = 0; i < kCpWords; i++) {
        uint32_t bits = vw[i];
        while (bits) {
            const int lead = __clz(bits) - 16;          // bit 15 - lead is the lowest remaining position
            bits &= ~(0x8000u >> lead);
            const uint64_t e = (w0 + i) * 16 + (uint64_t)lead;
            if (!have) {
                uint64_t lo = 0, hi = n_records;        // invariant: rec_start[lo] <= e < rec_start[hi]
                while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (rec_start[mid] <= e) lo = mid; else hi = mid; }
                rec = lo; have = true;
            } else if (e >= rec_start[rec + 1]) {
                atomicAdd(&counts[rec], (unsigned long long)run); run = 0;
                do rec++; while (e >= rec_start[rec + 1]);
            }
            run++;
            if (idx < cap) {
                if (pos_out) pos_out[idx] = e - index_shift - rec_start[rec];
                if (val_out) val_out[idx] = values[e];
                if (flag_out) flag_out[idx] = (uint8_t)((rw[i] >> (15 - lead)) & 1u);
            }
            idx++;
        }
    }
    if (run) atomicAdd(&counts[rec], (unsigned long long)run);
}

// ---------------------------------------------------------------------------------------------
// minimizers and quality masking
// ---------------------------------------------------------------------------------------------

// Windowed minimizers over a materialised canonical k-mer plane: the window of w+k-1 good bases ending at byte e holds
// the w k-mers ending at e-w+1 .. e; its minimizer (reference sequence::minimizer, src/sequence.rs:139-152, applied to
// that window) is the smallest of their canonical values, the leftmost one on ties.  A block streams tiles of 2048 window
// ends: the tile's values (+ the w-1 before it) and its flag bits go to LDS once - 8 B of HBM per position instead of
// 8 w, and no dependent chain of global loads - and every thread then walks its windows in LDS.  (A van Herk / Gil-Werman
// block scheme, 3 compare-selects per position whatever w, was measured SLOWER at w = 11: 17.9 vs 11.8 ms per 1.51 G
// positions - its per-block passes are serial LDS chains with a fifth of the threads idle and twice the LDS footprint.)
// Per-block partials like the scan.
constexpr int kWmThreads = 256, kWmTile = 2048, kWmMaxHalo = 256;
constexpr int kWmBitWords = (kWmTile + kWmMaxHalo) / 32;

// all of the w bits starting at bit `start` set?  (LSB-first u32 words)
__device__ __forceinline__ bool wm_all_set(const uint32_t *bits, uint32_t start, uint32_t w)
{
    uint32_t pos = start, left = w;
    while (left) {
        const uint32_t off = pos & 31u, room = 32u - off, take = left < room ? left : room;
        const uint32_t mask = (take == 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << off;
        if ((bits[pos >> 5] & mask) != mask) return false;
        pos += take; left -= take;
    }
    return true;
}

// W = 0: window size at run time (one LDS walk per window).  2 <= W <= 16: compile-time window, register van Herk - a thread
// takes a group of B = min(W-1, 4) consecutive window ends, loads the B+W-1 values they cover once, builds the running
// minimum of the first B values from the right (ties move left) and of the rest from the left (ties stay left) in
// registers, and every window is min(left part, right part) with ties to the left part: (2B+W-3)/B compare-selects and
// (B+W-1)/B LDS reads per window instead of W-1 and W.
template <int W>
__global__ __launch_bounds__(kWmThreads) void window_min_reduce_kernel(const uint64_t *values, const uint16_t *valid16, const uint16_t *rc16,
                                                                       uint64_t n, uint32_t w, uint32_t bin_shift,
                                                                       uint32_t *part_hist, uint64_t *part_scalars,
                                                                       uint64_t first_end = 0)
{
    // first_end: only windows ENDING at or after this position are counted (the host scans long inputs in chunks that
    // overlap by the w+k-2 bytes of left context)
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint64_t s_red[kWmThreads / 64][4];
    __shared__ uint64_t s_val[kWmTile + kWmMaxHalo];
    __shared__ uint32_t s_vbits[kWmBitWords], s_rbits[kWmBitWords];
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) s_hist[i] = 0;
    const uint32_t halo = (w - 1 + 15) & ~15u;             // positions kept before the tile, whole 16-bit plane words
    const uint64_t n16 = (n + 15) >> 4;                    // plane words
    const uint64_t n_tiles = (n + kWmTile - 1) / kWmTile;
    const uint32_t span = kWmTile + halo;
    uint64_t sum = 0, xr = 0, nv = 0, nf = 0;
    // The next tile's values and flag words are fetched into registers while the current tile is processed out of LDS
    // (one global round trip per tile would otherwise sit between the two barriers of every tile).
    constexpr int kPer = (kWmTile + kWmMaxHalo + kWmThreads - 1) / kWmThreads;   // values per thread per tile (9)
    uint64_t pv[kPer];
    uint32_t pb[2] = {0, 0};   // this thread's word of the valid / rc flag planes (threads 0 .. span/32)
    auto fetch = [&](uint64_t tile) {
        const int64_t region = (int64_t)(tile * kWmTile) - halo;   // first position held in LDS (may be negative)
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const uint32_t q = threadIdx.x + j * kWmThreads;
            const int64_t e = region + q;
            pv[j] = (q < span && e >= 0 && (uint64_t)e < (n16 << 4)) ? values[e] : 0ull;
        }
        if (threadIdx.x < (span + 31) / 32) {
            const int64_t w16 = (region >> 4) + 2 * (int64_t)threadIdx.x;   // region is a multiple of 16
            uint32_t v0 = 0, v1 = 0, r0 = 0, r1 = 0;
            if (w16 >= 0 && (uint64_t)w16 < n16) { v0 = valid16[w16]; r0 = rc16[w16]; }
            if (w16 + 1 >= 0 && (uint64_t)(w16 + 1) < n16) { v1 = valid16[w16 + 1]; r1 = rc16[w16 + 1]; }
            // plane words are MSB-first (bit 15 - e % 16); LDS words are LSB-first (bit = position % 32)
            pb[0] = (__brev(v0) >> 16) | (__brev(v1) & 0xFFFF0000u);
            pb[1] = (__brev(r0) >> 16) | (__brev(r1) & 0xFFFF0000u);
        }
    };
    const uint64_t tile0 = blockIdx.x + first_end / kWmTile;
    if (tile0 < n_tiles) fetch(tile0);
    for (uint64_t tile = tile0; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();   // the previous tile's readers are done (and the histogram is zeroed)
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const uint32_t q = threadIdx.x + j * kWmThreads;
            if (q < span) s_val[q] = pv[j];
        }
        if (threadIdx.x < (span + 31) / 32) { s_vbits[threadIdx.x] = pb[0]; s_rbits[threadIdx.x] = pb[1]; }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);   // in flight while this tile is processed
        if constexpr (W >= 2) {
            constexpr int B = W - 1 < 4 ? W - 1 : 4;          // window ends per group (more would only cost registers)
            constexpr int Z = B + W - 1;                      // values a group covers
            constexpr uint32_t n_groups = (kWmTile + B - 1) / B;
#pragma unroll 1
            for (uint32_t g = threadIdx.x; g < n_groups; g += kWmThreads) {
                const uint32_t p0 = g * B;                       // first window end of the group, tile-relative
                if (tile * kWmTile + p0 >= n) break;
                const uint32_t base = halo + p0 - (W - 1);       // LDS index of z[0]; window i of the group is z[i .. i+W-1]
                uint64_t z[Z];
                uint32_t zi[Z];
#pragma unroll
                for (int i = 0; i < Z; i++) {
                    const uint32_t q = base + i < span ? base + i : span - 1;   // past the tile: not used by a counted window
                    z[i] = s_val[q]; zi[i] = q;
                }
#pragma unroll
                for (int i = B - 2; i >= 0; i--)                 // left part z[0..B-1]: running minimum from its right end
                    if (!(z[i] <= z[i + 1])) { z[i] = z[i + 1]; zi[i] = zi[i + 1]; }
#pragma unroll
                for (int j = B + 1; j < Z; j++)                  // right part z[B..Z-1]: running minimum from its left end
                    if (!(z[j] < z[j - 1])) { z[j] = z[j - 1]; zi[j] = zi[j - 1]; }
#pragma unroll
                for (int i = 0; i < B; i++) {                    // window i = z[i..B-1] + z[B..i+W-1]
                    const uint32_t pos = p0 + i;
                    const uint64_t e = tile * kWmTile + pos;
                    if (pos >= (uint32_t)kWmTile || e >= n || e < first_end) continue;
                    if (!wm_all_set(s_vbits, base + i, W)) continue;
                    const bool right = z[i + W - 1] < z[i];      // tie: the older (left) part wins
                    const uint64_t best = right ? z[i + W - 1] : z[i];
                    const uint32_t at = right ? zi[i + W - 1] : zi[i];
                    const uint32_t flag = (s_rbits[at >> 5] >> (at & 31u)) & 1u;
                    sum += best; xr ^= best; nv++; nf += flag ? 0 : 1;
                    atomicAdd(&s_hist[(uint32_t)(best >> bin_shift)], 1u);
                }
            }
        } else {
#pragma unroll 1
        for (uint32_t pos = threadIdx.x; pos < (uint32_t)kWmTile; pos += kWmThreads) {
            if (tile * kWmTile + pos >= n) break;
            if (tile * kWmTile + pos < first_end) continue;
            const uint32_t first = halo + pos - (w - 1);          // leftmost k-mer of the window, LDS index
            if (!wm_all_set(s_vbits, first, w)) continue;
            uint64_t best = s_val[first];
            uint32_t at = first;
#pragma unroll 4
            for (uint32_t t = 1; t < w; t++) {                    // a strict '<' keeps the leftmost minimum
                const uint64_t v = s_val[first + t];
                if (v < best) { best = v; at = first + t; }
            }
            const uint32_t flag = (s_rbits[at >> 5] >> (at & 31u)) & 1u;
            sum += best; xr ^= best; nv++; nf += flag ? 0 : 1;
            atomicAdd(&s_hist[(uint32_t)(best >> bin_shift)], 1u);
        }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64); xr ^= __shfl_xor(xr, o, 64); nv += __shfl_xor(nv, o, 64); nf += __shfl_xor(nf, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = nv; s_red[threadIdx.x >> 6][1] = nf; s_red[threadIdx.x >> 6][2] = sum; s_red[threadIdx.x >> 6][3] = xr; }
    __syncthreads();
    uint32_t *ph = part_hist + (size_t)blockIdx.x * kHistBins;
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) ph[i] = s_hist[i];
    if (threadIdx.x == 0) {
        uint64_t tv = 0, tf = 0, ts = 0, tx = 0;
        for (int q = 0; q < kWmThreads / 64; q++) { tv += s_red[q][0]; tf += s_red[q][1]; ts += s_red[q][2]; tx ^= s_red[q][3]; }
        uint64_t *ps = part_scalars + (size_t)blockIdx.x * 4;
        ps[0] = tv; ps[1] = tf; ps[2] = ts; ps[3] = tx;
    }
}

// bitkmer::reverse_complement / bitkmer::minimizer on device (reference src/bitkmer.rs:112-132,146-162), element-wise.
// The reference takes the reverse complement of the m-mer at length k (not m) - reproduced as written.
__device__ __forceinline__ uint64_t bit_revcomp(uint64_t x, uint32_t k)
{
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    x = ~x;
    return x >> (2 * (32 - k));
}
__global__ void bit_minimizer_kernel(const uint64_t *in, uint64_t n, uint32_t k, uint32_t m, uint64_t *out)
{
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = in[i], lowest = ~0ull;
        for (uint32_t t = 0; t <= k - m; t++) {
            const uint64_t cur = v & mask;
            lowest = cur < lowest ? cur : lowest;
            const uint64_t r = bit_revcomp(cur, k);
            lowest = r < lowest ? r : lowest;
            v >>= 2;
        }
        out[i] = lowest;
    }
}

// bitkmer::reverse_complement / bitkmer::canonical (reference src/bitkmer.rs:112-143), element-wise.
// flags may be null (reverse complement only: out = rc); otherwise out = min-by-the-reference's-rule, flags = was_rc.
__global__ void bit_canonical_kernel(const uint64_t *in, uint64_t n, uint32_t k, int canonical, uint64_t *out, uint8_t *flags)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = in[i], r = bit_revcomp(v, k);
        if (!canonical) { out[i] = r; continue; }
        const bool was_rc = v > r;  // ties keep the forward k-mer (reference src/bitkmer.rs:138-142)
        out[i] = was_rc ? r : v;
        flags[i] = was_rc ? 1 : 0;
    }
}

// sequence::minimizer (reference src/sequence.rs:139-152) for one sequence: the lexicographically smallest length-m
// byte string among all windows of the sequence and of its reverse complement.  One block; candidates are compared
// as raw bytes exactly like the reference.  best[0] = candidate index (window start), best[1] = 1 if from the rc.
__device__ __forceinline__ uint8_t cand_byte(const uint8_t *seq, uint64_t n, const uint16_t *comp, uint64_t idx, uint32_t strand, uint32_t j)
{
    return strand ? (uint8_t)comp[seq[n - 1 - (idx + j)]] : seq[idx + j];
}
// three-way compare of two candidates (window start ia on strand sa / ib on sb) as raw bytes; a tie goes to the candidate the reference's loop
// meets first (src/sequence.rs:143-150: forward window i, then reverse-complement window i, i ascending: candidate number 2 i + strand)
__device__ __forceinline__ bool cand_less(const uint8_t *seq, uint64_t n, const uint16_t *comp, uint32_t m,
                                          uint64_t ia, uint32_t sa, uint64_t ib, uint32_t sb)
{
    for (uint32_t j = 0; j < m; j++) {
        const uint8_t a = cand_byte(seq, n, comp, ia, sa, j), b = cand_byte(seq, n, comp, ib, sb, j);
        if (a != b) return a < b;
    }
    return 2 * ia + sa < 2 * ib + sb;
}
__global__ __launch_bounds__(1024) void minimizer_bytes_kernel(const uint8_t *seq, uint64_t n, uint32_t m, const uint16_t *comp, uint64_t *best)
{
    __shared__ uint64_t s_idx[1024];
    __shared__ uint32_t s_str[1024];
    const uint64_t ncand = n - m + 1;
    uint64_t bi = 0; uint32_t bs = 0;
    bool have = false;
    for (uint64_t c = threadIdx.x; c < 2 * ncand; c += blockDim.x) {
        const uint64_t i = c >> 1; const uint32_t st = (uint32_t)(c & 1);
        if (!have || cand_less(seq, n, comp, m, i, st, bi, bs)) { bi = i; bs = st; have = true; }
    }
    s_idx[threadIdx.x] = have ? bi : ~0ull; s_str[threadIdx.x] = bs;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const uint64_t oi = s_idx[threadIdx.x + o]; const uint32_t os = s_str[threadIdx.x + o];
            if (oi != ~0ull && (s_idx[threadIdx.x] == ~0ull || cand_less(seq, n, comp, m, oi, os, s_idx[threadIdx.x], s_str[threadIdx.x]))) {
                s_idx[threadIdx.x] = oi; s_str[threadIdx.x] = os;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { best[0] = s_idx[0]; best[1] = s_str[0]; }
}
// sequence::minimizer for every record of a batch (records at offs[r] - offs[0] .. offs[r + 1] - offs[0] of seq): one wave per record, the
// 2 (n - m + 1) candidates dealt to the lanes, lane bests reduced through the wave; the m bytes of the winner, its window start (on its
// strand) and its strand are written per record.  Records shorter than m: the lowest such index is left in *bad (the reference panics
// there, src/sequence.rs:141); records longer than long_record are skipped here (the host runs the one-block kernel on each).
// (a record of up to kMinStage bytes is staged in LDS once, both strands - the candidates' bytes then come from there instead of
// global loads through the complement table: 3.5 -> 1.x ms per 1 M x 150 bp)
constexpr uint32_t kMinStage = 1024;
__device__ __forceinline__ bool staged_less(const uint8_t *fw, const uint8_t *rc, uint32_t m, uint32_t ca, uint32_t cb)   // candidates 2 i + strand
{
    const uint8_t *a = ((ca & 1u) ? rc : fw) + (ca >> 1), *b = ((cb & 1u) ? rc : fw) + (cb >> 1);
    for (uint32_t j = 0; j < m; j++)
        if (a[j] != b[j]) return a[j] < b[j];
    return ca < cb;
}
__global__ __launch_bounds__(256) void minimizer_batch_kernel(const uint8_t *seq, const uint64_t *offs, uint64_t n_records, uint32_t m, uint64_t long_record,
                                                              const uint16_t *comp, uint8_t *out, uint64_t *pos_out, uint8_t *rc_out, unsigned long long *bad)
{
    __shared__ uint8_t s_stage[4][2][kMinStage];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wv; r < n_records; r += waves) {
        const uint64_t b0 = offs[r] - offs[0], n = offs[r + 1] - offs[r];
        if (n < m) { if (lane == 0) atomicMin(bad, (unsigned long long)r); continue; }
        if (n > long_record) continue;
        const uint8_t *rec = seq + b0;
        const uint64_t ncand2 = 2 * (n - m + 1);
        uint64_t win;
        if (n <= kMinStage) {
            uint8_t *fw = s_stage[wv][0], *rcs = s_stage[wv][1];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the previous record's readers are done: one wave, in-order LDS)
            for (uint32_t j = lane; j < (uint32_t)n; j += 64) { const uint8_t by = rec[j]; fw[j] = by; rcs[(uint32_t)n - 1 - j] = (uint8_t)comp[by]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t bc = lane < (uint32_t)ncand2 ? lane : ~0u;
            for (uint32_t c = lane + 64; c < (uint32_t)ncand2; c += 64)
                if (staged_less(fw, rcs, m, c, bc)) bc = c;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t oc = (uint32_t)__shfl_down((int)bc, o, 64);
                if (lane + (uint32_t)o < 64u && oc != ~0u && (bc == ~0u || staged_less(fw, rcs, m, oc, bc))) bc = oc;
            }
            win = (uint32_t)__shfl((int)bc, 0, 64);
            const uint8_t *src = ((win & 1) ? rcs : fw) + (win >> 1);
            for (uint32_t j = lane; j < m; j += 64) out[r * m + j] = src[j];
        } else {
            uint64_t bc = lane;                       // candidate number 2 i + strand; lanes beyond the candidates hold none (~0)
            if (bc >= ncand2) bc = ~0ull;
            for (uint64_t c = (uint64_t)lane + 64; c < ncand2; c += 64)
                if (cand_less(rec, n, comp, m, c >> 1, (uint32_t)(c & 1), bc >> 1, (uint32_t)(bc & 1))) bc = c;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t oc = ((uint64_t)(uint32_t)__shfl_down((int)(bc >> 32), o, 64) << 32) | (uint32_t)__shfl_down((int)(uint32_t)bc, o, 64);
                if (lane + (uint32_t)o < 64u && oc != ~0ull && (bc == ~0ull || cand_less(rec, n, comp, m, oc >> 1, (uint32_t)(oc & 1), bc >> 1, (uint32_t)(bc & 1)))) bc = oc;
            }
            win = ((uint64_t)(uint32_t)__shfl((int)(bc >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)bc, 0, 64);
            for (uint32_t j = lane; j < m; j += 64) out[r * m + j] = cand_byte(rec, n, comp, win >> 1, (uint32_t)(win & 1), j);
        }
        if (lane == 0) {
            if (pos_out) pos_out[r] = win >> 1;
            if (rc_out) rc_out[r] = (uint8_t)(win & 1);
        }
    }
}
// (the one-block kernel's result for record r of a batch, written like the wave kernel writes its own)
__global__ void minimizer_emit_record_kernel(const uint8_t *rec, uint64_t n, uint32_t m, const uint16_t *comp, const uint64_t *best, uint64_t r, uint8_t *out,
                                             uint64_t *pos_out, uint8_t *rc_out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) out[r * m + j] = cand_byte(rec, n, comp, best[0], (uint32_t)best[1], j);
    if (j == 0) {
        if (pos_out) pos_out[r] = best[0];
        if (rc_out) rc_out[r] = (uint8_t)best[1];
    }
}
__global__ void minimizer_emit_kernel(const uint8_t *seq, uint64_t n, uint32_t m, const uint16_t *comp, const uint64_t *best, uint8_t *out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) out[j] = cand_byte(seq, n, comp, best[0], (uint32_t)best[1], j);
}

// QualitySequence::quality_mask (reference src/sequence.rs:285-296)
__global__ void quality_mask_kernel(const uint8_t *seq, const uint8_t *qual, uint64_t n, uint8_t score, uint8_t *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = qual[i] < score ? (uint8_t)'N' : seq[i];
}

// ---------------------------------------------------------------------------------------------
// synthetic reads: counter-based SplitMix64, one thread per 16 output bytes
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_reads_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                   uint32_t n_per_1024, uint8_t *out)
{
    const uint64_t total = n_reads * ((uint64_t)read_len + 1);
  for (uint64_t g0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; g0 < total; g0 += (uint64_t)gridDim.x * blockDim.x * 16) {
    const uint32_t wpr = (read_len + 31) / 32, npr = (read_len + 5) / 6;
    uint64_t r = g0 / (read_len + 1);
    uint32_t j = (uint32_t)(g0 - r * (read_len + 1));
    uint8_t buf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint8_t b = '\n';
        if (g0 + i < total) {
            if (j < read_len) {
                const uint64_t rr = first_read + r;
                const uint64_t w = splitmix64_at(seed, rr * wpr + j / 32);
                b = (uint8_t)"ACGT"[(w >> (2 * (j % 32))) & 3];
                if (n_per_1024) {
                    const uint64_t m = splitmix64_at(seed + 1, rr * npr + j / 6);
                    if (((m >> (10 * (j % 6))) & 1023) < n_per_1024) b = 'N';
                }
            }
            if (++j > read_len) { j = 0; r++; }
        }
        buf[i] = b;
    }
    if (g0 + 16 <= total) {
        *reinterpret_cast<u32x4 *>(out + g0) = *reinterpret_cast<u32x4 *>(buf);
    } else {
        for (int i = 0; i < 16 && g0 + i < total; i++) out[g0 + i] = buf[i];
    }
  }
}

#endif  // NTK_SCAN_TEMPLATES_ONLY

}  // namespace ntk
