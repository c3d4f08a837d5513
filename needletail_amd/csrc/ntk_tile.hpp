// ntk_tile.hpp — per-lane tile logic of the scan engine, written so that the SAME source compiles
// for gfx950 (inside the HIP kernels) and for the host (tests/emu/, a lock-step 64-lane emulation used
// by the CPU test-suite to check the bit manipulation without a GPU).  Nothing here touches memory.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NTK_HD __host__ __device__ __forceinline__
#else
#define NTK_HD inline
#endif

namespace ntk {

constexpr int kTileBytes = 1024;  // bytes LOADED per wave-tile = 64 lanes x 16 B = one coalesced dwordx4 load
constexpr int kHistBins = 4096;

struct ScanArgs {
    const uint8_t *seq;   // 16-B aligned, readable up to round_up(n_bytes, 16)
    uint64_t n_bytes;
    uint64_t n_tiles;     // ceil(ceil(n_bytes / 16) / 62): tiles of 62 emitting 16-byte slots
    uint64_t tile_begin;  // first tile of this launch (a launch covers tiles [tile_begin, tile_end))
    uint64_t tile_end;
    uint32_t *work_counters;  // n_shards counters, 16 u32 apart (one per 64-B line), zeroed before the launch
    uint32_t n_shards;        // min(8, grid): block b pulls tile chunks from shard b % n_shards
    uint32_t tiles_per_shard; // shard s owns tiles [tile_begin + s*tiles_per_shard, +tiles_per_shard) clipped to tile_end
    uint32_t chunk_tiles;     // tiles handed out per pull
    uint32_t tail_tile_rel;   // first tile (relative to tile_begin) that reaches n_bytes; 0xFFFFFFFF if beyond this launch
    uint32_t k;
    uint32_t sh_r;        // right shift of the reverse-complement stream: 64-2k (KW=2) / 32-2k (KW=1)
    uint32_t mask_hi;     // KW=2: (1 << (2k-32)) - 1
    uint32_t mask_lo;     // KW=1: (1 << 2k) - 1 (k=16: ~0); KW=2: ~0
    uint32_t smear[5];    // doubling shifts that OR a break bit over the k windows containing it
    uint32_t bin_shift;   // 2*(k - min(k,6))
    // reduce sink
    uint32_t *part_hist;     // [grid][kHistBins]
    uint64_t *part_scalars;  // [grid][4]: n_total, n_fwd, sum, xor
    // materialise sink
    uint64_t *values;
    uint16_t *valid16;
    uint16_t *rc16;
    // quality masking (QM builds): one quality byte per sequence byte, same alignment and padding rules as seq
    const uint8_t *qual;
    uint32_t q_add, q_sel;  // quality_cut(cutoff)
    // reduce scans started with NTK_FLAG_RESET: block 0 zeroes these words (the ctx accumulators) before anything else; the fold
    // kernel that adds this scan's partials into them runs after the scan in stream order
    uint64_t *zero_acc;
    uint32_t zero_words;
    // speculative scans of un-normalised byte-path input (scan2_kernel SPEC builds): raised when a loaded byte has bit 5 set; the launch also
    // clears the NEXT launch's flag word (a ring in the ctx), so that no memset sits between launches
    uint32_t *lower_flag, *lower_flag_next;
    // generic fused windowed minimizers (minimizer_scan_kernel): window of w k-mers; the first min_halo_lanes lanes of a tile emit no
    // window (2 lanes of k-mer halo + ceil((w - 1) / 16) lanes of k-mers that earlier windows need), the tile advances by
    // (64 - min_halo_lanes) * 16 bytes; min_smear = doubling shifts that OR a "k-mer invalid" bit over the w window ends it is part of
    uint32_t min_w, min_halo_lanes;
    uint32_t min_smear[6];
    uint32_t min_overlap;    // w - 2^floor(log2 w): the shift between the two overlapping power-of-two windows that make a window of w
    // k + w - 1 <= 49: ONE smear of the break bits over the k + w - 1 window ends a break spoils (own 16 + 48 earlier positions fit 64 bits)
    // instead of the k smear followed by the w smear; min_smear_kw[0] == 0: not used
    uint32_t min_smear_kw[6];
};

// Fills the k-derived fields (host side).  k must be 1..32.
inline void scan_args_set_k(ScanArgs &a, uint32_t k)
{
    a.k = k;
    if (k > 16) {
        a.sh_r = 64 - 2 * k;
        a.mask_hi = k == 32 ? 0xFFFFFFFFu : ((1u << (2 * k - 32)) - 1u);
        a.mask_lo = 0xFFFFFFFFu;
    } else {
        a.sh_r = 32 - 2 * k;
        a.mask_hi = 0;
        a.mask_lo = k == 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    }
    uint32_t len = 1;
    for (int i = 0; i < 5; i++) {
        uint32_t s = len < k ? (len < k - len ? len : k - len) : 0;
        a.smear[i] = s;
        len += s;
    }
    const uint32_t p = k < 6 ? k : 6;
    a.bin_shift = 2 * (k - p);
}

// ---------------------------------------------------------------------------------------------
// instruction-level helpers: gfx950 builtins on device, portable C on the host
// ---------------------------------------------------------------------------------------------
// v_bitop3_b32: any 3-input bitwise function in ONE full-rate VALU op (v_bfi/v_and_or/v_or3/v_lshl_or are half-rate
// on gfx950, tools/ubench.hip).  TT bit (a<<2 | b<<1 | c) is the result for that input combination.
template <int TT>
NTK_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (int i = 0; i < 8; i++)
        if ((TT >> i) & 1) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
#endif
}
NTK_HD uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return bitop3<0xCA>(mask, a, b); }  // (a & mask) | (b & ~mask)
NTK_HD uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return bitop3<0xEA>(a, b, c); }      // (a & b) | c
NTK_HD uint32_t or_and(uint32_t a, uint32_t b, uint32_t c) { return bitop3<0xA8>(a, b, c); }      // (a | b) & c

NTK_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)  // v_perm_b32: selector 0-3 -> lo bytes, 4-7 -> hi bytes
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xFF;
        const uint32_t b = s < 8 ? (uint32_t)((src >> (8 * s)) & 0xFF) : (s >= 13 ? 0xFFu : 0u);
        r |= b << (8 * i);
    }
    return r;
#endif
}

NTK_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)  // v_alignbit_b32: ({hi,lo} >> (sh & 31))[31:0]
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
#endif
}

NTK_HD uint32_t add_self(uint32_t t)  // t << 1 as a full-rate v_add_u32 (the compiler turns t + t into the half-rate v_lshlrev_b32)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(t));
    return r;
#else
    return t + t;
#endif
}

NTK_HD uint32_t brev32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

NTK_HD uint32_t dot4(uint32_t a, uint32_t b, uint32_t c)  // v_dot4_u32_u8: sum of the four byte products + c
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return c;
#endif
}

// 32 bits of a big-endian word stream starting `off` bits after the MSB of W[0].
template <int N>
NTK_HD uint32_t win32(const uint32_t (&W)[N], int off)
{
    const int wi = off >> 5, s = off & 31;
    if (s == 0) return W[wi];
    return alignbit(W[wi], (wi + 1 < N) ? W[wi + 1] : 0u, 32 - s);
}

// ---------------------------------------------------------------------------------------------
// encode: 16 raw ASCII bytes -> 2-bit codes (base 0 in bits 31:30), break mask (base i at bit 15-i),
// and the lane's word of the reverse-complement stream.  Pure SWAR, no LUT in memory.
// ---------------------------------------------------------------------------------------------
struct Raw16 { uint32_t x, y, z, w; };  // little-endian dwords: byte 0 of x is base 0

// ---------------------------------------------------------------------------------------------
// Quality masking fused into the load.  Reference QualitySequence::quality_mask, src/sequence.rs:285-296:
// `if q < score { b'N' } else { s }` per (base, quality) pair, ahead of normalize / the k-mer iterators.  On every path
// a masked base only has to stop being a base, so instead of writing 'N' the top bit of its byte is set: no good base
// (acgtuACGTU) and no deleted byte has it, normalize maps such a byte to N and the 2-bit LUT misses it.  Exact for all
// 256 quality values and cutoffs 1..255: q >= c  <=>  (c <= 128) ? (q >= 128 || q7 >= c) : (q >= 128 && q7 >= c - 128)
// with q7 = q & 127; `q7 + add` puts "q7 >= c mod 128" into the byte's top bit and cannot carry into the next byte
// (add <= 127).  4 full-rate VALU ops per dword (4 bases).
// ---------------------------------------------------------------------------------------------
struct QualityCut { uint32_t add, sel; };
inline QualityCut quality_cut(uint32_t cutoff)  // 1..255
{
    QualityCut c;
    c.add = ((cutoff <= 128 ? 128u - cutoff : 256u - cutoff) & 0x7Fu) * 0x01010101u;
    c.sel = cutoff <= 128 ? 0xFFFFFFFFu : 0u;
    return c;
}
NTK_HD uint32_t quality_break(uint32_t s, uint32_t q, uint32_t add, uint32_t sel)
{
    const uint32_t u = (q & 0x7F7F7F7Fu) + add;
    const uint32_t ge = bitop3<0xE8>(u, q, sel);      // majority: sel ? u | q : u & q  -> top bit of a byte = (q >= cutoff)
    return bitop3<0xF2>(s, ge, 0x80808080u);          // s | (~ge & 0x80808080)
}
NTK_HD Raw16 quality_break16(Raw16 s, Raw16 q, uint32_t add, uint32_t sel)
{
    return Raw16{quality_break(s.x, q.x, add, sel), quality_break(s.y, q.y, add, sel),
                 quality_break(s.z, q.z, add, sel), quality_break(s.w, q.w, add, sel)};
}
struct Enc {
    uint32_t code;   // 2-bit codes, MSB-first
    uint32_t rcode;  // group g (bits 2g+1:2g) = complement of base g: the reverse-complement stream word
    uint32_t bad;    // bit (15-i) set when byte i is not a base
};

template <bool ACCEPT_U>
NTK_HD Enc encode16(Raw16 d)
{
    // 4x4 byte transpose so that E_s holds bases {s, 4+s, 8+s, 12+s} with base s in the top byte.
    const uint32_t x0 = perm(d.x, d.y, 0x04000501u), x1 = perm(d.x, d.y, 0x06020703u);
    const uint32_t y0 = perm(d.z, d.w, 0x04000501u), y1 = perm(d.z, d.w, 0x06020703u);
    const uint32_t e0 = perm(x0, y0, 0x07060302u), e1 = perm(x0, y0, 0x05040100u);
    const uint32_t e2 = perm(x1, y1, 0x07060302u), e3 = perm(x1, y1, 0x05040100u);

    // ASCII bits 2:1 give A0 C1 T2 G3 (either case, U == T); merge the four slots of every byte.
    const uint32_t m = bfi(0xC0C0C0C0u, e0 << 5, bfi(0x30303030u, e1 << 3, bfi(0x0C0C0C0Cu, e2 << 1, e3 >> 1)));
    Enc r;
    r.code = m ^ ((m >> 1) & 0x55555555u);  // -> A0 C1 G2 T3 (reference src/bitkmer.rs:8-15)

    // complement (3 - c == ~c on 2 bits) and reverse the order of the 16 groups
    const uint32_t t = brev32(~r.code);
    r.rcode = bfi(0x55555555u, t >> 1, t << 1);

    // break mask: a byte is a base iff (byte & 0xDF) equals the letter its bits 2:1 select.
    // ACCEPT_U (normalize pipeline, reference src/sequence.rs:30): fold T(0x54) and U(0x55) together.
    constexpr uint32_t kTable = ACCEPT_U ? 0x47554341u : 0x47544341u;  // byte x: A C T/U G
    uint32_t h[4];
    const uint32_t e[4] = {e0, e1, e2, e3};
#pragma unroll
    for (int s = 0; s < 4; s++) {
        uint32_t u = e[s] & 0xDFDFDFDFu;
        if (ACCEPT_U) u = and_or(u >> 4, 0x01010101u, u);
        const uint32_t sel = (e[s] >> 1) & 0x03030303u;
        const uint32_t dif = perm(0u, kTable, sel) ^ u;
        h[s] = ((dif & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | dif;  // bit 7 of every byte: byte differs
    }
    uint32_t g = bfi(0x80808080u, h[0], bfi(0x40404040u, h[1] >> 1, bfi(0x20202020u, h[2] >> 2, h[3] >> 3)));
    g = (g >> 4) & 0x0F0F0F0Fu;
    g = or_and(g, g >> 4, 0x00FF00FFu);
    r.bad = or_and(g, g >> 8, 0xFFFFu);
    return r;
}

// ---------------------------------------------------------------------------------------------
// lane_tile: everything one lane does for one wave-tile.
//
// Tile geometry: a wave-tile is 64 lanes x 16 bytes, loaded with one coalesced dwordx4 per lane.  Lanes 0 and 1 only
// provide the k-1 <= 31 base halo; lanes 2..63 each own the 16 windows that END at their 16 bytes.  Consecutive
// tiles of a wave therefore advance by kTileStride = 62 x 16 = 992 bytes (the 32 halo bytes are re-read, an L2 hit),
// and no state is carried from tile to tile.
//
// Cross-lane traffic is always "the value the previous lane holds in the same register" (DPP wave_shr:1 on the
// device, XL::prev); slot ids only matter to the host emulation.
//   forward value, right-aligned, window ending at own base j:
//       lo = 32 bits ending at base j   = alignbit(C[lane-1], C[lane], 30-2j)        (independent of k)
//       hi = the 2k-32 bits before that = lo of the PREVIOUS LANE, same j, masked     (one v_and_b32_dpp)
//   reverse-complement value: R = per-lane word of the reverse-complement stream (group g = complement of base g);
//       {R[lane], R[lane-1], R[lane-2]} >> sh_r (sh_r = 64-2k or 32-2k) is read at the k-independent bit offset 2(15-j).
//   KW       1: k <= 16 (32-bit values)   2: 17 <= k <= 32 (64-bit values)
//   CANON    emit min(fwd, revcomp) with the strand flag; else the forward value, flag false
//   TIE_RC   fwd == rc reports flag true (byte path, reference src/kmer.rs:124-128);
//            false: flag false (bit path, reference src/bitkmer.rs:138-142)
// ---------------------------------------------------------------------------------------------
constexpr int kHaloLanes = 2;
constexpr int kTileSlots = 64 - kHaloLanes;         // emitting 16-byte slots per tile
constexpr int kTileStride = kTileSlots * 16;        // 992 bytes

enum { kSlotCode = 16, kSlotRcode = 17, kSlotBad = 18, kSlotBad1 = 19, kSlotQ1 = 20, kSlotCode1 = 21,
       kSlotFw = 22, kSlotRw = 38, kSlotSufLo = 54, kSlotSufHi = 70, kNumSlots = 86 };  // kSlotFw + g / kSlotRw + g, g = 0..15: window words handed to the next lane

template <int KW, bool CANON, bool TIE_RC, bool ACCEPT_U, int KFIX, class Sink, class XL>
NTK_HD void lane_tile(const ScanArgs &a, Sink &sink, XL &xl, Raw16 raw, int64_t lane_base, bool halo_lane, bool tail_tile)
{
    Enc en = encode16<ACCEPT_U>(raw);
    if (tail_tile) {  // wave-uniform: this tile reaches the end of the input; bytes at or beyond n_bytes are breaks
        const int64_t keep = (int64_t)a.n_bytes - lane_base;
        en.bad |= keep >= 16 ? 0u : (keep <= 0 ? 0xFFFFu : (0xFFFFu >> (uint32_t)keep));
    }
    const uint32_t c1 = xl.prev(kSlotCode, en.code);
    const uint32_t r1 = xl.prev(kSlotRcode, en.rcode);
    const uint32_t b1 = xl.prev(kSlotBad, en.bad);
    const uint32_t b2 = xl.prev(kSlotBad1, b1);

    // k-specialised builds (KFIX = 21, 31): every k-derived quantity is an immediate, and the hi word of a value is a
    // plain shift of the lo word computed D = k-16 positions away (see below) instead of its own funnel shift + mask.
    constexpr bool FIX = KFIX > 16 && KW == 2;
    const uint32_t sh_r = FIX ? (uint32_t)(64 - 2 * KFIX) : a.sh_r;
    const uint32_t mask_hi = FIX ? (KFIX == 32 ? 0xFFFFFFFFu : ((1u << ((2 * KFIX - 32) & 31)) - 1u)) : a.mask_hi;

    // windows containing a break: OR every break bit over the k window-end positions that follow it
    uint64_t bw = ((uint64_t)b2 << 32) | ((uint64_t)b1 << 16) | en.bad;
    if (FIX) {
        int len = 1;
#pragma unroll
        for (int i = 0; i < 5; i++) { const int sft = len < KFIX ? (len < KFIX - len ? len : KFIX - len) : 0; bw |= bw >> sft; len += sft; }
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) bw |= bw >> a.smear[i];
    }
    const uint32_t inval = halo_lane ? 0xFFFFu : ((uint32_t)bw & 0xFFFFu);
    sink.begin_tile(lane_base, inval, halo_lane);

    uint32_t Q[3];
    Q[0] = en.rcode >> sh_r;
    Q[1] = alignbit(en.rcode, r1, sh_r);
    Q[2] = KW == 2 ? xl.prev(kSlotQ1, Q[1]) : 0u;  // == alignbit(R[lane-1], R[lane-2], sh_r)
    uint32_t vbits = inval << 16;  // bit 31 = window ending at own base 0; one flag is shifted out per position
    const uint32_t mask_hi_v = mask_hi;
    // FIX: all lo words first.  hi of the forward value at j = top 2k-32 bits of the lo word at j-D (same lane for
    // j >= D); hi of the reverse-complement value at j = top bits of its lo word at j+D (same lane for j+D <= 15).
    constexpr int D = FIX ? KFIX - 16 : 0, S = FIX ? 64 - 2 * KFIX : 0;
    uint32_t fls[16], rls[16], W2[3] = {0u, c1, en.code};
    if (FIX) {
        W2[0] = xl.prev(kSlotCode1, c1);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            fls[j] = j == 15 ? en.code : alignbit(c1, en.code, 30 - 2 * j);
            rls[j] = win32(Q, 32 + 2 * (15 - j));
        }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t fh = 0, fl, rh = 0, rl;
        if (FIX) {
            fl = fls[j]; rl = rls[j];
            fh = j >= D ? (S ? fls[j >= D ? j - D : 0] >> S : fls[j >= D ? j - D : 0]) : (win32(W2, 2 + 2 * j) & mask_hi);
            rh = j + D <= 15 ? (S ? rls[j + D <= 15 ? j + D : 0] >> S : rls[j + D <= 15 ? j + D : 0]) : (win32(Q, 2 * (15 - j)) & mask_hi);
        } else if (KW == 2) {
            fl = j == 15 ? en.code : alignbit(c1, en.code, 30 - 2 * j);
            fh = xl.prev_and(j, fl, mask_hi_v);
            rh = win32(Q, 2 * (15 - j)) & mask_hi;
            rl = win32(Q, 32 + 2 * (15 - j));
        } else {
            fl = j == 15 ? en.code : alignbit(c1, en.code, 30 - 2 * j);
            fl &= a.mask_lo;
            rl = win32(Q, 2 * (15 - j)) & a.mask_lo;
        }
        bool take_fwd = true;
        if (CANON) {
            if (KW == 2) {
                const uint64_t f = ((uint64_t)fh << 32) | fl, r = ((uint64_t)rh << 32) | rl;
                take_fwd = TIE_RC ? (f < r) : (f <= r);
            } else {
                take_fwd = TIE_RC ? (fl < rl) : (fl <= rl);
            }
        }
        const bool valid = !__builtin_add_overflow(vbits, vbits, &vbits);  // v_add_co_u32: carry-out = the flag
        sink.emit(j, valid, take_fwd, take_fwd ? fh : rh, take_fwd ? fl : rl);
    }
    sink.end_tile();
}

// ---------------------------------------------------------------------------------------------
// Generic fused windowed minimizers (minimizer_scan_kernel, ntk_kernels.hpp): the per-lane part - keys of the lane's 16 k-mers, window
// validity, and the sliding minimum over ANY run-time w <= 49 (k <= 31).  Semantics and scheme: the comment block above the kernel.
// Cross-lane words go through XL::prev_auto (device: DPP wave_shr:1; host emulation: the call sites of a lane numbered in order).
// Round 5: the bytes are decoded by the scan2 encode (encode16_sv2 below; the round-1 transpose encode cost 56 VALU per tile, this one 19 +
// 22 for the lane's break mask), and the keys of 26 <= k <= 31 are (value, strand) register triples compared strictly (4 VALU per minimum
// where the (value << 1 | strand) form needed 5 and a register pair built per compare).
//
// Two key forms:
//   KeyF (k <= 25): bit 62 | value << 11 | (tile position x = 16 * lane + j) << 1 | strand bit - unique per position and ordered by
//        (value, position), bit 61 clear: a positive NORMAL double whose order is its bit pattern's, so ONE v_min_f64 is the leftmost minimum;
//   KeyG (any k): the value in a register pair, the strand flag beside it.  A minimum takes its RIGHT operand only when that one's value is
//        strictly smaller: the left operand always covers the older positions, so ties go to the leftmost k-mer with no position in the key.
// ---------------------------------------------------------------------------------------------
struct KeyF { uint64_t k; };
struct KeyG { uint64_t v; uint32_t s; };

NTK_HD KeyF key_min(KeyF l, KeyF r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    KeyF m;
    asm("v_min_f64 %0, %1, %2" : "=v"(m.k) : "v"(l.k), "v"(r.k));
    return m;
#else
    return l.k < r.k ? l : r;   // (keys are unique per position; positive normal doubles order like their patterns)
#endif
}
NTK_HD KeyG key_min(KeyG l, KeyG r)
{
    const bool t = r.v < l.v;   // v_cmp_lt_u64 into vcc, three v_cndmask on it
    KeyG m;
    m.v = t ? r.v : l.v;
    m.s = t ? r.s : l.s;
    return m;
}
template <class XL>
NTK_HD KeyF key_prev(XL &xl, KeyF v)
{
    const uint32_t hi = xl.prev_auto((uint32_t)(v.k >> 32));
    KeyF r;
    r.k = ((uint64_t)hi << 32) | xl.prev_auto((uint32_t)v.k);
    return r;
}
template <class XL>
NTK_HD KeyG key_prev(XL &xl, KeyG v)
{
    uint32_t hi = xl.prev_auto((uint32_t)(v.v >> 32));
    KeyG r;
    uint32_t lo = xl.prev_auto((uint32_t)v.v);
    r.v = ((uint64_t)hi << 32) | lo;
    r.s = xl.prev_auto(v.s);
    return r;
}
// X[x] <- min(Y[x - Q], X[x]) for the 16 own positions, Q = 1, 2, 4, 8, 16; Y may be X itself (doubling).  In place, descending j;
// the words that come from the previous lane are fetched first.
template <int Q, class Key, class XL>
NTK_HD void min_shifted(XL &xl, Key (&X)[16], const Key (&Y)[16])
{
    if constexpr (Q < 16) {
        Key imp[Q];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < Q; j++) imp[j] = key_prev(xl, Y[16 + j - Q]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 15; j >= 0; j--) X[j] = key_min(j >= Q ? Y[j - Q] : imp[j], X[j]);
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int h = 0; h < 2; h++) {   // eight at a time: fewer live registers
            Key imp[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int j = 0; j < 8; j++) imp[j] = key_prev(xl, Y[8 * h + j]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int j = 0; j < 8; j++) X[8 * h + j] = key_min(imp[j], X[8 * h + j]);
        }
    }
}
// window[x] = min(M[x - S], M[x]) for a shift 0 < S < 16: two overlapping windows of M's span cover span + S positions.  The windows leave in
// groups of four positions (sink.emit4(first position, keys)): whoever consumes them does so at once, so that the 16 results are never all
// live beside the keys they are made from (register budget).
template <int S, class Key, class XL, class Sink>
NTK_HD void min_overlap(XL &xl, Sink &sink, const Key (&M)[16])
{
    static_assert(S > 0 && S <= 17, "shifts the two overlapping windows of W <= 49 can need");
    constexpr int NI = S < 16 ? S : 16;
    Key imp[NI];   // M[x - S] for the positions whose source lies in the previous lane (S = 17, j = 0: the lane before that)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NI; j++) imp[j] = (j - S + 16 >= 0) ? key_prev(xl, M[(j - S + 16) & 15]) : key_prev(xl, key_prev(xl, M[(j - S + 32) & 15]));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jb = 0; jb < 16; jb += 4) {
        Key g[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; i++) { const int j = jb + i; g[i] = key_min(j >= S ? M[(j - S) & 15] : imp[j < NI ? j : 0], M[j]); }
        sink.emit4(jb, g);
    }
}

// The lane's 16-bit break mask (base i at bit 15 - i) from the scan2 encode's expected / actual letters: per dword one xor, the SWAR
// "byte is not zero" test and ONE v_dot4_u32_u8 that gathers the four flags with the weights of their bit positions (22 VALU for 16 bases).
NTK_HD uint32_t bad16_from_letters(const uint32_t (&ex)[4], const uint32_t (&uu)[4])
{
    uint32_t hi = 0, lo = 0;   // 0x80 x (flags of bases 0..7 / 8..15, first base in bit 7)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int d = 0; d < 4; d++) {
        const uint32_t dif = ex[d] ^ uu[d];
        const uint32_t nz = or_and((dif & 0x7F7F7F7Fu) + 0x7F7F7F7Fu, dif, 0x80808080u);   // 0x80 in every byte that differs
        const uint32_t wt = (d & 1) ? 0x01020408u : 0x10204080u;                           // byte b of dword d = base 4 d + b -> weight 2^(7 - (4 d + b) % 8)
        if (d < 2) hi = dot4(nz, wt, hi); else lo = dot4(nz, wt, lo);
    }
    return (hi << 1) | (lo >> 7);
}

// invw bit 15 - j set = the window ending at own byte j is not emitted: one of its k + w - 1 bytes is a break (bad = the lane's break mask,
// base i at bit 15 - i), or the lane is one of the tile's a.min_halo_lanes non-emitting lanes.  k + w - 1 <= 49: ONE smear of the break bits
// over the window ends a break spoils (own 16 + 48 earlier positions fit 64 bits); longer spans: the k smear (k-mers), then the w smear.
// (Round 5 tried the scalar form instead - the bytes compared into lane masks, window_masks_runtime below for the windows: 28 VALU instead of
//  63 - and dropped it: the 64 mask SGPRs do not fit beside this kernel's other scalar state and travel through v_writelane / v_readlane,
//  ~75 VALU per tile; profiles/r05b/README.md.)
template <class XL>
NTK_HD uint32_t minimizer_invalid16(const ScanArgs &a, XL &xl, uint32_t bad, int64_t lane_base, uint32_t lane, bool tail_tile)
{
    if (tail_tile) {  // wave-uniform: this tile reaches the end of the input; bytes at or beyond n_bytes are breaks
        const int64_t keep = (int64_t)a.n_bytes - lane_base;
        bad |= keep >= 16 ? 0u : (keep <= 0 ? 0xFFFFu : (0xFFFFu >> (uint32_t)keep));
    }
    uint32_t inval = bad;
    const bool one_smear = a.min_smear_kw[0] != 0;
    if (!one_smear) {
        const uint32_t b1 = xl.prev_auto(bad), b2 = xl.prev_auto(b1);
        uint64_t bw = ((uint64_t)b2 << 32) | ((uint64_t)b1 << 16) | bad;   // windows of k containing a break
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 5; i++) bw |= bw >> a.smear[i];
        inval = lane < (uint32_t)kHaloLanes ? 0xFFFFu : ((uint32_t)bw & 0xFFFFu);
    }
    const uint32_t b1 = xl.prev_auto(inval), b2 = xl.prev_auto(b1), b3 = xl.prev_auto(b2);
    uint64_t bw = ((uint64_t)b3 << 48) | ((uint64_t)b2 << 32) | ((uint64_t)b1 << 16) | inval;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 6; i++) {
        const uint32_t sft = one_smear ? a.min_smear_kw[i] : a.min_smear[i];
        if (sft) bw |= bw >> sft;   // (wave-uniform)
    }
    return lane < a.min_halo_lanes ? 0xFFFFu : ((uint32_t)bw & 0xFFFFu);
}

// The f64 keys of a lane's 16 k-mers (k <= 25) straight from the code streams, both strands, no compare and no select: the key of a strand is
// two stream windows with constant fields merged in, and the strand choice is the same v_min_f64 as everything after it.
//     key = bit 62 | value << 11 | tag,   tag = tile position (16 * lane + j) << 1 | strand bit
//   * lo word = the 32 stream bits that start 11 bits below the value's end, low 11 bits replaced by the tag (one funnel shift + one
//     v_bitop3); forward positions j >= 10 would need bits of the NEXT lane there and shift the window word up instead;
//   * hi word = the 32 stream bits 21 above the value's end, cut to the value's remaining 2k - 21 bits, bit 30 set (funnel shift + v_bitop3);
//   * the reverse-complement stream is first right-aligned to the value (R >> (64 - 2k), four words: the run-time k is all in that shift);
//   * strand bit: the strand that wins a tie (equal values at the same position) carries 0 - reverse complement under TIE_RC (reference
//     src/kmer.rs:124-128), forward otherwise (src/bitkmer.rs:138-142); so bit 0 of a key says "forward" under TIE_RC and "reverse
//     complement" otherwise.
template <bool TIE_RC, class XL>
NTK_HD void minimizer_keys_f64(const ScanArgs &a, XL &xl, uint32_t code, uint32_t rcode, uint32_t lane, KeyF (&key)[16])
{
    const uint32_t c1 = xl.prev_auto(code), c2 = xl.prev_auto(c1);
    const uint32_t r1 = xl.prev_auto(rcode);
    // Q = (rcode : r1 : r2 : r3) >> (64 - 2k), Q[3] the least significant word; the value whose top group is own base j sits at Q bits
    // [34 + 2j, 34 + 2j + 2k)
    const uint32_t sh = 64u - 2u * a.k;   // 14 .. 62
    uint32_t Q[4];
    if (sh < 32u) {
        Q[0] = rcode >> sh; Q[1] = alignbit(rcode, r1, sh); Q[2] = xl.prev_auto(Q[1]); Q[3] = xl.prev_auto(Q[2]);
    } else {
        Q[0] = 0u; Q[1] = rcode >> (sh - 32u); Q[2] = alignbit(rcode, r1, sh - 32u); Q[3] = xl.prev_auto(Q[2]);
    }
    const uint32_t vbits = 2u * a.k;
    uint32_t mask_lo = vbits >= 21u ? 0xFFFFF800u : (((1u << vbits) - 1u) << 11);
    uint32_t mask_hi = vbits > 21u ? (1u << (vbits - 21u)) - 1u : 0u;
    uint32_t marker = 0x40000000u;
#if defined(__HIP_DEVICE_COMPILE__)
    // the three wave-uniform operands of the 64 v_bitop3 below, held in VGPRs: with an SGPR operand a v_bitop3 issues at half rate
    asm("" : "+v"(mask_lo)); asm("" : "+v"(mask_hi)); asm("" : "+v"(marker));
#endif
    // (the tag's per-position part, 2 j, goes in AFTER the strand choice - both strands carry the same position - as one v_or with an inline
    //  constant: with it inside, the compiler keeps 32 loop-invariant tag words in registers and the kernel spills)
    constexpr uint32_t fbitF = TIE_RC ? 1u : 0u, fbitR = TIE_RC ? 0u : 1u;
    const uint32_t tagF = (lane << 5) | fbitF, tagR = (lane << 5) | fbitR;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 16; j++) {
        // forward: the value ends (LSB) at stream bit e = 30 - 2j of (c2 : c1 : code)
        uint32_t fL, fH;
        if (j <= 9) {
            fL = alignbit(c1, code, 19 - 2 * j);            // stream bits [e - 11, e + 21)
            fH = alignbit(c2, c1, 19 - 2 * j);              // stream bits [e + 21, e + 53)
        } else {
            const uint32_t fl = j == 15 ? code : alignbit(c1, code, 30 - 2 * j);
            fL = fl << 11;
            fH = alignbit(c1, code, 51 - 2 * j);
        }
        // reverse complement: the value's LSB at Q bit p0 = 34 + 2j; lo window starts at p0 - 11 = 23 + 2j, hi window at p0 + 21 = 55 + 2j
        const int sL = 23 + 2 * j, sH = 55 + 2 * j;
        uint32_t rL = sL < 32 ? alignbit(Q[2], Q[3], sL) : alignbit(Q[1], Q[2], sL - 32);
        uint32_t rH = sH < 64 ? alignbit(Q[1], Q[2], sH - 32) : alignbit(Q[0], Q[1], sH - 64);
        KeyF kf, kr;
        kf.k = ((uint64_t)and_or(fH, mask_hi, marker) << 32) | and_or(fL, mask_lo, tagF);
        kr.k = ((uint64_t)and_or(rH, mask_hi, marker) << 32) | and_or(rL, mask_lo, tagR);
        key[j] = key_min(kf, kr);
        key[j].k |= (uint64_t)(2 * j);
    }
}

// The (value, strand) keys of a lane's 16 k-mers for any k <= 32: both strands' values as stream windows (forward: the 32 bits ending at
// base j + the previous lane's same word, masked; reverse complement: the stream right-aligned to the value, two windows), one 64-bit compare
// with the path's tie rule, three selects.  s = 0: forward strand, 1: reverse complement.
template <int KW, bool TIE_RC, class XL>
NTK_HD void minimizer_keys_general(const ScanArgs &a, XL &xl, uint32_t code, uint32_t rcode, KeyG (&key)[16])
{
    const uint32_t c1 = xl.prev_auto(code), r1 = xl.prev_auto(rcode);
    uint32_t mask_hi = a.mask_hi, mask_lo = a.mask_lo;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(mask_hi)); asm("" : "+v"(mask_lo));   // (in VGPRs: a plain v_and with an SGPR operand issues at half rate)
#endif
    uint32_t Q[3];
    Q[0] = rcode >> a.sh_r;
    Q[1] = alignbit(rcode, r1, a.sh_r);
    Q[2] = KW == 2 ? xl.prev_auto(Q[1]) : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 16; j++) {
        uint32_t fl = j == 15 ? code : alignbit(c1, code, 30 - 2 * j), fh = 0, rl, rh = 0;
        if (KW == 2) {
            fh = xl.prev_auto(fl) & mask_hi;
            rh = win32(Q, 2 * (15 - j)) & mask_hi;
            rl = win32(Q, 32 + 2 * (15 - j));
        } else {
            fl &= a.mask_lo;
            rl = win32(Q, 2 * (15 - j)) & mask_lo;
        }
        const uint64_t f = ((uint64_t)fh << 32) | fl, r = ((uint64_t)rh << 32) | rl;
        const bool take_rc = TIE_RC ? (r <= f) : (r < f);   // one compare: ties report the reverse complement on the byte path only
        const uint32_t vl = take_rc ? rl : fl, vh = take_rc ? rh : fh;   // (selected word by word: as one 64-bit select the compiler turns it into a minimum and compares twice)
        key[j].v = ((uint64_t)vh << 32) | vl;
        key[j].s = take_rc ? 1u : 0u;
    }
}

// Windows of W >= 17 k-mers by prefix / suffix minima over the lanes' blocks of 16 (van Herk / Gil-Werman): such a window holds the lane's own
// positions 0 .. j (the prefix minimum P[j]), H or H - 1 whole lanes before them and a suffix of one lane more.  With W - 1 = 16 H + D:
//     j >= D:  min(S_-H[j - D], F_(H-1), P[j])          j < D:  min(S_-(H+1)[16 + j - D], F_H, P[j])
// S_-h = the suffix minima of the lane h lanes back, F_m = the minimum over the m whole lanes before this one.  Operands are always ordered
// older first, so the leftmost rule holds for both key forms.  30 + 16 + D (+ 16 + H for H > 1) minima where doubling takes 16 per round -
// 48 instead of 80 at W = 19 - and 16 H + D + H imported keys.  D is a template constant (it indexes registers), H a wave-uniform loop count.
template <int D, class Key, class XL, class Sink>
NTK_HD void min_van_herk(uint32_t H, XL &xl, Sink &sink, const Key (&M)[16])
{
    // (register budget: the suffix minima T and the keys M are all that stays live - the prefix minimum runs along with the windows, which
    //  leave in groups of four; the lane's whole minimum is T[0])
    Key T[16];
    T[15] = M[15];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 14; j >= 0; j--) T[j] = key_min(M[j], T[j + 1]);
    // hop the suffix minima and the lane minimum back lane by lane; Fa = F_(H-1), Fb = F_H (oldest lane first in every minimum)
    Key f = T[0], Fa = T[0], Fb = T[0];      // (Fa is not used when H == 1)
    for (uint32_t h = 1; h <= H; h++) {      // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 16; i++) T[i] = key_prev(xl, T[i]);
        f = key_prev(xl, f);                 // the whole-lane minimum of lane -h
        Fa = Fb;
        Fb = h == 1 ? f : key_min(f, Fb);
    }
    Key p = M[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jb = 0; jb < 16; jb += 4) {
        Key g[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; i++) {
            const int j = jb + i;
            if (j) p = key_min(p, M[j]);
            if (j >= D) {
                Key l = T[j >= D ? j - D : 0];
                if (H > 1) l = key_min(l, Fa);
                g[i] = key_min(l, p);
            } else {
                g[i] = key_min(key_min(key_prev(xl, T[j < D ? 16 - D + j : 0]), Fb), p);   // S_-(H+1)[16 + j - D]: one hop more
            }
        }
        sink.emit4(jb, g);
    }
}

// sliding minimum over W = a.min_w.  W <= 16: M doubles while 2q <= W, then two overlapping windows of q make W; W >= 17: prefix / suffix
// minima (above).  Every branch is wave-uniform.  A[j] = the minimizer's key of the window ending at own byte j.
template <class Key, class XL, class Sink>
NTK_HD void minimizer_slide(const ScanArgs &a, XL &xl, Key (&M)[16], Sink &sink)
{
    const uint32_t W = a.min_w;
    constexpr bool kVanHerk = sizeof(Key) == sizeof(KeyF);   // (the (value, strand) triples need 96 registers for it and spill: they double on)
    if (kVanHerk && W >= 17) {
        const uint32_t H = (W - 1) >> 4;
        switch ((W - 1) & 15) {
#define NTK_VH_CASE(D) case D: min_van_herk<D>(H, xl, sink, M); break;
            NTK_VH_CASE(0) NTK_VH_CASE(1) NTK_VH_CASE(2) NTK_VH_CASE(3) NTK_VH_CASE(4) NTK_VH_CASE(5) NTK_VH_CASE(6) NTK_VH_CASE(7)
            NTK_VH_CASE(8) NTK_VH_CASE(9) NTK_VH_CASE(10) NTK_VH_CASE(11) NTK_VH_CASE(12) NTK_VH_CASE(13) NTK_VH_CASE(14) NTK_VH_CASE(15)
#undef NTK_VH_CASE
        }
        return;
    }
    if (W >= 2) min_shifted<1>(xl, M, M);
    if (W >= 4) min_shifted<2>(xl, M, M);
    if (W >= 8) min_shifted<4>(xl, M, M);
    if (W >= 16) min_shifted<8>(xl, M, M);
    if (!kVanHerk && W >= 32) min_shifted<16>(xl, M, M);
    switch (a.min_overlap) {   // W - (M's span): 0 .. 7 for W <= 16, 0 .. 17 for W <= 49
#define NTK_MIN_CASE(S) case S: min_overlap<S>(xl, sink, M); break;
        NTK_MIN_CASE(1) NTK_MIN_CASE(2) NTK_MIN_CASE(3) NTK_MIN_CASE(4) NTK_MIN_CASE(5) NTK_MIN_CASE(6) NTK_MIN_CASE(7)
#define NTK_MIN_CASE_G(S) case S: if constexpr (!kVanHerk) min_overlap<S>(xl, sink, M); break;
        NTK_MIN_CASE_G(8) NTK_MIN_CASE_G(9) NTK_MIN_CASE_G(10) NTK_MIN_CASE_G(11) NTK_MIN_CASE_G(12) NTK_MIN_CASE_G(13) NTK_MIN_CASE_G(14)
        NTK_MIN_CASE_G(15) NTK_MIN_CASE_G(16) NTK_MIN_CASE_G(17)
#undef NTK_MIN_CASE_G
#undef NTK_MIN_CASE
        default:
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int jb = 0; jb < 16; jb += 4) { Key g[4] = {M[jb], M[jb + 1], M[jb + 2], M[jb + 3]}; sink.emit4(jb, g); }
    }
}
// (value's low 32 bits, bits above them, strand bit as the key form carries it) of a window's minimizer
NTK_HD void key_fields(KeyF k, uint32_t &lo, uint32_t &hi, uint32_t &sbit)
{
    const uint32_t mh = (uint32_t)(k.k >> 32), ml = (uint32_t)k.k;
    lo = alignbit(mh, ml, 11);
    hi = mh >> 11;          // (bit 19 = the key's marker bit: taken out by whoever sums these)
    sbit = ml & 1u;
}
NTK_HD void key_fields(KeyG k, uint32_t &lo, uint32_t &hi, uint32_t &sbit) { lo = (uint32_t)k.v; hi = (uint32_t)(k.v >> 32); sbit = k.s; }

// host side: the run-time geometry of a window length (ScanArgs::min_*), w = 1..49
inline void scan_args_set_window(ScanArgs &a, uint32_t w)
{
    a.min_w = w;
    a.min_halo_lanes = (uint32_t)kHaloLanes + (w - 1 + 15) / 16;
    uint32_t q = 1;
    while (2 * q <= w) q *= 2;
    a.min_overlap = w - q;
    uint32_t len = 1;
    for (int i = 0; i < 6; i++) { const uint32_t sft = len < w ? (len < w - len ? len : w - len) : 0; a.min_smear[i] = sft; len += sft; }
    const uint32_t kw = a.k + w - 1;   // (scan_args_set_k first)
    len = 1;
    for (int i = 0; i < 6; i++) { const uint32_t sft = kw <= 49 && len < kw ? (len < kw - len ? len : kw - len) : 0; a.min_smear_kw[i] = sft; len += sft; }
    if (kw == 1) a.min_smear_kw[0] = 0;
}

// ---------------------------------------------------------------------------------------------
// "sv" variant of the tile logic: window validity lives in SCALAR registers.
//
// On the device the 16 per-byte break flags are produced directly as 64-bit lane masks B[i] (bit l = byte i of lane l
// is a break) by 16 SDWA byte compares, and the "window ending at byte j of lane l contains a break" masks V[j] follow
// from them with scalar OR / shift only: the window covers bytes j-k+1..j, i.e. (for k >= 17) bytes 0..j of the lane,
// bytes a..15 of the previous lane (a = max(0, 17+j-k); a lane shift is a 1-bit shift of the mask) and, when 17+j-k < 0,
// bytes 33+j-k..15 of the lane before.  The mask then gates the emit through exec with no VALU work per position.
// k is a compile-time constant here.
// The host emulation computes B from the same per-lane expected/actual bytes and runs the same mask algebra.
// (Rounds 1 - 2 had a first generation of this scheme - transpose encode, finished masks OK[j], one emit per position: lane_tile_sv /
// lane_tile_sv1 - which no build of the library used after round 3; removed in round 6, see profiles/history/.)
// ---------------------------------------------------------------------------------------------
// OK[j] (window ending at byte j is emitted) = A[j] & B[j] with the last AND left open (A = the lane's own prefix, B = what the previous
// lanes contribute): the masked region forms exec with that AND itself (s_and_b64 exec, A, B) - one scalar op instead of AND + move.
// Positive logic; a lane shift fills lane 0 with "not a base", and lanes 0 and 1 (halo lanes) are cleared: all their windows are invalid.
template <int K>
NTK_HD void window_masks_ab(const uint64_t (&G)[16], uint64_t (&A)[16], uint64_t (&B)[16])
{
    static_assert(K >= 17 && K <= 32, "sv path is built for 17 <= k <= 32");
    uint64_t S[16];
    S[15] = G[15];
#pragma unroll
    for (int i = 14; i >= 0; i--) S[i] = S[i + 1] & G[i];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int c = 17 + j - K, a_ = c > 0 ? c : 0;
        uint64_t v = S[a_] << 1;
        if (c < 0) v &= S[(33 + j - K) & 15] << 2;
        B[j] = v;
    }
#ifdef NTK_X_EXACTHALO   // the lane shifts of B already clear every window that reaches before lane 0: lane 0 always, lane 1 for j < K - 17
    A[0] = G[0];
#else
    A[0] = G[0] & ~3ull;  // halo lanes 0/1: cleared once here, inherited by every prefix
#endif
#pragma unroll
    for (int j = 1; j < 16; j++) A[j] = A[j - 1] & G[j];
}

// ---------------------------------------------------------------------------------------------
// The same for k <= 16 (32-bit values).  The window ending at byte j covers bytes j-K+1 .. j: inside the lane when j >= K-1, else
// reaching into the previous lane (never further: K <= 16).  9 <= K <= 16: prefix / suffix ANDs (about 70 scalar ops: the scalar unit
// is shared by the CU's four SIMDs and these kernels keep it busy) - a window that reaches into the previous lane is (own prefix [0..j])
// AND (the previous lane's suffix from byte 17+j-K, one lane shift); one inside the lane straddles the middle of the 16, so it is
// (suffix of the first half from its start) AND (prefix of the second half up to j), or a whole prefix / suffix when it touches byte
// 0 / 15.  K <= 8: with E = [previous lane's 16 masks shifted one lane up, own 16 masks] a sliding AND of width K over E, built by
// doubling (widths 1, 2, 4, 8) and one overlap step.  OK[j] = A[j] & B[j] as in window_masks_ab.
// ---------------------------------------------------------------------------------------------
template <int K>
NTK_HD void window_masks1_ab(const uint64_t (&G)[16], uint64_t (&A)[16], uint64_t (&B)[16], const uint64_t kNoHalo = ~3ull)
{
    static_assert(K >= 1 && K <= 16, "k <= 16 variant");
    constexpr uint64_t kAll = ~0ull;   // kNoHalo: halo lanes (0 / 1 in the k-mer kernels) emit nothing
    if constexpr (K >= 9) {
        uint64_t P[16], S[16], S8[8], P8[16];
        P[0] = G[0] & kNoHalo;            // cleared in every window that contains own byte 0 ...
#pragma unroll
        for (int j = 1; j < 16; j++) P[j] = P[j - 1] & G[j];
        S[15] = G[15];
#pragma unroll
        for (int i = 14; i >= 0; i--) S[i] = S[i + 1] & G[i];
        S8[7] = G[7];
#pragma unroll
        for (int i = 6; i >= 1; i--) S8[i] = S8[i + 1] & G[i];
        P8[8] = G[8] & kNoHalo;           // ... in every inside window (they all contain byte 8) ...
#pragma unroll
        for (int j = 9; j < 15; j++) P8[j] = P8[j - 1] & G[j];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int a = j - K + 1;   // first byte of the window (negative: in the previous lane)
            if (a < 0) { A[j] = P[j]; B[j] = S[(16 + a) & 15] << 1; }
            else if (a == 0) { A[j] = P[j]; B[j] = kAll; }
            else if (j == 15) { A[j] = S[a & 15]; B[j] = kNoHalo; }     // ... and explicitly in the suffix windows
            else { A[j] = S8[a & 7]; B[j] = P8[j & 15]; }
        }
        return;
    }
    constexpr int P = K >= 8 ? 8 : (K >= 4 ? 4 : (K >= 2 ? 2 : 1));  // largest power of two <= K
    uint64_t E[32];  // E[16 + i] = own byte i, E[i] = previous lane's byte i (a lane shift is a 1-bit shift of the mask)
#pragma unroll
    for (int i = 0; i < 16; i++) { E[16 + i] = G[i]; E[i] = G[i] << 1; }
    E[16] &= kNoHalo;  // cleared in own byte 0 (part of the windows ending at j <= K-1) ...
#pragma unroll
    for (int w = 1; w < P; w *= 2) {   // E[e] becomes the AND of the 2w entries ending at e
#pragma unroll
        for (int e = 31; e >= 2 * w - 1; e--) E[e] &= E[e - w];
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        // ... and explicitly in the windows that do not contain own byte 0 (j >= K)
        if constexpr (K > P) {
            A[j] = E[16 + j];
            B[j] = j >= K ? (E[16 + j - (K - P)] & kNoHalo) : E[16 + j - (K - P)];
        } else {
            A[j] = E[16 + j];
            B[j] = j >= K ? kNoHalo : kAll;
        }
    }
}

// Tile geometry of the sv2 kernels as a function of the bytes a window needs (KM = K, or K + W - 1 for the fused minimizers): up to 32 bytes
// reach at most into the lane before the previous one - lanes 0 / 1 are halo, a tile advances by 62 lanes; 33 .. 48 bytes (fused minimizers
// such as (23, 11)) reach one lane further: three halo lanes, 61 emitting ones.
template <int KM> struct Sv2Geom {
    static_assert(KM >= 1 && KM <= 48, "window of at most 48 bytes");
    static constexpr int kHalo = KM <= 32 ? 2 : 3;
    static constexpr int kSlots = 64 - kHalo;
#ifdef NTK_X_EXACTHALO   // kbench experiment (profiles/r06q): a tile advances by all the bytes whose windows lie inside it, 1024 - (k - 1) rounded down to a dword
    static constexpr int kStride = (KM >= 17 && KM <= 32) ? ((1024 - (KM - 1)) & ~3) : kSlots * 16;
#else
    static constexpr int kStride = kSlots * 16;
#endif
    static constexpr int kHaloBytes = kHalo * 16;
    static constexpr uint64_t kKeep = ~((1ull << kHalo) - 1ull);   // lanes that emit
};

// The same for a window length known only at RUN TIME, any L >= 1 (written for the generic fused minimizer kernel, L = k + w - 1 <= 79; that
// kernel does not use it - see minimizer_invalid16 - and it stays as the tested run-time form of the algebra).  L >= 17: with
// L - 2 = 16 q + C the window ending at own byte j needs the lane's own bytes 0 .. j (A[j], the prefix), q whole lanes before it and the
// last C + 1 - j bytes of the lane before those (j <= C) - or q - 1 whole lanes and the last 17 + C - j bytes (j > C).  C indexes registers and
// is a template constant (a 16-way switch on it), q is a shift count; a lane shift is a 1-bit shift of a mask.  keep = the lanes that may emit.
template <int C>
NTK_HD void window_masks_span(const uint64_t (&G)[16], uint64_t (&A)[16], uint64_t (&B)[16], uint32_t q, uint64_t keep)
{
    uint64_t S[16];
    S[15] = G[15];
#pragma unroll
    for (int i = 14; i >= 0; i--) S[i] = S[i + 1] & G[i];
    uint64_t Fq1 = ~0ull;                                   // the q - 1 lanes before this one are whole
    for (uint32_t i = 1; i < q; i++) Fq1 &= S[0] << i;      // (wave-uniform: q <= 4)
    const uint64_t Fq = q ? (Fq1 & (S[0] << q)) : ~0ull;    // ... the q lanes
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j <= C) B[j] = (S[(15 - C + j) & 15] << (q + 1)) & Fq;
        else B[j] = (S[(j - C - 1) & 15] << q) & Fq1;
    }
    A[0] = G[0] & keep;
#pragma unroll
    for (int j = 1; j < 16; j++) A[j] = A[j - 1] & G[j];
}
// OK[j] = A[j] & B[j] for any compile-time window length 1 .. 48 (Sv2Geom<KM> says which lanes are halo)
template <int KM>
NTK_HD void window_masks_ab_any(const uint64_t (&G)[16], uint64_t (&A)[16], uint64_t (&B)[16])
{
    if constexpr (KM > 32) window_masks_span<(KM - 2) & 15>(G, A, B, (uint32_t)((KM - 2) >> 4), Sv2Geom<KM>::kKeep);
    else if constexpr (KM >= 17) window_masks_ab<KM>(G, A, B);
    else window_masks1_ab<KM>(G, A, B);
}
NTK_HD void window_masks_runtime(const uint64_t (&G)[16], uint64_t (&A)[16], uint64_t (&B)[16], uint32_t L, uint64_t keep)
{
    if (L >= 17) {
        const uint32_t q = (L - 2) >> 4;
        switch ((L - 2) & 15) {
#define NTK_WM_CASE(C) case C: window_masks_span<C>(G, A, B, q, keep); break;
            NTK_WM_CASE(0) NTK_WM_CASE(1) NTK_WM_CASE(2) NTK_WM_CASE(3) NTK_WM_CASE(4) NTK_WM_CASE(5) NTK_WM_CASE(6) NTK_WM_CASE(7)
            NTK_WM_CASE(8) NTK_WM_CASE(9) NTK_WM_CASE(10) NTK_WM_CASE(11) NTK_WM_CASE(12) NTK_WM_CASE(13) NTK_WM_CASE(14) NTK_WM_CASE(15)
#undef NTK_WM_CASE
        }
        return;
    }
    switch (L) {
#define NTK_WM_CASE(K) case K: window_masks1_ab<K>(G, A, B, keep); break;
        NTK_WM_CASE(1) NTK_WM_CASE(2) NTK_WM_CASE(3) NTK_WM_CASE(4) NTK_WM_CASE(5) NTK_WM_CASE(6) NTK_WM_CASE(7) NTK_WM_CASE(8)
        NTK_WM_CASE(9) NTK_WM_CASE(10) NTK_WM_CASE(11) NTK_WM_CASE(12) NTK_WM_CASE(13) NTK_WM_CASE(14) NTK_WM_CASE(15) NTK_WM_CASE(16)
#undef NTK_WM_CASE
    }
}


// ---------------------------------------------------------------------------------------------
// "sv2": the scalar-validity reduce path (17 <= K <= 32, canonical).
//
// * No byte transpose: the four dwords are used as loaded (byte b of dword i = base 4i + b).  Two 8-entry LUTs
//   (v_perm_b32, selector = the byte's low 3 bits) give the only letter with those bits and twice its 2-bit code; the
//   "is a base" test is an SDWA byte compare of that letter with the case-folded byte, and the four codes of a dword
//   are gathered by ONE v_dot4_u32_u8 (weights 64, 16, 4, 1).
// * Strand choice on 32 bits.  With T = the TOP 32 bits (first 16 bases) of a 2K-bit value, K <= 32:
//       T_fwd != T_rc  ->  the values order like their T words (T is the most significant part);
//       T_fwd == T_rc  ->  base i = complement of base K-1-i for i = 0..15, and (i <-> K-1-i is the same condition)
//                          therefore for every i as long as K <= 32: the k-mer is its own reverse complement, the two
//                          values are EQUAL, and `<` / `<=` on T gives what it gives on the values.
//   So one v_cmp_*_u32 decides, no 64-bit register pairs are built (reference src/kmer.rs:124-128, src/bitkmer.rs:136-143).
// * Window words are shared: fw[g] = the 32 bits of the forward stream ENDING at base g, rw[g] = the 32 bits of the
//   reverse-complement stream whose TOP group is base g (g = -D .. 15, D = K - 16; g < 0 = the previous lane's word, one
//   DPP move).  The window ending at base j has  fwd (T, lo) = (fw[j-D], fw[j]),  rc (T, lo) = (rw[j], rw[j-D]):
//   2 (16 + D) / 16 half-rate ops per position instead of 4.
// * Everything that does not depend on validity runs under the full exec mask for four positions at a time; the side
//   effects of those four positions are then applied in ONE masked region that moves the validity mask into exec per
//   position and restores exec once (MP::emit4; on the device a single asm block).
// ---------------------------------------------------------------------------------------------
struct EncSV2 {
    uint32_t code, rcode;
    uint32_t ex[4];  // expected letter per byte (byte b of word i = base 4i + b)
    uint32_t uu[4];  // the byte as read, case-folded
};

template <bool ACCEPT_U>
NTK_HD EncSV2 encode16_sv2(Raw16 d)
{
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    EncSV2 r;
    uint32_t p[4];
    // 8-entry LUTs (v_perm_b32, selector = the byte's low 3 bits):
    //   letter: 1 -> A, 3 -> C, 4 -> T, 5 -> U (normalize pipeline only, reference src/sequence.rs:30), 7 -> G; 0xFF elsewhere.
    //           A byte is a base iff its case-folded value EQUALS the letter its low bits select (SDWA compare).
    //   code  : the letter's 2-bit code A0 C1 G2 T3 (reference src/bitkmer.rs:8-15); 0 elsewhere (every window over a
    //           non-letter is dropped, and 0 cannot spill into the neighbouring bases' bits in the dot product)
    constexpr uint32_t kLutLo = 0x43FF41FFu, kLutHi = ACCEPT_U ? 0x47FF5554u : 0x47FFFF54u;
    constexpr uint32_t kCodeLo = 0x01000000u, kCodeHi = ACCEPT_U ? 0x02000303u : 0x02000003u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t n = w[i] & 0x07070707u;
        r.ex[i] = perm(kLutHi, kLutLo, n);
        p[i] = dot4(perm(kCodeHi, kCodeLo, n), 0x01041040u, 0u);   // c0 c1 c2 c3 as one byte (weights 64, 16, 4, 1)
        r.uu[i] = w[i] & 0xDFDFDFDFu;
    }
    // the four bytes into one word, first base on top: three v_perm_b32 (as shifts and ors the compiler makes it five ops; inline asm
    // may not consume a v_dot4 result: gfx950 needs wait states between a DOT write and a VALU read that the compiler only inserts
    // for instructions it can see - measured: wrong, run-to-run different results)
    r.code = perm(perm(p[0], p[1], 0x0C0C0400u), perm(p[2], p[3], 0x0C0C0400u), 0x05040100u);
    const uint32_t t = brev32(r.code);
    r.rcode = bitop3<0x35>(0x55555555u, t >> 1, add_self(t));  // complement, pairs swapped back after the bit reversal
    return r;
}

NTK_HD bool sv2_base_is_break(const EncSV2 &e, int i)  // host side of the SDWA compare: base i = byte i%4 of word i/4
{
    const int sh = 8 * (i & 3);
    return ((e.ex[i >> 2] >> sh) & 0xFFu) != ((e.uu[i >> 2] >> sh) & 0xFFu);
}

// LIGHT (2K - HB <= 32: K <= 22 with a 12-bit histogram, K <= 23 with the shipped 14-bit one): the top HB bits of a value (its
// histogram cell) reach down to bit 2K-HB <= 32, so cell and lo word together cover every bit of the value.  The per-position
// work then only touches the lo word (sum of lo words in 64 bits, xor of lo words); the high part of both digests follows from
// the block's histogram when it is written out:
//     sum of hi words = sum_b (b >> (32 + HB - 2K)) * H[b]          xor, bits 2K-HB and up = xor_b (H[b] odd ? b : 0)
// and the histogram address only needs the top 16 bits of the chosen T word = min of the two top halves (one packed min).
template <int K, int HB> struct Sv2Light { static constexpr bool value = K >= 17 && 2 * K - HB <= 32; };

template <bool TIE_RC, int K, class Sink, class XL, class MP>
NTK_HD void lane_tile_sv2(Sink &sink, XL &xl, MP &mp, uint32_t code, uint32_t rcode)
{
    static_assert(K >= 17 && K <= 32, "sv2 is the 64-bit-value path");
    constexpr int D = K - 16;
    constexpr bool LIGHT = MP::kLight;
    uint32_t fw[16 + D], rw[16 + D];   // index g + D
    const uint32_t c1 = xl.prev(kSlotCode, code), r1 = xl.prev(kSlotRcode, rcode);
    fw[D + 15] = code; rw[D + 15] = rcode;
    fw[D - 1] = c1;    rw[D - 1] = r1;
#pragma unroll
    for (int j = 0; j < 15; j++) {
        fw[D + j] = alignbit(c1, code, 30 - 2 * j);
        rw[D + j] = alignbit(rcode, r1, 2 * j + 2);
    }
    // The previous lane's forward words 16-g (its window ending g bases before our base 0) are T words here: compared and
    // min-ed, so they are fetched once (DPP move).  Its reverse-complement words are only ever the lo candidate of one
    // position and are fetched where that position is handled.
#pragma unroll
    for (int g = 2; g <= D; g++) {
#ifdef NTK_ABL_HALFIMPORTS   // kbench what-if (WRONG results): every other cross-lane word taken from the lane's own register - the most a longer lane pitch could save
        if (g & 1) { fw[D - g] = fw[D + 16 - g]; continue; }
#endif
        fw[D - g] = xl.prev(kSlotFw + 16 - g, fw[D + 16 - g]);
    }
    // Positions are taken in groups {jp, jp+1, jp+8, jp+9}: the T words of positions j and j+8 sit in ONE register on
    // either strand - fw[j-D] = (top half of T_fwd(j) : top half of T_fwd(j+8)), rw[j+8] = (top half of T_rc(j+8) : top half
    // of T_rc(j)) - so in the LIGHT build one packed 16-bit min with crossed halves yields both histogram prefixes
    // (min of the halves == half of the min, whatever the tie rule).
    // The strand compare and the select of the lo word happen inside the group's masked region (MP::emit_canon / emit_canon_wide).
#pragma unroll
    for (int jp = 0; jp < 8; jp += 2) {
        const int pos[4] = {jp, jp + 1, jp + 8, jp + 9};
        uint32_t ft[4], rt[4], fl[4], rl[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = pos[i];
            ft[i] = fw[j]; rt[i] = rw[D + j];                   // T words: fw[(j - D) + D], rw[j + D]
            fl[i] = fw[D + j];                                  // lo words: forward = the word ending at base j,
#ifdef NTK_ABL_HALFIMPORTS
            if (j - D < -1 && (j & 1)) { rl[i] = rw[16 + j]; continue; }
#endif
            rl[i] = j - D >= -1 ? rw[j] : xl.prev(kSlotRw + 16 + j - D, rw[16 + j]);   // reverse complement = rw[(j - D) + D]: own word, r1, or the previous lane's word 16 + (j - D)
        }
        if constexpr (LIGHT)
            mp.template emit_canon<TIE_RC>(sink, pos, ft, rt, fl, rl, mp.pk_min16_crossed(fw[pos[0]], rw[D + pos[2]]),   // bits 31:16 -> position jp, bits 15:0 -> jp + 8
                                           mp.pk_min16_crossed(fw[pos[1]], rw[D + pos[3]]));
        else
            mp.template emit_canon_wide<TIE_RC>(sink, pos, ft, rt, fl, rl);
    }
}

// sv2 for k <= 16 ("word" builds: a value is one 32-bit word).  Values are kept LEFT-ALIGNED (the k-mer in the top 2K bits,
// zeros below): forward = the window ending 16 - K bases after j, reverse complement = rw[j], each with the low bits cleared (one AND); left-aligned words order
// like the values, their top 14 bits are the histogram cell whatever K is (K < 7: the cell index is the value shifted up), and
// the digests are accumulated left-aligned and shifted down once per block (sum of < 2^32 words of < 2^32: no overflow).
// No cross-lane words beyond the two code words (a window never reaches past the previous lane).  FWD: forward-only builds.
// Odd K, canonical (LAZY): a k-mer of odd length is never its own reverse complement (its middle base would have to be its own
// complement), so the top 2K bits of the two candidates always differ and decide compare and minimum whatever sits below them:
// the candidates go into the region UNMASKED and only the chosen word is masked (one AND per position instead of two).
template <bool TIE_RC, int K, bool FWD, class Sink, class XL, class MP>
NTK_HD void lane_tile_sv2w(Sink &sink, XL &xl, MP &mp, uint32_t code, uint32_t rcode)
{
    static_assert(K >= 1 && K <= 16, "word builds");
    constexpr int S = 32 - 2 * K;
    constexpr bool LAZY = (K & 1) && !FWD;
    constexpr uint32_t vmask = K == 16 ? 0xFFFFFFFFu : ~((1u << (S & 31)) - 1u);   // the value's bits
    constexpr uint32_t hmask = LAZY ? 0xFFFFFFFFu : vmask;                           // applied to each candidate
    const uint32_t c1 = xl.prev(kSlotCode, code);
    uint32_t r1 = 0;
    if (!FWD) r1 = xl.prev(kSlotRcode, rcode);
#pragma unroll
    for (int jp = 0; jp < 8; jp += 2) {
        const int pos[4] = {jp, jp + 1, jp + 8, jp + 9};
        uint32_t f[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = pos[i];
            // forward: the K bases ending at base j on top, zeros below = the stream window that ends S / 2 bases later with its low S
            // bits cleared (one shift-class op and a full-rate AND; shifting the window ending at j up would be two half-rate ops:
            // v_lshlrev_b32 issues at half rate on gfx950 even with a constant shift, v_lshrrev_b32 at full rate - tools/ubench.hip)
            const int sh = 30 - 2 * j - S;   // bits of `code` below the wanted window
            f[i] = j == 15 ? (S ? code << S : code) : ((sh > 0 ? alignbit(c1, code, sh & 31) : (sh == 0 ? code : code << ((-sh) & 31))) & hmask);
            r[i] = FWD ? 0u : ((j == 15 ? rcode : alignbit(rcode, r1, 2 * j + 2)) & hmask);
        }
        if constexpr (FWD) mp.emit_word_fwd(sink, pos, f);
        else mp.template emit_word<TIE_RC, (LAZY ? vmask : 0xFFFFFFFFu)>(sink, pos, f, r);   // the chosen value is min(f, r) [& vmask]; the compare (ties: TIE_RC) only feeds the strand count
    }
}

// Forward-only sv2 (BitNuclKmer with canonical = false, reference src/bitkmer.rs:80-108 and Sequence::bit_kmers(k, false),
// src/sequence.rs:250-252): no reverse-complement stream, no strand compare and no strand counter - the T word of position j is
// fw[j - D], its lo word fw[j]; every k-mer counts as forward (n_fwd = n_total at block end).
template <int K, class Sink, class XL, class MP>
NTK_HD void lane_tile_sv2_fwd(Sink &sink, XL &xl, MP &mp, uint32_t code)
{
    static_assert(K >= 17 && K <= 32, "sv2 is the 64-bit-value path");
    constexpr int D = K - 16;
    constexpr bool LIGHT = MP::kLight;
    uint32_t fw[16 + D];   // index g + D
    const uint32_t c1 = xl.prev(kSlotCode, code);
    fw[D + 15] = code;
    fw[D - 1] = c1;
#pragma unroll
    for (int j = 0; j < 15; j++) fw[D + j] = alignbit(c1, code, 30 - 2 * j);
#pragma unroll
    for (int g = 2; g <= D; g++) fw[D - g] = xl.prev(kSlotFw + 16 - g, fw[D + 16 - g]);
#pragma unroll
    for (int jp = 0; jp < 8; jp += 2) {
        const int pos[4] = {jp, jp + 1, jp + 8, jp + 9};
        uint32_t T[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // LIGHT: fw[j - D] carries the histogram prefix of position j in bits 31:16 and that of position j + 8 in bits 15:0
            T[i] = LIGHT ? fw[pos[i & 1]] : fw[pos[i]];
            lo[i] = fw[D + pos[i]];
        }
        mp.emit_fwd(sink, pos, T, lo);
    }
}

// ---------------------------------------------------------------------------------------------
// "sv2" windowed minimizers, fused into the scan (BASELINE.json configs[4]; SURVEY.md A.7: sequence::minimizer, reference
// src/sequence.rs:139-152, applied to every window of W+K-1 good bases).  The window ending at byte e holds the W k-mers
// ending at e-W+1 .. e; its minimizer is the smallest of their canonical values, the LEFTMOST one on ties, reported with
// that k-mer's strand flag.  Nothing is written to HBM: the canonical values live in registers as 64-bit KEYS
//     key = (value << 6) | (index << 1) | strand bit      (< 2^52: as a double, positive with a zero exponent field - a denormal)
// so that one unsigned 64-bit minimum does everything at once - on the device v_min_f64: positive doubles, denormal ones included, order
// like their bit patterns (rounds 2 - 4 set bit 62 to make the keys normal doubles; not needed, see key_hi below):
//   * strand choice: min(forward key, reverse-complement key); the strand bit is arranged so that the tie goes where the
//     reference's iterator sends it (TIE_RC: rc carries 0, reference src/kmer.rs:124-128; else forward carries 0, bitkmer.rs:138-142);
//   * window minimum with the leftmost tie rule: the index (own position j -> 16 + j; a key imported from the previous
//     lane has 16 subtracted, so it is older than every own key) breaks value ties towards the older k-mer.
// Validity is the plain window rule with K + W - 1 in place of K (mask algebra unchanged).  The window minimum uses the
// prefix / suffix (van Herk) decomposition around the lane boundary and around own position W-1, which needs 2W - 3 >= 15.
// The digests follow the LIGHT scheme of lane_tile_sv2 (K <= 22): lo word per position, high parts from the histogram.
// ---------------------------------------------------------------------------------------------
template <int K, int W> struct Sv2MinFused {
    static constexpr bool value = K >= 15 && K <= 23 && W >= 2 && W <= 16 && K + W - 1 <= 48;   // (W <= 16: the keys of a window reach one lane back; K + W - 1 > 32: three halo lanes, Sv2Geom)
};

template <bool TIE_RC, int K, int W, class Sink, class XL, class MP>
NTK_HD void lane_tile_sv2_min(Sink &sink, XL &xl, MP &mp, uint32_t code, uint32_t rcode)
{
    static_assert(Sv2MinFused<K, W>::value, "fused minimizers: 15 <= K <= 23, 2 <= W <= 16, K + W - 1 <= 48");
    constexpr int D = K > 16 ? K - 16 : 0;
    constexpr int HS = K > 16 ? 58 - 2 * K : 26;           // key hi word = T >> HS (| bit 30); K <= 16: the value is one word, T = value
    // The key's high word = T >> HS: fewer than 2^20, so the key's exponent field is zero and the key is a DENORMAL double - which v_min_f64
    // orders like any other positive double (f64 denormals are never flushed in the mode HIP kernels run in), so no marker bit has to make it a
    // normal one: a full-rate shift instead of the funnel shift that merged the marker in (32 of them per tile; NTK_MINKEY_MARKER=1 restores it).
#ifndef NTK_MINKEY_MARKER
#define NTK_MINKEY_MARKER 0
#endif
    constexpr uint32_t kBit62 = 1u << (HS - 2);            // alignbit(kBit62, T, HS) == (T >> HS) | 0x40000000
    auto key_hi = [](uint32_t t) -> uint32_t { return NTK_MINKEY_MARKER ? alignbit(kBit62, t, HS) : (t >> HS); };
    constexpr uint32_t fbitF = TIE_RC ? 1u : 0u, fbitR = TIE_RC ? 0u : 1u;
    uint32_t fw[16 + D + 1], rw[16 + D + 1];   // index g + D  (+ 1: never a zero-length array)
    const uint32_t c1 = xl.prev(kSlotCode, code), r1 = xl.prev(kSlotRcode, rcode);
    fw[D + 15] = code; rw[D + 15] = rcode;
    if constexpr (D > 0) { fw[D - 1] = c1; rw[D - 1] = r1; }
#pragma unroll
    for (int j = 0; j < 15; j++) {
        fw[D + j] = alignbit(c1, code, 30 - 2 * j);
        rw[D + j] = alignbit(rcode, r1, 2 * j + 2);
    }
#pragma unroll
    for (int g = 2; g <= D; g++) {
        fw[D - g] = xl.prev(kSlotFw + 16 - g, fw[D + 16 - g]);
        rw[D - g] = xl.prev(kSlotRw + 16 - g, rw[D + 16 - g]);
    }
    // canonical keys of the 16 own positions.  Low word = (lo word << 6) | tag: shifting a stream window up by three bases is the window
    // that ends three bases LATER (forward) / starts three bases EARLIER (reverse complement) with its low 6 bits replaced - one
    // full-rate v_bitop3 on a word that is in a register anyway, where the shift-and-or is a half-rate op (13 of the 16 positions
    // per strand: the other three would need the next lane's / an unfetched word).
    uint64_t key[16];
    constexpr uint32_t kTagMask = 0xFFFFFFC0u;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t idx2 = (uint32_t)(16 + j) << 1;
        if constexpr (K > 16) {
            const uint32_t lf = j <= 12 ? and_or(fw[D + (j <= 12 ? j + 3 : j)], kTagMask, idx2 | fbitF) : ((fw[D + j] << 6) | (idx2 | fbitF));
            const uint32_t lr = j >= 3 ? and_or(rw[j >= 3 ? j - 3 : j], kTagMask, idx2 | fbitR) : ((rw[j] << 6) | (idx2 | fbitR));
            const uint64_t kf = ((uint64_t)key_hi(fw[j]) << 32) | lf;
            const uint64_t kr = ((uint64_t)key_hi(rw[D + j]) << 32) | lr;
            key[j] = mp.min64(kf, kr);
        } else {
            // K <= 16 (the common (15, 10) sketch): the value is the low 2K bits of the forward word ending at base j / the top 2K
            // bits of the reverse-complement word starting there; the key is built from that one word
            constexpr uint32_t vmask = K == 16 ? 0xFFFFFFFFu : ((1u << ((2 * K) & 31)) - 1u);
            constexpr int G = K - 13;   // (value << 6) of the right-aligned reverse-complement value = the word starting G bases earlier, low 6 bits cleared
            const uint32_t vf = K == 16 ? fw[j] : (fw[j] & vmask), vr = K == 16 ? rw[j] : (rw[j] >> ((32 - 2 * K) & 31));
            const uint32_t lf = j <= 12 ? and_or(fw[j <= 12 ? j + 3 : j], vmask << 6, idx2 | fbitF) : ((vf << 6) | (idx2 | fbitF));
            const uint32_t lr = j >= G ? and_or(rw[j >= G ? j - G : j], kTagMask, idx2 | fbitR) : ((vr << 6) | (idx2 | fbitR));
            const uint64_t kf = ((uint64_t)key_hi(vf) << 32) | lf;
            const uint64_t kr = ((uint64_t)key_hi(vr) << 32) | lr;
            key[j] = mp.min64(kf, kr);
        }
    }
    uint64_t win[16];
    if constexpr (W <= 8) {
        // short windows (the w = 5 of several sketching presets): the W - 1 keys before own position 0 are the previous lane's last ones (index
        // - 16: older than every own key), then a sliding minimum by doubling over the 16 + W - 1 keys: M_2q[i] = min(M_q[i], M_q[i + q]) while
        // 2q <= W, and two overlapping windows of q make W (3 - 3.5 minima per position; the prefix / suffix scheme below needs 2W - 3 >= 15)
        constexpr int E = W - 1, N = 16 + E;
        uint64_t M[N];
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int a = 16 - E + i;
            const uint32_t lo = xl.prev_add(kSlotSufLo + a, (uint32_t)key[a], 0u - 32u);
            const uint32_t hi = xl.prev(kSlotSufHi + a, (uint32_t)(key[a] >> 32));
            M[i] = ((uint64_t)hi << 32) | lo;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) M[E + j] = key[j];
        constexpr int Q = W >= 8 ? 8 : (W >= 4 ? 4 : (W >= 2 ? 2 : 1));   // the largest power of two <= W
#pragma unroll
        for (int q = 1; q < Q; q *= 2)
#pragma unroll
            for (int i = 0; i + q < N; i++) M[i] = mp.min64(M[i], M[i + q]);   // (M[i + q] is still the previous round's: i ascends)
#pragma unroll
        for (int j = 0; j < 16; j++) win[j] = W == Q ? M[j] : mp.min64(M[j], M[j + W - Q]);
    } else {
    // suffix minima of the own keys, handed to the next lane; the previous lane's arrive with 16 taken off their index
    constexpr int A0 = 17 - W;                             // the previous lane's positions A0 .. 15 can be in a window of ours
    uint64_t suf[16], imp[16];
    suf[15] = key[15];
#pragma unroll
    for (int a = 14; a >= A0; a--) suf[a] = mp.min64(key[a], suf[a + 1]);
#pragma unroll
    for (int a = A0; a < 16; a++) {
        const uint32_t lo = xl.prev_add(kSlotSufLo + a, (uint32_t)suf[a], 0u - 32u);       // low word: index -= 16
        const uint32_t hi = xl.prev(kSlotSufHi + a, (uint32_t)(suf[a] >> 32));
        imp[a] = ((uint64_t)hi << 32) | lo;
    }
    // windows reaching into the previous lane: j = 0 .. W-2
    uint64_t pre = key[0];
#pragma unroll
    for (int j = 0; j <= W - 2; j++) {
        if (j) pre = mp.min64(pre, key[j]);
        win[j] = mp.min64(imp[A0 + j], pre);
    }
    // own-lane windows j = W-1 .. 15: [j-W+1, W-2] (suffix ending at W-2) and [W-1, j] (prefix from W-1); 2W-3 >= 15
    uint64_t sfx[16];
    sfx[W - 2] = key[W - 2];
#pragma unroll
    for (int a = W - 3; a >= 0; a--) sfx[a] = mp.min64(key[a], sfx[a + 1]);
    uint64_t pfx = key[W - 1];
#pragma unroll
    for (int j = W - 1; j < 16; j++) {
        if (j > W - 1) pfx = mp.min64(pfx, key[j]);
        win[j] = j - W + 1 <= W - 2 ? mp.min64(sfx[j - W + 1], pfx) : pfx;
    }
    }
#pragma unroll
    for (int jb = 0; jb < 16; jb += 4) {
        const int pos[4] = {jb, jb + 1, jb + 2, jb + 3};
        uint32_t lo[4], cell4[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t mh = (uint32_t)(win[jb + i] >> 32), ml = (uint32_t)win[jb + i];
            lo[i] = alignbit(mh, ml, 6);                                   // low 32 bits of the value
            // the value's top 14 bits = key bits [2K-8, 2K+6); times four (the byte offset of a 14-bit histogram cell) = key >> (2K-10), low two bits cleared
            cell4[i] = (2 * K - 10 >= 32 ? mh >> ((2 * K - 42) & 31) : alignbit(mh, ml, (2 * K - 10) & 31)) & 0xFFFCu;
            fb[i] = ml & 1u;
        }
        mp.emit_min4(sink, pos, cell4, lo, fb);
    }
}

// One lane of one tile of the generic fused minimizer kernel: sink.begin(invw) with invw bit 15 - j set = the window ending at own byte j is
// not emitted, then sink.emit4(jb, keys) for jb = 0, 4, 8, 12 with the minimizers' keys of the windows ending at bytes jb .. jb + 3.  Lanes 0
// and 1 are the k-mer halo, lanes below a.min_halo_lanes hold k-mers that the first emitting lanes' windows need.
template <bool F64> struct MinKey { typedef KeyG type; };
template <> struct MinKey<true> { typedef KeyF type; };
template <int KW, bool TIE_RC, bool ACCEPT_U, bool F64, class XL, class Sink>
NTK_HD void minimizer_lane(const ScanArgs &a, XL &xl, Sink &sink, Raw16 raw, int64_t lane_base, uint32_t lane, bool tail_tile)
{
    const EncSV2 en = encode16_sv2<ACCEPT_U>(raw);
    typename MinKey<F64>::type M[16];
    if constexpr (F64) minimizer_keys_f64<TIE_RC>(a, xl, en.code, en.rcode, lane, M);
    else minimizer_keys_general<KW, TIE_RC>(a, xl, en.code, en.rcode, M);
    sink.begin(minimizer_invalid16(a, xl, bad16_from_letters(en.ex, en.uu), lane_base, lane, tail_tile));
    minimizer_slide(a, xl, M, sink);
}

// ---------------------------------------------------------------------------------------------
// CanonicalKmers with 33 <= k <= 255 on the reduce face (wide_canonical_reduce_kernel in ntk_kernels.hpp; tests/emu runs the same functions
// on the host): a block stages kWkThreads slots of 16 window ENDS + kWkHaloSlots slots before them (k - 1 <= 254 bytes) as code word,
// reverse-complement word and break mask per slot.  Positions are "staged": 16 x slot + byte.
// ---------------------------------------------------------------------------------------------
constexpr int kWkThreads = 256, kWkTile = kWkThreads * 16, kWkHaloSlots = 16, kWkSlots = kWkThreads + kWkHaloSlots;

// OR of the first `keep` bytes of a 16-byte line (keep <= 0: none, >= 16: all): what the speculative kernels watch for bit 5 - the padding
// behind the input's last byte is nobody's base and must not send a launch to the byte-walking kernel.
NTK_HD uint32_t or_of_input_bytes(Raw16 raw, int64_t keep)
{
    if (keep >= 16) return raw.x | raw.y | raw.z | raw.w;
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t r = 0;
    for (int d = 0; d < 4; d++) {
        const int64_t nb = keep - 4 * d;
        if (nb > 0) r |= nb >= 4 ? w[d] : (w[d] & ((1u << (8 * (uint32_t)nb)) - 1u));
    }
    return r;
}

// One slot: `keep` = how many of its 16 bytes belong to the input (<= 0: none - before the input's start or beyond its end; >= 16: all).
// Returns the staged position of the slot's last break, -1 if it holds none.
struct WkSlot { uint32_t code, rcode, bad, or_bytes; int32_t last_break; };   // bad: base i at bit 15 - i; or_bytes: OR of the slot's input bytes
template <bool ACCEPT_U>
NTK_HD WkSlot wk_stage_slot(Raw16 raw, int32_t slot, int64_t keep)
{
    const EncSV2 en = encode16_sv2<ACCEPT_U>(raw);
    WkSlot r;
    r.code = en.code; r.rcode = en.rcode;
    r.or_bytes = or_of_input_bytes(raw, keep);
    r.bad = bad16_from_letters(en.ex, en.uu);
    if (keep <= 0) r.bad = 0xFFFFu;
    else if (keep < 16) r.bad |= 0xFFFFu >> (uint32_t)keep;
    r.last_break = r.bad ? 16 * slot + 15 - (int32_t)__builtin_ctz(r.bad) : -1;
    return r;
}

// The window ending at byte j of slot `slot` is emitted iff j lies below the slot's own first break and at least k bytes behind the last
// break before the slot (`before`: staged position, -1 if the staged bytes before the slot hold none): position j at bit 15 - j.
NTK_HD uint32_t wk_valid16(uint32_t bad_own, int32_t before, uint32_t k, int32_t slot)
{
    const uint32_t first_bad = bad_own ? (uint32_t)__builtin_clz(bad_own) - 16u : 16u;
    const uint32_t mask_own = first_bad >= 16u ? 0xFFFFu : ((0xFFFFu << (16u - first_bad)) & 0xFFFFu);
    const int32_t thr = before + (int32_t)k - 16 * slot;
    const uint32_t mask_inh = thr <= 0 ? 0xFFFFu : (thr >= 16 ? 0u : (0xFFFFu >> (uint32_t)thr));
    return mask_own & mask_inh;
}

// Where the window ending at byte 0 of a slot starts: `back` code words before the slot's own, `bits` bits below that word's top.
NTK_HD uint32_t wk_back_words(uint32_t k) { return (k - 1u + 15u) >> 4; }
NTK_HD uint32_t wk_base_bits(uint32_t k) { return 2u * ((16u - ((k - 1u) & 15u)) & 15u); }
NTK_HD uint32_t wk_take32(uint32_t hi, uint32_t lo, uint32_t bits)   // 32 bits of the stream (hi : lo) starting `bits` (0..30) after hi's top bit
{
    return bits ? alignbit(hi, lo, 32u - bits) : hi;
}

// The strand of the window ending at byte J: the k-mer's first 16 bases (F1, from the three code words R realigned to the first window's first
// base) against its reverse complement's first 16 (V1: the complement of its last 16, reversed - from the own and the previous slot's
// reverse-complement words); the second 16 only where the first tie.  lt: the forward k-mer is the smaller slice (reference src/kmer.rs:124-128:
// ties report the reverse complement); tie: equal over 32 bases - the caller's launch is redone by the byte-walking kernel; top: the chosen
// strand's first 16 bases (its leading six are the histogram bin).
struct WkWords { uint32_t R0, R1, R2, rc0, rc1, rc2; };
template <int J>
NTK_HD void wk_strand(const WkWords &w, bool &lt, bool &tie, uint32_t &top)
{
    const uint32_t F1 = J ? alignbit(w.R0, w.R1, 32 - 2 * J) : w.R0, V1 = J < 15 ? alignbit(w.rc0, w.rc1, 2 * J + 2) : w.rc0;
    lt = F1 < V1; tie = false;
    if (F1 == V1) {   // 4^-16 per position on random text: the second 16 bases are looked at only here
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");   // (keeps the branch: if-converted, the two funnel shifts and compares run for every position)
#endif
        const uint32_t F2 = J ? alignbit(w.R1, w.R2, 32 - 2 * J) : w.R1, V2 = J < 15 ? alignbit(w.rc1, w.rc2, 2 * J + 2) : w.rc1;
        lt = F2 < V2; tie = F2 == V2;
    }
    top = lt ? F1 : V1;
}

}  // namespace ntk
