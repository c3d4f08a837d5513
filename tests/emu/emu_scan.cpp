// emu_scan.cpp — TEST-ONLY lock-step emulation of one wave64 of the scan kernel on the host.
// It compiles the very same per-lane source the HIP kernels use (needletail_amd/csrc/ntk_tile.hpp:
// encode16, emit_windows) with portable stand-ins for v_perm/v_alignbit/bitreverse, and re-creates the
// cross-lane halo exchange (DPP wave_shr/wave_ror in the kernel) with plain arrays.  It lets the CPU
// test-suite check the kernel's bit manipulation against the oracle without a GPU.  It is not part of
// the product library and is never a fallback for it.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../needletail_amd/csrc/ntk_tile.hpp"

using namespace ntk;

namespace {

struct HostStats {
    uint64_t n_total = 0, n_fwd = 0, sum = 0, xr = 0;
    uint64_t hist[kHistBins] = {0};
};

template <int KW>
struct HostSink {
    HostStats *st;
    uint32_t bin_shift;
    uint64_t *values; uint16_t *valid16, *rc16; uint64_t n_bytes = 0;
    int64_t base = 0; uint32_t inval = 0, rcbits = 0; bool skip = true;
    void begin_tile(int64_t lane_base, uint32_t inval16, bool halo) { base = lane_base; inval = inval16; rcbits = 0; skip = halo || lane_base >= (int64_t)n_bytes; }
    void emit(int j, bool valid, bool take_fwd, uint32_t hi, uint32_t lo)
    {
        const uint64_t v = KW == 2 ? (((uint64_t)hi << 32) | lo) : (uint64_t)lo;
        if (valid) {
            st->n_total++; st->n_fwd += take_fwd; st->sum += v; st->xr ^= v; st->hist[v >> bin_shift]++;
        }
        if (values && !skip) values[base + j] = v;
        rcbits |= (take_fwd ? 0u : 1u) << (15 - j);
    }
    void end_tile()
    {
        if (skip) return;
        const uint32_t v = ~inval & 0xFFFFu;
        if (valid16) valid16[base >> 4] = (uint16_t)v;
        if (rc16) rc16[base >> 4] = (uint16_t)(rcbits & v);
    }
};

// Host stand-in for DPP wave_shr:1: lanes are emulated in increasing order, so "the previous lane's value in the
// same register" is simply what the previous lane stored in that slot (lane 0 reads 0, like bound_ctrl:0).
struct EmuXL {
    uint32_t last[kNumSlots];   // values written by the previous lane
    uint32_t cur[kNumSlots];    // values written by the current lane
    // prev_auto: call sites without a slot id are numbered in the order a lane reaches them (every lane of a wave runs the same sequence:
    // the branches around them are wave-uniform)
    std::vector<uint32_t> last_auto, cur_auto;
    size_t n_auto = 0;
    bool first_lane = true;
    void next_lane(bool first)
    {
        if (!first) memcpy(last, cur, sizeof(last)); else memset(last, 0, sizeof(last));
        if (!first) last_auto = cur_auto; else last_auto.clear();
        cur_auto.clear(); n_auto = 0;
        first_lane = first;
    }
    uint32_t prev_auto(uint32_t x)
    {
        cur_auto.push_back(x);
        const uint32_t r = n_auto < last_auto.size() ? last_auto[n_auto] : 0u;   // lane 0 reads 0 (bound_ctrl)
        n_auto++;
        return r;
    }
    uint32_t prev(int slot, uint32_t x) { cur[slot] = x; return last[slot]; }
    uint32_t prev_and(int slot, uint32_t x, uint32_t mask) { return prev(slot, x) & mask; }
    uint32_t select_prev(int slot, bool take, uint32_t a, uint32_t b) { const uint32_t p = prev(slot, b); return take ? a : p; }
    uint32_t prev_add(int slot, uint32_t x, uint32_t c) { return prev(slot, x) + c; }
};

Raw16 load16(const uint8_t *buf, uint64_t n_padded, int64_t off)
{
    uint8_t b[16];
    for (int i = 0; i < 16; i++) b[i] = (off + i >= 0 && (uint64_t)(off + i) < n_padded) ? buf[off + i] : 0;  // buffer bounds check returns 0
    Raw16 r;
    memcpy(&r.x, b, 4); memcpy(&r.y, b + 4, 4); memcpy(&r.z, b + 8, 4); memcpy(&r.w, b + 12, 4);
    return r;
}

// QM builds of the kernel: the quality tile is loaded with the same geometry and folded into the sequence bytes
// (ntk_tile.hpp quality_break16) before anything else looks at them.
const uint8_t *g_qual = nullptr;
QualityCut g_qc = {0, 0};
Raw16 load16q(const uint8_t *buf, uint64_t n_padded, int64_t off)
{
    Raw16 r = load16(buf, n_padded, off);
    if (g_qual) r = quality_break16(r, load16(g_qual, n_padded, off), g_qc.add, g_qc.sel);
    return r;
}

template <int KW, bool CANON, bool TIE_RC, bool ACCEPT_U, int KFIX = 0>
void run(const uint8_t *buf, uint64_t n, uint64_t n_padded, ScanArgs a, HostStats *st, uint64_t *values,
         uint16_t *valid16, uint16_t *rc16)
{
    const uint64_t n_tiles = ((n + 15) / 16 + kTileSlots - 1) / kTileSlots;
    {   // tiles carry no state from their predecessor, so the order in which waves pull them is irrelevant
        const uint64_t t0 = 0, t1 = n_tiles;
        HostSink<KW> sinks[64];
        for (auto &s : sinks) { s.st = st; s.bin_shift = a.bin_shift; s.values = values; s.valid16 = valid16; s.rc16 = rc16; s.n_bytes = n; }
        for (uint64_t t = t0; t < t1; t++) {
            const bool tail = (t + 1) * kTileStride > n;
            EmuXL xl;
            for (int l = 0; l < 64; l++) {
                xl.next_lane(l == 0);
                const int64_t lane_base = (int64_t)(t * kTileStride) - 32 + l * 16;
                lane_tile<KW, CANON, TIE_RC, ACCEPT_U, KFIX>(a, sinks[l], xl, load16q(buf, n_padded, lane_base), lane_base,
                                                       l < kHaloLanes, tail);
            }
        }
    }
}

// "sv2" variant (ntk_tile.hpp lane_tile_sv2): same mask algebra; the digests are kept the way the device keeps them -
// sum of (hi:lo) / xor words of (T, lo), or for K <= 22 ("light") only the lo words plus a per-block histogram from
// which the high parts are rebuilt - so that the reconstruction arithmetic of scan2_kernel is what gets checked.
template <int K, int HB>
struct EmuMP2 {
    static constexpr bool kLight = Sv2Light<K, HB>::value;
    static constexpr int kWordCopies = K <= 6 ? (((1 << HB) >> (2 * (K <= 6 ? K : 0))) < 64 ? ((1 << HB) >> (2 * (K <= 6 ? K : 0))) : 64) : 1;
    uint64_t VA[16], VB[16];   // the window ending at byte j is emitted where VA[j] & VB[j] is set (the region forms exec with that AND)
    int lane = 0;
    uint64_t sum = 0, sumh = 0, n_fwd = 0;
    uint32_t xh = 0, xlo = 0;
    std::vector<uint32_t> cells = std::vector<uint32_t>(1u << HB, 0u);
    bool valid(int j) const { return ((VA[j] & VB[j]) >> lane) & 1; }
    uint32_t pk_min16_crossed(uint32_t a, uint32_t b) const   // v_pk_min_u16 op_sel:[0,1] op_sel_hi:[1,0]
    {
        const uint32_t h = (a >> 16) < (b & 0xFFFFu) ? (a >> 16) : (b & 0xFFFFu), l = (a & 0xFFFFu) < (b >> 16) ? (a & 0xFFFFu) : (b >> 16);
        return (h << 16) | l;
    }
    static uint32_t cell_offset_hi(uint32_t T) { return HB == 14 ? ((T >> 16) & 0xFFFCu) : ((T >> 18) & 0x3FFCu); }
    static uint32_t cell_offset_lo(uint32_t T) { return HB == 14 ? (T & 0xFFFCu) : ((T >> 2) & 0x3FFCu); }
    // the masked regions of DevMasks2 (ntk_kernels.hpp), one lane at a time: strand compare and select under the validity mask
    template <bool TIE_RC_, class S>
    void emit_canon(S &, const int (&pos)[4], const uint32_t (&ft)[4], const uint32_t (&rt)[4], const uint32_t (&fl)[4], const uint32_t (&rl)[4],
                    uint32_t Tm0, uint32_t Tm1)
    {
        const uint32_t off[4] = {cell_offset_hi(Tm0), cell_offset_hi(Tm1), cell_offset_lo(Tm0), cell_offset_lo(Tm1)};
        for (int i = 0; i < 4; i++) {
            if (!valid(pos[i])) continue;
            const bool fwd = TIE_RC_ ? ft[i] < rt[i] : ft[i] <= rt[i];
            const uint32_t t = fwd ? fl[i] : rl[i];
            sum += t; xlo ^= t; cells[off[i] >> 2]++; n_fwd += fwd ? 1 : 0;
        }
    }
    template <bool TIE_RC_, class S>
    void emit_canon_wide(S &, const int (&pos)[4], const uint32_t (&ft)[4], const uint32_t (&rt)[4], const uint32_t (&fl)[4], const uint32_t (&rl)[4])
    {
        constexpr int SH = 64 - 2 * K;
        for (int i = 0; i < 4; i++) {
            const uint32_t T = ft[i] < rt[i] ? ft[i] : rt[i], hi = SH ? T >> (SH & 31) : T, off = cell_offset_hi(T);
            if (!valid(pos[i])) continue;
            const bool fwd = TIE_RC_ ? ft[i] < rt[i] : ft[i] <= rt[i];
            const uint32_t t = fwd ? fl[i] : rl[i];
            sum += t; xlo ^= t; sumh += hi; xh ^= hi; cells[off >> 2]++; n_fwd += fwd ? 1 : 0;
        }
    }
    bool fwd_only = false;
    uint32_t word_cell(uint32_t v) const
    {
        const uint32_t rep = (uint32_t)lane & (uint32_t)(kWordCopies - 1);
        return K <= 6 ? ((v >> ((30 - 2 * K) & 31)) | (rep << ((2 * K + 2) & 31))) : cell_offset_hi(v);
    }
    template <bool TIE_RC_, uint32_t VMASK, class S>
    void emit_word(S &, const int (&pos)[4], const uint32_t (&f)[4], const uint32_t (&r)[4])   // word builds (K <= 16): left-aligned values
    {
        for (int i = 0; i < 4; i++) {
            if (!valid(pos[i])) continue;
            const uint32_t v = (f[i] < r[i] ? f[i] : r[i]) & VMASK;
            cells[word_cell(v) >> 2]++;
            sum += v; xlo ^= v;
            n_fwd += (TIE_RC_ ? f[i] < r[i] : f[i] <= r[i]) ? 1 : 0;
        }
    }
    template <class S>
    void emit_word_fwd(S &, const int (&pos)[4], const uint32_t (&v)[4])
    {
        fwd_only = true;
        for (int i = 0; i < 4; i++) {
            if (!valid(pos[i])) continue;
            cells[word_cell(v[i]) >> 2]++;
            sum += v[i]; xlo ^= v[i];
        }
    }
    template <class S>
    void emit_fwd(S &, const int (&pos)[4], const uint32_t (&T)[4], const uint32_t (&lo)[4])
    {
        constexpr int SH = 64 - 2 * K;
        fwd_only = true;
        for (int i = 0; i < 4; i++) {
            if (!valid(pos[i])) continue;
            const uint32_t off = (kLight && i >= 2) ? cell_offset_lo(T[i]) : cell_offset_hi(T[i]);
            cells[off >> 2]++;
            sum += lo[i]; xlo ^= lo[i];
            if (!kLight) { const uint32_t hi = SH ? T[i] >> (SH & 31) : T[i]; sumh += hi; xh ^= hi; }
        }
    }
    uint64_t min64(uint64_t a, uint64_t b) const { return a < b ? a : b; }   // v_min_f64 on positive normal doubles
    uint64_t nf_bits = 0;
    bool min_mode = false, tie_rc = false;
    template <class S>
    void emit_min4(S &, const int (&pos)[4], const uint32_t (&cell4)[4], const uint32_t (&lo)[4], const uint32_t (&fb)[4])
    {
        min_mode = true;
        for (int i = 0; i < 4; i++) {
            if (!valid(pos[i])) continue;
            const uint32_t off = HB == 14 ? cell4[i] : (cell4[i] >> 4) << 2;
            cells[off >> 2]++;
            sum += lo[i]; xlo ^= lo[i]; nf_bits += fb[i];
        }
    }
    void finish(HostStats *st)   // the block-end arithmetic of scan2_kernel
    {
        constexpr bool WORD = K <= 16, LIGHT = kLight && !WORD;
        uint64_t s = sum + (sumh << 32), xr = (WORD || LIGHT) ? (uint64_t)xlo : ((uint64_t)xh << 32) | xlo, shi = 0, xf = 0, nv = 0;
        const uint32_t per = (1u << HB) / kHistBins;
        for (uint32_t c = 0; c < (uint32_t)kHistBins; c++) {
            uint32_t tot = 0;
            if (K <= 6) {
                if (c < (1u << ((2 * K) & 31))) for (int r = 0; r < kWordCopies; r++) tot += cells[((uint32_t)r << ((2 * K) & 31)) + c];
            } else {
                for (uint32_t q = 0; q < per; q++) {
                    const uint32_t f = c * per + q, h = cells[f];
                    tot += h;
                    if (LIGHT) { shi += (uint64_t)(f >> ((32 + HB - 2 * K) & 31)) * h; xf ^= (h & 1u) ? f : 0u; }
                }
            }
            st->hist[c] += tot; nv += tot;
        }
        if (LIGHT) {
            constexpr uint32_t low_mask = 2 * K - HB >= 32 ? 0xFFFFFFFFu : ((1u << ((2 * K - HB) & 31)) - 1u);
            s += shi << 32;
            xr = (xf << ((2 * K - HB) & 63)) | (uint64_t)(xlo & low_mask);
        }
        if (WORD && K < 16 && !min_mode) { s >>= (32 - 2 * K) & 31; xr >>= (32 - 2 * K) & 31; }   // (the fused minimizers keep plain values)
        if (min_mode) n_fwd = tie_rc ? nf_bits : nv - nf_bits;
        if (fwd_only) n_fwd = nv;   // the forward-only kernel keeps no strand counter: n_fwd = n_total
        st->n_total += nv; st->n_fwd += n_fwd; st->sum += s; st->xr ^= xr;
    }
};

struct EmuNoSink {};

template <bool TIE_RC, bool ACCEPT_U, int K, int HB, int W = 0, bool FWD = false>
void run_sv2(const uint8_t *buf, uint64_t n, uint64_t n_padded, HostStats *st)
{
    using Geo = Sv2Geom<(W ? K + W - 1 : K)>;   // the kernel's tile geometry: 2 halo lanes, or 3 where a window needs more than 32 bytes
    const uint64_t n_tiles = ((n + 15) / 16 + Geo::kSlots - 1) / Geo::kSlots;
    EmuMP2<K, HB> mp;
    EmuNoSink sink;
    for (uint64_t t = 0; t < n_tiles; t++) {
        const bool tail = (t + 1) * Geo::kStride > n;
        EncSV2 en[64];
        uint64_t G[16] = {0};
        for (int l = 0; l < 64; l++) {
            const int64_t lane_base = (int64_t)(t * Geo::kStride) - Geo::kHaloBytes + l * 16;
            en[l] = encode16_sv2<ACCEPT_U>(load16q(buf, n_padded, lane_base));
            for (int i = 0; i < 16; i++) {
                bool good = !sv2_base_is_break(en[l], i);
                if (tail && lane_base + i >= (int64_t)n) good = false;
                if (good) G[i] |= 1ull << l;
            }
        }
        window_masks_ab_any<(W ? K + W - 1 : K)>(G, mp.VA, mp.VB);
        mp.tie_rc = TIE_RC;
        EmuXL xl;
        for (int l = 0; l < 64; l++) {
            xl.next_lane(l == 0);
            mp.lane = l;
            if constexpr (W > 0) lane_tile_sv2_min<TIE_RC, K, W>(sink, xl, mp, en[l].code, en[l].rcode);
            else if constexpr (K <= 16) lane_tile_sv2w<TIE_RC, K, FWD>(sink, xl, mp, en[l].code, en[l].rcode);
            else if constexpr (FWD) lane_tile_sv2_fwd<K>(sink, xl, mp, en[l].code);
            else lane_tile_sv2<TIE_RC, K>(sink, xl, mp, en[l].code, en[l].rcode);
        }
    }
    mp.finish(st);
}

// The generic fused minimizer kernel (minimizer_scan_kernel): the same per-lane source (ntk_tile.hpp minimizer_lane: scan2 encode, keys, window validity,
// doubling + two overlapping windows), the kernel's run-time tile geometry (a.min_halo_lanes non-emitting lanes) and its output stage
// restated on the decoded keys.
template <int KW, bool TIE_RC, bool ACCEPT_U, bool F64>
void run_min_generic(const uint8_t *buf, uint64_t n, uint64_t n_padded, ScanArgs a, HostStats *st)
{
    const uint64_t slots = 64 - a.min_halo_lanes, stride = slots * 16, halo_bytes = a.min_halo_lanes * 16;
    const uint64_t n_tiles = ((n + 15) / 16 + slots - 1) / slots;
    for (uint64_t t = 0; t < n_tiles; t++) {
        const bool tail = (t + 1) * stride > n;
        EmuXL xl;
        for (uint32_t l = 0; l < 64; l++) {
            xl.next_lane(l == 0);
            const int64_t lane_base = (int64_t)(t * stride) - (int64_t)halo_bytes + l * 16;
            struct Sink {
                HostStats *st; uint32_t bin_shift; uint32_t invw = 0;
                void begin(uint32_t inv) { invw = inv; }
                void emit4(int jb, const typename MinKey<F64>::type (&g)[4])
                {
                    for (int i = 0; i < 4; i++) {
                        if ((invw >> (15 - (jb + i))) & 1) continue;
                        uint32_t lo, hi, sbit;
                        key_fields(g[i], lo, hi, sbit);
                        if (F64) hi &= ~(1u << 19);   // the key's marker bit (the kernel takes it out of its sums once per block)
                        const uint64_t v = ((uint64_t)hi << 32) | lo;
                        const bool is_rc = (F64 && TIE_RC) ? !sbit : (sbit != 0);   // f64 keys: the tie-winning strand carries 0
                        st->n_total++; st->n_fwd += !is_rc; st->sum += v; st->xr ^= v; st->hist[v >> bin_shift]++;
                    }
                }
            } sink{st, a.bin_shift};
            minimizer_lane<KW, TIE_RC, ACCEPT_U, F64>(a, xl, sink, load16q(buf, n_padded, lane_base), lane_base, l, tail);
        }
    }
}

}  // namespace

// The window-mask algebra alone: OK[j] = A[j] & B[j] (window_masks_ab / window_masks1_ab: the masks with their last AND left to the
// masked region) for window length k.  Returns -1 on bad k.
template <int K>
static void masks_both(const uint64_t (&G)[16], uint64_t *ok, uint64_t *ab)
{
    uint64_t A[16], B[16];
    window_masks_ab_any<K>(G, A, B);
    for (int j = 0; j < 16; j++) { ok[j] = A[j] & B[j]; ab[j] = A[j] & B[j]; }
}

// CanonicalKmers with 33 <= k <= 255 as wide_canonical_reduce_kernel runs it (ntk_kernels.hpp): the same per-slot functions of ntk_tile.hpp,
// tile by tile, with the block's max-scan of break positions as a running maximum.  out: [n_total, n_fwd, windows equal to their reverse
// complement over 32 bases (the kernel raises its redo flag), a byte with bit 5 set was loaded (ditto, when the input is not normalised),
// hist[4096]].
template <int J>
static void wide_position(const WkWords &ww, uint32_t valid, uint64_t *out)
{
    if (!((valid >> (15 - J)) & 1u)) return;
    bool lt, tie; uint32_t top;
    wk_strand<J>(ww, lt, tie, top);
    out[0]++; out[1] += lt ? 1 : 0; out[2] += tie ? 1 : 0; out[4 + (top >> 20)]++;
}
template <bool ACCEPT_U>
static void wide_reduce(const uint8_t *buf, uint64_t n, uint64_t n_padded, uint32_t k, uint64_t *out)
{
    const uint64_t n_tiles = (n + kWkTile - 1) / kWkTile;
    const uint32_t back = wk_back_words(k), base_bits = wk_base_bits(k);
    std::vector<uint32_t> code(kWkSlots + 2, 0u), rcode(kWkSlots, 0u), bad(kWkSlots, 0u);
    std::vector<int32_t> last(kWkSlots, -1);
    for (uint64_t tile = 0; tile < n_tiles; tile++) {
        const int64_t t0 = (int64_t)(tile * kWkTile) - 16 * kWkHaloSlots;
        for (int32_t s = 0; s < kWkSlots; s++) {
            const int64_t p = t0 + 16 * (int64_t)s;
            uint32_t x[4] = {0u, 0u, 0u, 0u};
            if (p >= 0 && (uint64_t)p + 16 <= n_padded) memcpy(x, buf + p, 16);
            const WkSlot sl = wk_stage_slot<ACCEPT_U>(Raw16{x[0], x[1], x[2], x[3]}, s, p < 0 ? 0 : (int64_t)n - p);
            if (sl.or_bytes & 0x20202020u) out[3] = 1;
            code[s] = sl.code; rcode[s] = sl.rcode; bad[s] = sl.bad; last[s] = sl.last_break;
        }
        int32_t before = -1;
        for (int32_t s = 0; s < kWkSlots; s++) {
            if (s >= kWkHaloSlots) {
                const uint32_t valid = wk_valid16(bad[s], before, k, s);
                const uint32_t i0 = (uint32_t)s - back;
                const WkWords ww = {wk_take32(code[i0], code[i0 + 1], base_bits), wk_take32(code[i0 + 1], code[i0 + 2], base_bits),
                                    wk_take32(code[i0 + 2], code[i0 + 3], base_bits), rcode[s], rcode[s - 1], rcode[s - 2]};
                wide_position<0>(ww, valid, out); wide_position<1>(ww, valid, out); wide_position<2>(ww, valid, out); wide_position<3>(ww, valid, out);
                wide_position<4>(ww, valid, out); wide_position<5>(ww, valid, out); wide_position<6>(ww, valid, out); wide_position<7>(ww, valid, out);
                wide_position<8>(ww, valid, out); wide_position<9>(ww, valid, out); wide_position<10>(ww, valid, out); wide_position<11>(ww, valid, out);
                wide_position<12>(ww, valid, out); wide_position<13>(ww, valid, out); wide_position<14>(ww, valid, out); wide_position<15>(ww, valid, out);
            }
            if (last[s] > before) before = last[s];
        }
    }
}

extern "C" {

// see wide_reduce above; returns 0, -1 on bad k
int emu_wide_reduce(const uint8_t *buf, uint64_t n, uint64_t n_padded, uint32_t k, int accept_u, uint64_t *out)
{
    if (k < 33 || k > 255) return -1;
    for (int i = 0; i < 4 + 4096; i++) out[i] = 0;
    if (accept_u) wide_reduce<true>(buf, n, n_padded, k, out); else wide_reduce<false>(buf, n, n_padded, k, out);
    return 0;
}

// out: [n_total, n_fwd, sum, xor, hist[4096]].  canon/tie_rc/accept_u as the kernel's template flags.
// values/valid16/rc16 may be null.  Returns 0, -1 on bad k.
int emu_scan(const uint8_t *buf, uint64_t n, uint64_t n_padded, uint32_t k, int canon, int tie_rc, int accept_u,
             uint32_t tiles_per_wave, uint64_t *out, uint64_t *values, uint16_t *valid16, uint16_t *rc16)
{
    if (k < 1 || k > 32) return -1;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    scan_args_set_k(a, k);
    a.n_bytes = n; (void)tiles_per_wave;
    HostStats *st = new HostStats();
    const int kw = k > 16 ? 2 : 1;
    // tiles_per_wave doubles as a switch in this emulation: an odd value selects the k-specialised build when one exists
    const bool fix = (tiles_per_wave & 1) && canon && (k == 21 || k == 31);
    // (bit 1 of tiles_per_wave selected the first scalar-validity generation until round 6; it is ignored now)
    // bit 2 of tiles_per_wave: the second-generation scalar-validity variant (17 <= k <= 32); bit 3 picks its 14-bit histogram
    const bool sv2 = (tiles_per_wave & 4) && canon && !values && k >= 17;
    const bool hb14 = (tiles_per_wave & 8) != 0;
    // the forward-only sv2 builds (BitNuclKmer, canonical = false)
    const bool sv2f = (tiles_per_wave & 4) && !canon && !values && k >= 17;
    const bool sv2w = (tiles_per_wave & 4) && !values && k <= 16;
#define EMU_SV2(KF, T, U) if (sv2 && k == KF && !!tie_rc == T && !!accept_u == U) { if (hb14) run_sv2<T, U, KF, 14>(buf, n, n_padded, st); else run_sv2<T, U, KF, 12>(buf, n, n_padded, st); } else
#define EMU_SV24(KF) EMU_SV2(KF, false, false) EMU_SV2(KF, false, true) EMU_SV2(KF, true, false) EMU_SV2(KF, true, true)
    EMU_SV24(17) EMU_SV24(18) EMU_SV24(19) EMU_SV24(20) EMU_SV24(21) EMU_SV24(22) EMU_SV24(23) EMU_SV24(24)
    EMU_SV24(25) EMU_SV24(26) EMU_SV24(27) EMU_SV24(28) EMU_SV24(29) EMU_SV24(30) EMU_SV24(31) EMU_SV24(32)
#define EMU_SV2F(KF, U) if (sv2f && k == KF && !!accept_u == U) { if (hb14) run_sv2<false, U, KF, 14, 0, true>(buf, n, n_padded, st); else run_sv2<false, U, KF, 12, 0, true>(buf, n, n_padded, st); } else
#define EMU_SV2F2(KF) EMU_SV2F(KF, false) EMU_SV2F(KF, true)
    EMU_SV2F2(17) EMU_SV2F2(18) EMU_SV2F2(19) EMU_SV2F2(20) EMU_SV2F2(21) EMU_SV2F2(22) EMU_SV2F2(23) EMU_SV2F2(24)
    EMU_SV2F2(25) EMU_SV2F2(26) EMU_SV2F2(27) EMU_SV2F2(28) EMU_SV2F2(29) EMU_SV2F2(30) EMU_SV2F2(31) EMU_SV2F2(32)
    // the word builds (k <= 16) of sv2: canonical (both tie rules) and forward-only
#define EMU_SV2W(KF, T, U) if (sv2w && canon && k == KF && !!tie_rc == T && !!accept_u == U) { if (hb14) run_sv2<T, U, KF, 14>(buf, n, n_padded, st); else run_sv2<T, U, KF, 12>(buf, n, n_padded, st); } else
#define EMU_SV2WF(KF, U) if (sv2w && !canon && k == KF && !!accept_u == U) { if (hb14) run_sv2<false, U, KF, 14, 0, true>(buf, n, n_padded, st); else run_sv2<false, U, KF, 12, 0, true>(buf, n, n_padded, st); } else
#define EMU_SV2W6(KF) EMU_SV2W(KF, false, false) EMU_SV2W(KF, false, true) EMU_SV2W(KF, true, false) EMU_SV2W(KF, true, true) EMU_SV2WF(KF, false) EMU_SV2WF(KF, true)
    EMU_SV2W6(1) EMU_SV2W6(2) EMU_SV2W6(3) EMU_SV2W6(4) EMU_SV2W6(5) EMU_SV2W6(6) EMU_SV2W6(7) EMU_SV2W6(8)
    EMU_SV2W6(9) EMU_SV2W6(10) EMU_SV2W6(11) EMU_SV2W6(12) EMU_SV2W6(13) EMU_SV2W6(14) EMU_SV2W6(15) EMU_SV2W6(16)
#define EMU_FIX(KF, T, U) if (fix && k == KF && !!tie_rc == T && !!accept_u == U) { run<2, true, T, U, KF>(buf, n, n_padded, a, st, values, valid16, rc16); } else
    EMU_FIX(21, false, false) EMU_FIX(21, false, true) EMU_FIX(21, true, false) EMU_FIX(21, true, true)
    EMU_FIX(31, false, false) EMU_FIX(31, false, true) EMU_FIX(31, true, false) EMU_FIX(31, true, true)
    {
#define EMU_CASE(KW, C, T, U) \
    if (kw == KW && !!canon == C && !!tie_rc == T && !!accept_u == U) run<KW, C, T, U>(buf, n, n_padded, a, st, values, valid16, rc16);
    EMU_CASE(1, false, false, false) EMU_CASE(1, false, false, true)
    EMU_CASE(1, true, false, false) EMU_CASE(1, true, false, true)
    EMU_CASE(1, true, true, false) EMU_CASE(1, true, true, true)
    EMU_CASE(1, false, true, false) EMU_CASE(1, false, true, true)
    EMU_CASE(2, false, false, false) EMU_CASE(2, false, false, true)
    EMU_CASE(2, true, false, false) EMU_CASE(2, true, false, true)
    EMU_CASE(2, true, true, false) EMU_CASE(2, true, true, true)
    EMU_CASE(2, false, true, false) EMU_CASE(2, false, true, true)
    }
    out[0] = st->n_total; out[1] = st->n_fwd; out[2] = st->sum; out[3] = st->xr;
    memcpy(out + 4, st->hist, sizeof(st->hist));
    delete st;
    return 0;
}

// The same with the quality stream of the QM builds: bases whose quality byte is below `cutoff` (1..255) are masked.
int emu_scan_quality(const uint8_t *buf, const uint8_t *qual, uint32_t cutoff, uint64_t n, uint64_t n_padded, uint32_t k, int canon,
                     int tie_rc, int accept_u, uint32_t tiles_per_wave, uint64_t *out, uint64_t *values, uint16_t *valid16,
                     uint16_t *rc16)
{
    if (cutoff < 1 || cutoff > 255 || !qual) return -1;
    g_qual = qual; g_qc = quality_cut(cutoff);
    const int rc = emu_scan(buf, n, n_padded, k, canon, tie_rc, accept_u, tiles_per_wave, out, values, valid16, rc16);
    g_qual = nullptr;
    return rc;
}

// Fused windowed minimizers (lane_tile_sv2_min).  Returns -2 when (k, w) has no fused build.
int emu_minimizers(const uint8_t *buf, uint64_t n, uint64_t n_padded, uint32_t k, uint32_t w, int tie_rc, int accept_u, int hb14,
                   uint64_t *out)
{
    HostStats *st = new HostStats();
    bool done = false;
#define EMU_MIN(KF, WF, T, U) if (!done && k == KF && w == WF && !!tie_rc == T && !!accept_u == U) { \
        if (hb14) run_sv2<T, U, KF, 14, WF>(buf, n, n_padded, st); else run_sv2<T, U, KF, 12, WF>(buf, n, n_padded, st); done = true; }
#define EMU_MIN4(KF, WF) EMU_MIN(KF, WF, false, false) EMU_MIN(KF, WF, false, true) EMU_MIN(KF, WF, true, false) EMU_MIN(KF, WF, true, true)
    EMU_MIN4(17, 11) EMU_MIN4(18, 11) EMU_MIN4(19, 11) EMU_MIN4(20, 11) EMU_MIN4(21, 11) EMU_MIN4(22, 11)
    EMU_MIN4(21, 9) EMU_MIN4(21, 10) EMU_MIN4(21, 12) EMU_MIN4(17, 16) EMU_MIN4(19, 14)
    EMU_MIN4(15, 10) EMU_MIN4(15, 9) EMU_MIN4(16, 12) EMU_MIN4(16, 16) EMU_MIN4(15, 16) EMU_MIN4(19, 10) EMU_MIN4(22, 9) EMU_MIN4(20, 13)
    EMU_MIN4(15, 5) EMU_MIN4(19, 5) EMU_MIN4(21, 5) EMU_MIN4(23, 5) EMU_MIN4(16, 2) EMU_MIN4(17, 3) EMU_MIN4(18, 4) EMU_MIN4(20, 6) EMU_MIN4(22, 7) EMU_MIN4(19, 8)   // short windows: doubling
    EMU_MIN4(23, 9) EMU_MIN4(23, 10) EMU_MIN4(23, 11) EMU_MIN4(23, 12) EMU_MIN4(22, 12) EMU_MIN4(21, 16) EMU_MIN4(23, 16)   // k = 23; windows of 33 .. 38 bytes: three halo lanes
    if (done) {
        out[0] = st->n_total; out[1] = st->n_fwd; out[2] = st->sum; out[3] = st->xr;
        memcpy(out + 4, st->hist, sizeof(st->hist));
    }
    delete st;
    return done ? 0 : -2;
}

// The generic fused minimizer kernel: any k <= 31, w <= 49; f64 = 1 takes the v_min_f64 key form (k <= 25 only), 0 the general keys.
// Returns -2 outside those ranges.
int emu_minimizers_generic(const uint8_t *buf, uint64_t n, uint64_t n_padded, uint32_t k, uint32_t w, int tie_rc, int accept_u, int f64,
                           uint64_t *out)
{
    if (k < 1 || k > 31 || w < 1 || w > 49 || (f64 && k > 25)) return -2;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    scan_args_set_k(a, k);
    scan_args_set_window(a, w);
    a.n_bytes = n;
    HostStats *st = new HostStats();
    const int kw = k > 16 ? 2 : 1;
#define EMU_MG(KW, T, U, F) if (kw == KW && !!tie_rc == T && !!accept_u == U && !!f64 == F) run_min_generic<KW, T, U, F>(buf, n, n_padded, a, st);
#define EMU_MG4(KW, F) EMU_MG(KW, false, false, F) EMU_MG(KW, false, true, F) EMU_MG(KW, true, false, F) EMU_MG(KW, true, true, F)
    EMU_MG4(1, false) EMU_MG4(1, true) EMU_MG4(2, false) EMU_MG4(2, true)
#undef EMU_MG4
#undef EMU_MG
    out[0] = st->n_total; out[1] = st->n_fwd; out[2] = st->sum; out[3] = st->xr;
    memcpy(out + 4, st->hist, sizeof(st->hist));
    delete st;
    return 0;
}

// ... with the quality stream of the QM builds: bases whose quality byte is below `cutoff` (1..255) are breaks.
int emu_minimizers_generic_quality(const uint8_t *buf, const uint8_t *qual, uint32_t cutoff, uint64_t n, uint64_t n_padded, uint32_t k, uint32_t w,
                                   int tie_rc, int accept_u, int f64, uint64_t *out)
{
    if (cutoff < 1 || cutoff > 255 || !qual) return -1;
    g_qual = qual; g_qc = quality_cut(cutoff);
    const int rc = emu_minimizers_generic(buf, n, n_padded, k, w, tie_rc, accept_u, f64, out);
    g_qual = nullptr;
    return rc;
}

int emu_window_masks(const uint64_t *g16, uint32_t k, uint64_t *ok16, uint64_t *ab16)
{
    uint64_t G[16];
    memcpy(G, g16, sizeof(G));
    switch (k) {
#define EMU_WM(KF) case KF: masks_both<KF>(G, ok16, ab16); return 0;
    EMU_WM(1) EMU_WM(2) EMU_WM(3) EMU_WM(4) EMU_WM(5) EMU_WM(6) EMU_WM(7) EMU_WM(8) EMU_WM(9) EMU_WM(10) EMU_WM(11) EMU_WM(12) EMU_WM(13)
    EMU_WM(14) EMU_WM(15) EMU_WM(16) EMU_WM(17) EMU_WM(18) EMU_WM(19) EMU_WM(20) EMU_WM(21) EMU_WM(22) EMU_WM(23) EMU_WM(24) EMU_WM(25)
    EMU_WM(26) EMU_WM(27) EMU_WM(28) EMU_WM(29) EMU_WM(30) EMU_WM(31) EMU_WM(32)
#undef EMU_WM
    }
    return -1;
}

// window_masks_ab_any<KM> for the windows of 33 .. 48 bytes of the fused-minimizer builds with three halo lanes (Sv2Geom): ab16[j] = A[j] & B[j]
int emu_window_masks_wide(const uint64_t *g16, uint32_t km, uint64_t *ab16)
{
    uint64_t G[16], A[16], B[16];
    memcpy(G, g16, sizeof(G));
    switch (km) {
#define EMU_WMW(KF) case KF: window_masks_ab_any<KF>(G, A, B); break;
    EMU_WMW(33) EMU_WMW(34) EMU_WMW(35) EMU_WMW(36) EMU_WMW(37) EMU_WMW(38) EMU_WMW(39) EMU_WMW(40)
    EMU_WMW(41) EMU_WMW(42) EMU_WMW(43) EMU_WMW(44) EMU_WMW(45) EMU_WMW(46) EMU_WMW(47) EMU_WMW(48)
#undef EMU_WMW
    default: return -1;
    }
    for (int j = 0; j < 16; j++) ab16[j] = A[j] & B[j];
    return 0;
}

// window_masks_runtime: the run-time form of the mask algebra (the generic fused minimizer kernel), any window length L = 1 .. 79
int emu_window_masks_runtime(const uint64_t *g16, uint32_t L, uint32_t halo_lanes, uint64_t *ok16)
{
    if (L < 1 || L > 79 || halo_lanes > 63) return -1;
    uint64_t G[16], A[16], B[16];
    memcpy(G, g16, sizeof(G));
    window_masks_runtime(G, A, B, L, ~((1ull << halo_lanes) - 1ull));
    for (int j = 0; j < 16; j++) ok16[j] = A[j] & B[j];
    return 0;
}

void emu_encode16(const uint8_t *raw16, int accept_u, uint32_t *out3)
{
    Raw16 r;
    memcpy(&r.x, raw16, 4); memcpy(&r.y, raw16 + 4, 4); memcpy(&r.z, raw16 + 8, 4); memcpy(&r.w, raw16 + 12, 4);
    Enc e = accept_u ? encode16<true>(r) : encode16<false>(r);
    out3[0] = e.code; out3[1] = e.rcode; out3[2] = e.bad;
}
}
