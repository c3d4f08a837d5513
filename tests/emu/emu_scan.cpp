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
    uint64_t *values; uint16_t *valid16, *rc16;
    uint64_t base = 0; uint32_t inval = 0, rcbits = 0;
    void begin_tile(uint64_t lane_base, uint32_t inval16) { base = lane_base; inval = inval16; rcbits = 0; }
    void emit(int j, bool valid, bool take_fwd, uint32_t hi, uint32_t lo)
    {
        const uint64_t v = KW == 2 ? (((uint64_t)hi << 32) | lo) : (uint64_t)lo;
        if (valid) {
            st->n_total++; st->n_fwd += take_fwd; st->sum += v; st->xr ^= v; st->hist[v >> bin_shift]++;
        }
        if (values) values[base + j] = v;
        rcbits |= (take_fwd ? 0u : 1u) << (15 - j);
    }
    void end_tile()
    {
        const uint32_t v = ~inval & 0xFFFFu;
        if (valid16) valid16[base >> 4] = (uint16_t)v;
        if (rc16) rc16[base >> 4] = (uint16_t)(rcbits & v);
    }
};

struct LaneHistE { uint32_t code = 0, code1 = 0, rcode = 0, rcode1 = 0, bad = 0xFFFFu, bad1 = 0xFFFFu; };

// lane_prev over a whole wave: out[l] = l ? cur[l-1] : prev_tile[63]
void lane_prev_wave(const uint32_t *cur, const uint32_t *prev_tile, uint32_t *out)
{
    out[0] = prev_tile[63];
    for (int l = 1; l < 64; l++) out[l] = cur[l - 1];
}

Raw16 load16(const uint8_t *buf, uint64_t n_padded, uint64_t off)
{
    uint8_t b[16];
    for (int i = 0; i < 16; i++) b[i] = off + i < n_padded ? buf[off + i] : 0;  // buffer bounds check returns 0
    Raw16 r;
    memcpy(&r.x, b, 4); memcpy(&r.y, b + 4, 4); memcpy(&r.z, b + 8, 4); memcpy(&r.w, b + 12, 4);
    return r;
}

template <int KW, bool CANON, bool TIE_RC, bool ACCEPT_U>
void run(const uint8_t *buf, uint64_t n, uint64_t n_padded, ScanArgs a, HostStats *st, uint64_t *values,
         uint16_t *valid16, uint16_t *rc16)
{
    const uint64_t n_tiles = (n + kTileBytes - 1) / kTileBytes;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += a.tiles_per_wave) {  // one wave per run of tiles
        uint64_t t1 = t0 + a.tiles_per_wave; if (t1 > n_tiles) t1 = n_tiles;
        LaneHistE ph[64];
        HostSink<KW> sinks[64];
        for (auto &s : sinks) { s.st = st; s.bin_shift = a.bin_shift; s.values = values; s.valid16 = valid16; s.rc16 = rc16; }
        for (uint64_t t = t0 ? t0 - 1 : 0; t < t1; t++) {
            const bool emit = t >= t0;
            Enc en[64];
            uint32_t code[64], rcode[64], bad[64], pc[64], pc1[64], pr[64], pr1[64], pb[64], pb1[64];
            for (int l = 0; l < 64; l++) {
                const uint64_t lane_base = t * kTileBytes + l * 16;
                en[l] = encode16<ACCEPT_U>(load16(buf, n_padded, lane_base));
                if (lane_base + 16 > n) {
                    const uint32_t keep = lane_base >= n ? 0u : (uint32_t)(n - lane_base);
                    en[l].bad |= 0xFFFFu >> keep;
                }
                code[l] = en[l].code; rcode[l] = en[l].rcode; bad[l] = en[l].bad;
                pc[l] = ph[l].code; pc1[l] = ph[l].code1; pr[l] = ph[l].rcode; pr1[l] = ph[l].rcode1;
                pb[l] = ph[l].bad; pb1[l] = ph[l].bad1;
            }
            uint32_t c1[64], c2[64], r1[64], r2[64], b1[64], b2[64];
            lane_prev_wave(code, pc, c1); lane_prev_wave(c1, pc1, c2);
            lane_prev_wave(rcode, pr, r1); lane_prev_wave(r1, pr1, r2);
            lane_prev_wave(bad, pb, b1); lane_prev_wave(b1, pb1, b2);
            for (int l = 0; l < 64; l++) {
                ph[l].code = code[l]; ph[l].code1 = c1[l]; ph[l].rcode = rcode[l]; ph[l].rcode1 = r1[l];
                ph[l].bad = bad[l]; ph[l].bad1 = b1[l];
                if (!emit) continue;
                TileWords tw;
                tw.W[0] = c2[l]; tw.W[1] = c1[l]; tw.W[2] = code[l];
                tw.R[0] = rcode[l]; tw.R[1] = r1[l]; tw.R[2] = r2[l];
                tw.bad48 = ((uint64_t)b2[l] << 32) | ((uint64_t)b1[l] << 16) | bad[l];
                emit_windows<KW, CANON, TIE_RC>(a, sinks[l], tw, t * kTileBytes + l * 16);
            }
        }
    }
}

}  // namespace

extern "C" {

// out: [n_total, n_fwd, sum, xor, hist[4096]].  canon/tie_rc/accept_u as the kernel's template flags.
// values/valid16/rc16 may be null.  Returns 0, -1 on bad k.
int emu_scan(const uint8_t *buf, uint64_t n, uint64_t n_padded, uint32_t k, int canon, int tie_rc, int accept_u,
             uint32_t tiles_per_wave, uint64_t *out, uint64_t *values, uint16_t *valid16, uint16_t *rc16)
{
    if (k < 1 || k > 32) return -1;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    scan_args_set_k(a, k);
    a.n_bytes = n; a.tiles_per_wave = tiles_per_wave ? tiles_per_wave : 1;
    HostStats *st = new HostStats();
    const int kw = k > 16 ? 2 : 1;
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
    out[0] = st->n_total; out[1] = st->n_fwd; out[2] = st->sum; out[3] = st->xr;
    memcpy(out + 4, st->hist, sizeof(st->hist));
    delete st;
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
