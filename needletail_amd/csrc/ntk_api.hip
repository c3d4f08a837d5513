// ntk_api.hip — host side of the C ABI declared in include/needletail_amd.h.
// Everything that computes launches HIP kernels (ntk_kernels.hpp); there is no CPU fallback.
#include "../../include/needletail_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is loaded at run time (load_rccl)
#include <dlfcn.h>
#include <time.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <thread>
#include <algorithm>
#include <vector>

#include "ntk_kernels.hpp"

using namespace ntk;

// the sv2 builds (ntk_scan2.hip): nullptr when (k, flags) has none
const void *ntk_pick_scan2(int k, bool tie_rc, bool accept_u);
const void *ntk_pick_scan2_q(int k, bool canonical, bool tie_rc, bool accept_u);
const void *ntk_pick_scan2_min_a(int k, int w, bool tie_rc, bool accept_u, bool quality);   // ntk_scan2_min.o: k = 15..18
const void *ntk_pick_scan2_min_b(int k, int w, bool tie_rc, bool accept_u, bool quality);   // ntk_scan2_min2.o: k = 19..22
const void *ntk_pick_scan2_fwd(int k, bool accept_u);

namespace {

thread_local int g_last_hip = 0;
thread_local int g_last_rccl = 0;

// HIP caps gridDim.x * blockDim.x below 2^32 and silently wraps beyond it: element-wise kernels are grid-stride and are
// launched with at most 2^20 blocks.
inline unsigned grid_for(uint64_t items, unsigned block) { const uint64_t b = (items + block - 1) / block; return (unsigned)(b > (1u << 20) ? (1u << 20) : (b ? b : 1)); }

#define HIPCHK(expr)                                     \
    do {                                                 \
        hipError_t e__ = (expr);                         \
        if (e__ != hipSuccess) {                         \
            g_last_hip = (int)e__;                       \
            (void)hipGetLastError();                     \
            return NTK_ERR_HIP;                          \
        }                                                \
    } while (0)

constexpr uint32_t kLutChanged = 0x100, kLutDeleted = 0x200;

// Byte maps, restated from the reference's match arms.
// normalize: reference src/sequence.rs:24-51 (priority order matters only for overlapping arms; none overlap).
void build_normalize_lut(uint16_t *lut, bool iupac)
{
    for (int c = 0; c < 256; c++) {
        uint32_t v;
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'N': case '-': v = (uint32_t)c; break;
        case 'a': v = 'A' | kLutChanged; break;
        case 'c': v = 'C' | kLutChanged; break;
        case 'g': v = 'G' | kLutChanged; break;
        case 't': case 'u': case 'U': v = 'T' | kLutChanged; break;
        case '.': case '~': v = '-' | kLutChanged; break;
        case 'B': case 'D': case 'H': case 'V': case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M':
            v = iupac ? (uint32_t)c : ('N' | kLutChanged); break;
        case 'b': case 'd': case 'h': case 'v': case 'r': case 'y': case 's': case 'w': case 'k': case 'm':
            v = iupac ? ((uint32_t)(c - 32) | kLutChanged) : ('N' | kLutChanged); break;
        case ' ': case '\t': case '\r': case '\n': v = ' ' | kLutChanged | kLutDeleted; break;
        default: v = 'N' | kLutChanged; break;
        }
        lut[c] = (uint16_t)v;
    }
}
// strip_returns: reference src/sequence.rs:165-191.
void build_strip_lut(uint16_t *lut)
{
    for (int c = 0; c < 256; c++) lut[c] = (uint16_t)((c == '\r' || c == '\n') ? (c | kLutChanged | kLutDeleted) : c);
}
// complement: reference src/sequence.rs:68-105.
void build_complement_lut(uint16_t *lut)
{
    for (int c = 0; c < 256; c++) lut[c] = (uint16_t)c;
    const char *pairs = "atcgrykmbvdhATCGRYKMBVDH";  // consecutive pairs swap; s/w/S/W map to themselves
    for (int i = 0; pairs[i]; i += 2) {
        lut[(uint8_t)pairs[i]] = (uint8_t)pairs[i + 1];
        lut[(uint8_t)pairs[i + 1]] = (uint8_t)pairs[i];
    }
}

struct Scratch {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// Everything one in-flight chunk of the batched compat face owns (ntk_canonical_kmers_batch / ntk_bit_kmers_batch).
struct CompatBank {
    void *h_stage = nullptr; size_t h_stage_bytes = 0;   // pinned: record starts + packed bytes
    uint64_t *h_total = nullptr;                         // pinned: the chunk's item total
    Scratch d[7];   // 0 packed bytes, 1 values (bit path) / flag bytes (byte path), 2 valid16, 3 rc16, 4 compaction scratch, 5 record starts, 6 dense outputs
    hipEvent_t ev_total = nullptr, ev_scattered = nullptr, ev_done = nullptr;
    uint64_t r0 = 0, nrec = 0, n = 0, n_words = 0, nblocks = 0;
    size_t o_off = 0, o_total = 0, o_counts = 0;
    bool busy = false;   // its D2H may still be running
};

struct ntk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // compute stream (kernels, compat-face copies)
    hipStream_t copy_stream = nullptr;  // H2D copies of pinned batches
    hipStream_t copy_stream2 = nullptr; // ... every second batch's when NTK_OPT_COPY_STREAMS is 2: the next copy is queued on the other DMA queue
                                        // while one runs, so the link does not idle between batches
    uint32_t copy_streams = 2, copy_rr = 0;
    hipStream_t down_stream = nullptr;  // D2H copies of the bit-plane compat face (its uploads keep the copy stream to themselves)
    bool owns_stream = false;
    int n_cu = 256;
    int launch_blocks = 0, launch_threads = 0;   // 0 = automatic
    uint64_t *d_acc = nullptr;      // accumulators in use (own or caller-bound)
    uint64_t *d_acc_own = nullptr;
    uint32_t *d_part_hist = nullptr;
    uint32_t *d_work = nullptr;     // kMaxShards work counters, one per 64-B line
    bool work_dirty = true;         // not known to be zero
    uint32_t *d_lower = nullptr;    // kLowerRing flag words behind the work counters: speculative scans of un-normalised byte-path input (run_scan)
    uint32_t lower_idx = 0;
    // ntk_ctx_set_option (test / A-B support, per ctx: nothing in the dispatch reads the environment)
    uint64_t minimizer_chunk = (uint64_t)256 << 20;  // NTK_OPT_MINIMIZER_CHUNK_BYTES: bytes of input per two-pass minimizer pass
    uint64_t compat_chunk = (uint64_t)16 << 20;      // NTK_OPT_COMPAT_CHUNK_BYTES: packed bytes per chunk of the batched compat faces
    uint32_t route_off = 0;                          // NTK_OPT_MINIMIZER_ROUTE: NTK_ROUTE_NO_* bits
    uint32_t pack_threads = 8;                       // NTK_OPT_COMPAT_PACK_THREADS
    uint64_t wait_poll_us = 50;                      // NTK_OPT_BATCH_WAIT_POLL_US (0: block in hipEventSynchronize)
    uint64_t gz_limit = 0, gz_window = 0, pipe_stats = 0;   // NTK_OPT_GZ_INMEM_LIMIT_BYTES / _STREAM_WINDOW_BYTES / NTK_OPT_PIPE_STATS: read by the producer (ntk_fastx_api.cpp) through ntk_ctx_get_option; 0 = its default
    uint64_t *d_part_scalars = nullptr;
    int part_blocks = 0;
    uint16_t *d_lut = nullptr;  // [0]=normalize(false) [1]=normalize(true) [2]=strip [3]=complement, 256 each
    Scratch scratch[6];
    CompatBank bank[3];        // the chunks the batched compat face keeps in flight (see compat_batch)
    void *h_pinned = nullptr;  // small pinned staging for scalar read-backs
    bool timing = false;
    std::vector<hipEvent_t> ev_free;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used;
    std::map<std::pair<const void *, int>, int> occupancy;  // resident blocks per CU of a scan build at a block size
    std::map<const void *, int> auto_threads;               // block size chosen for a reduce build (run_scan)
    // released batches are kept for re-use: pinned + device allocations cost milliseconds each, a parser thread's two
    // batches more than its share of a multi-GB input (tools/pipeline_bench.py).  Guarded: producer threads acquire and
    // release concurrently.
    std::mutex pool_mu;
    std::vector<ntk_batch *> pool;
    uint64_t pool_bytes = 0;
};

struct ntk_batch {
    uint8_t *h_seq = nullptr;
    uint64_t *h_off = nullptr;
    uint8_t *d_seq = nullptr;
    uint8_t *h_qual = nullptr, *d_qual = nullptr;  // allocated by the first ntk_batch_append_quality
    bool has_qual = false;                         // some record of the current fill carries qualities
    uint32_t qual_cutoff = 0;                      // ... appended for this cutoff (must be the one submitted)
    int device = 0;
    uint64_t cap_bytes = 0, cap_records = 0, n_bytes = 0, n_records = 0;
    hipEvent_t ev_copied = nullptr, ev_done = nullptr;
    bool in_flight = false;
};

namespace {

int ensure_scratch(ntk_ctx *c, int slot, size_t bytes)
{
    if (bytes < 256) bytes = 256;
    Scratch &s = c->scratch[slot];
    if (s.bytes >= bytes) return NTK_OK;
    if (s.p) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(s.p)); s.p = nullptr; s.bytes = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc(&s.p, want));
    s.bytes = want;
    return NTK_OK;
}

int ensure_partials(ntk_ctx *c, int blocks)
{
    if (c->part_blocks >= blocks) return NTK_OK;
    if (c->d_part_hist) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->d_part_hist)); HIPCHK(hipFree(c->d_part_scalars)); }
    c->d_part_hist = nullptr; c->d_part_scalars = nullptr; c->part_blocks = 0;
    HIPCHK(hipMalloc(&c->d_part_hist, (size_t)blocks * kHistBins * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_part_scalars, (size_t)blocks * 4 * sizeof(uint64_t)));
    c->part_blocks = blocks;
    return NTK_OK;
}

struct Mode { int kw; bool canon, tie_rc, accept_u; bool raw_bytes = false; };
inline uint32_t quality_cutoff(const ntk_params *p) { return (p->flags >> 8) & 0xFFu; }

int resolve_mode(const ntk_params *p, bool batch_face, Mode *m)
{
    if (!p) return NTK_ERR_BAD_ARG;
    // CanonicalKmers takes k: u8 (reference src/kmer.rs:48-82); 33 <= k <= 255 exists on the byte path's batch face in reduce mode only
    // (counters + histogram; 2-bit values of more than 64 bits have no sum / xor: ntk_result.n_undigested) - every other face is k <= 32
    if (p->k < 1 || p->k > (batch_face && p->path == NTK_PATH_BYTES_CANONICAL ? 255u : 32u)) return NTK_ERR_BAD_K;
    if ((p->flags & ~(0xFFFFu | NTK_FLAG_RESET)) != 0 || p->pre > NTK_PRE_NORMALIZE_IUPAC) return NTK_ERR_BAD_ARG;
    m->kw = p->k > 16 ? 2 : 1;
    m->accept_u = p->pre >= NTK_PRE_NORMALIZE;
    switch (p->path) {
    case NTK_PATH_BYTES_CANONICAL:
        // The byte path compares RAW bytes (reference src/kmer.rs:124); that equals the 2-bit order only when
        // every base has the same case, which normalize guarantees.  Un-normalised byte-path input is scanned speculatively (the packed-value
        // scan watches for lower case; canonical_bytes_reduce_kernel redoes the launch if there was any: run_scan); items: ntk_canonical_kmers*.
        m->raw_bytes = batch_face && (p->pre < NTK_PRE_NORMALIZE || p->k > 32);
        m->canon = true; m->tie_rc = true; break;
    case NTK_PATH_BITS: m->canon = false; m->tie_rc = false; break;
    case NTK_PATH_BITS_CANONICAL: m->canon = true; m->tie_rc = false; break;
    default: return NTK_ERR_BAD_ARG;
    }
    return NTK_OK;
}

// The scan kernel build for a mode.  Reduce mode: every (path, k, quality stream or not) has its own scan2_kernel instantiation (k is
// a template constant: the window-mask algebra indexes lane masks by k; ntk_scan2.hip).  Materialise mode: the round-1 scan_kernel with
// k at run time, plus a k = 21 build for the canonical paths (-5 %); it runs at the rate of a plain read-1-write-8 expansion kernel
// (2.45 ms per 1.51 GB of input against 2.45-2.50 ms, profiles/r04b/wbw.txt), i.e. it is bound by the 8 bytes it writes per position.
constexpr int kMaxShards = 256;      // work counters: the pull atomics of > 6000 waves on 8 counters were the bottleneck (profiles/r02)
constexpr uint32_t kLowerRing = 64;  // "a lower-case byte was seen" flags of consecutive speculative launches (each launch clears its successor's)

template <bool REDUCE, bool QM>
const void *pick_scan(const Mode &m, uint32_t k)
{
    // reduce mode: every (path, k, quality) has an sv2 build; the kernel lives in its own translation unit (ntk_scan2.hip, built
    // with the ILP-driven iterative scheduler).  The round-1 kernel below serves materialise mode only.
    if constexpr (REDUCE) {
        if (QM) return ntk_pick_scan2_q((int)k, m.canon, m.tie_rc, m.accept_u);
        return m.canon ? ntk_pick_scan2((int)k, m.tie_rc, m.accept_u) : ntk_pick_scan2_fwd((int)k, m.accept_u);
    } else {
#define NTK_PICK_FIX(KF, T, U)                                                                      \
    if (!QM && m.kw == 2 && m.canon && k == KF && m.tie_rc == T && m.accept_u == U)                  \
        return (const void *)&scan_kernel<2, true, T, U, false, KF>;
    NTK_PICK_FIX(21, false, false) NTK_PICK_FIX(21, false, true) NTK_PICK_FIX(21, true, false) NTK_PICK_FIX(21, true, true)
#undef NTK_PICK_FIX
#define NTK_PICK(KW, C, T, U)                                                                       \
    if (m.kw == KW && m.canon == C && m.tie_rc == T && m.accept_u == U)                             \
        return (const void *)&scan_kernel<KW, C, T, U, false, 0, QM>;
    NTK_PICK(1, false, false, false) NTK_PICK(1, false, false, true)
    NTK_PICK(1, true, false, false) NTK_PICK(1, true, false, true)
    NTK_PICK(1, true, true, false) NTK_PICK(1, true, true, true)
    NTK_PICK(2, false, false, false) NTK_PICK(2, false, false, true)
    NTK_PICK(2, true, false, false) NTK_PICK(2, true, false, true)
    NTK_PICK(2, true, true, false) NTK_PICK(2, true, true, true)
#undef NTK_PICK
    return nullptr;
    }
}

// Fused windowed-minimizer builds of the sv2 kernel (ntk_tile.hpp lane_tile_sv2_min): k = 15..23 x w = 5, 9..12 (windows of up to 34 bytes),
// and quality-masked builds of (21, 11) and (15, 10) (ntk_scan2.hip); every other (k, w) takes the two-pass path (materialise +
// window-min).
const void *pick_scan_min(const Mode &m, uint32_t k, uint32_t w, bool quality)
{
    if (!m.canon) return nullptr;
    if (const void *fn = ntk_pick_scan2_min_a((int)k, (int)w, m.tie_rc, m.accept_u, quality)) return fn;
    return ntk_pick_scan2_min_b((int)k, (int)w, m.tie_rc, m.accept_u, quality);
}

int get_event(ntk_ctx *c, hipEvent_t *e)
{
    if (!c->ev_free.empty()) { *e = c->ev_free.back(); c->ev_free.pop_back(); return NTK_OK; }
    HIPCHK(hipEventCreate(e));
    return NTK_OK;
}

// CanonicalKmers over raw (un-normalised) bytes into the accumulators: canonical_bytes_reduce_kernel + fold (ntk_kernels.hpp).
int raw_bytes_blocks(const ntk_ctx *c, uint64_t n)
{
    const uint64_t tile = (uint64_t)kPlThreads * 32;   // (canonical_bytes_reduce_kernel: 32 starts per thread)
    const uint64_t n_tiles = (n + tile - 1) / tile;
    const uint64_t max_blocks = c->launch_blocks > 0 ? (uint64_t)c->launch_blocks : (uint64_t)c->n_cu * 8;
    return (int)(n_tiles < max_blocks ? n_tiles : max_blocks);
}

int run_raw_bytes_reduce(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p, bool zero_first, bool normalized)
{
    if (zero_first) HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
    const int blocks = raw_bytes_blocks(c, n);
    int rc = ensure_partials(c, blocks);
    if (rc) return rc;
    const uint32_t pb = p->k < 6 ? p->k : 6;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        rc = get_event(c, &e0); if (rc) return rc;
        rc = get_event(c, &e1); if (rc) { c->ev_free.push_back(e0); return rc; }
        HIPCHK(hipEventRecord(e0, c->stream));
    }
    if (p->k > 32)
        hipLaunchKernelGGL(canonical_bytes_reduce_kernel<true>, dim3(blocks), dim3(kPlThreads), 0, c->stream, d_seq, n, (n + 15) & ~(uint64_t)15, p->k,
                           2u * (p->k - pb), (const uint16_t *)(c->d_lut + 768), c->d_part_hist, c->d_part_scalars, (const uint32_t *)nullptr, normalized ? 1u : 0u);
    else
        hipLaunchKernelGGL(canonical_bytes_reduce_kernel<false>, dim3(blocks), dim3(kPlThreads), 0, c->stream, d_seq, n, (n + 15) & ~(uint64_t)15, p->k,
                           2u * (p->k - pb), (const uint16_t *)(c->d_lut + 768), c->d_part_hist, c->d_part_scalars, (const uint32_t *)nullptr, 0u);
    HIPCHK(hipGetLastError());
    if (c->timing) { HIPCHK(hipEventRecord(e1, c->stream)); c->ev_used.emplace_back(e0, e1); }
    hipLaunchKernelGGL(fold_kernel, dim3(kFoldBlocks), dim3(kFoldThreads), 0, c->stream,
                       (const uint32_t *)c->d_part_hist, (const uint64_t *)c->d_part_scalars, blocks, c->d_acc, (uint32_t *)nullptr, 0,
                       (const uint32_t *)nullptr, 0, p->k > 32 ? 1 : 0);
    HIPCHK(hipGetLastError());
    return NTK_OK;
}

// CanonicalKmers with k = 33..255 (counters + histogram): wide_canonical_reduce_kernel decides the strand on the first 32 bases of the packed
// 2-bit streams; canonical_bytes_reduce_kernel<true> is queued behind it and returns at once unless that launch raised its flag (two k-mers
// equal over 32 bases, or - on input that was not normalised - a byte with bit 5 set), and the fold takes whichever partials are valid.  No
// host round trip; 0.9 ms instead of 9.5 per 1.5 GB at k = 64 (profiles/r06r).  The direct route under NTK_ROUTE_NO_SPECULATION.
int run_wide_reduce(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p, bool zero_first, bool normalized)
{
    if (c->route_off & NTK_ROUTE_NO_SPECULATION) return run_raw_bytes_reduce(c, d_seq, n, p, zero_first, normalized);
    if (zero_first) HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
    const uint64_t n_tiles = (n + kWkTile - 1) / kWkTile;
    const uint64_t max_blocks = c->launch_blocks > 0 ? (uint64_t)c->launch_blocks : (uint64_t)c->n_cu * 8;
    const int blocks = (int)(n_tiles < max_blocks ? n_tiles : max_blocks), blocks_raw = raw_bytes_blocks(c, n);
    int rc = ensure_partials(c, blocks > blocks_raw ? blocks : blocks_raw);
    if (rc) return rc;
    uint32_t *flag = c->d_lower + c->lower_idx;
    c->lower_idx = (c->lower_idx + 1) % kLowerRing;
    uint32_t *flag_next = c->d_lower + c->lower_idx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        rc = get_event(c, &e0); if (rc) return rc;
        rc = get_event(c, &e1); if (rc) { c->ev_free.push_back(e0); return rc; }
        HIPCHK(hipEventRecord(e0, c->stream));
    }
    if (normalized)
        hipLaunchKernelGGL(wide_canonical_reduce_kernel<true>, dim3(blocks), dim3(kWkThreads), 0, c->stream, d_seq, n, p->k, c->d_part_hist, c->d_part_scalars, flag, flag_next);
    else
        hipLaunchKernelGGL(wide_canonical_reduce_kernel<false>, dim3(blocks), dim3(kWkThreads), 0, c->stream, d_seq, n, p->k, c->d_part_hist, c->d_part_scalars, flag, flag_next);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(canonical_bytes_reduce_kernel<true>, dim3(blocks_raw), dim3(kPlThreads), 0, c->stream, d_seq, n, (n + 15) & ~(uint64_t)15, p->k,
                       2u * (p->k - 6u), (const uint16_t *)(c->d_lut + 768), c->d_part_hist, c->d_part_scalars, (const uint32_t *)flag, normalized ? 1u : 0u);
    HIPCHK(hipGetLastError());
    if (c->timing) { HIPCHK(hipEventRecord(e1, c->stream)); c->ev_used.emplace_back(e0, e1); }
    hipLaunchKernelGGL(fold_kernel, dim3(kFoldBlocks), dim3(kFoldThreads), 0, c->stream,
                       (const uint32_t *)c->d_part_hist, (const uint64_t *)c->d_part_scalars, blocks, c->d_acc, (uint32_t *)nullptr, 0,
                       (const uint32_t *)flag, blocks_raw, 1);
    HIPCHK(hipGetLastError());
    return NTK_OK;
}

// One scan over d_seq[0, n): launches cover at most kMaxTilesPerLaunch tiles each so that per-block u32
// histogram cells and 32-bit buffer offsets cannot overflow.
int run_scan(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p, const Mode &m, bool reduce,
             uint64_t *d_values, uint16_t *d_valid16, uint16_t *d_rc16, const uint8_t *d_qual = nullptr, const void *fused_min_fn = nullptr,
             int halo_lanes = kHaloLanes)   // 3 for the fused-minimizer builds whose windows need more than 32 bytes (ntk_tile.hpp Sv2Geom)
{
    const uint64_t tile_slots = 64 - (uint64_t)halo_lanes, tile_stride = tile_slots * 16;
    bool zero_first = reduce && (p->flags & NTK_FLAG_RESET);   // the first launch zeroes the accumulators in its prologue
    if (n == 0) {
        if (zero_first) HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
        return NTK_OK;
    }
    if (!d_seq || ((uintptr_t)d_seq & 15) || ((uintptr_t)d_qual & 15)) return NTK_ERR_BAD_ARG;
    const uint32_t cutoff = d_qual ? quality_cutoff(p) : 0u;  // cutoff 0 masks nothing: the plain build runs
    const uint64_t kMaxTilesPerLaunch = (uint64_t)8 << 22;
    bool speculate = false;
    if (m.raw_bytes) {
        // byte path on input that was not normalised: reduce mode only; dense values, quality masking and windowed minimizers on such
        // input are not built (normalize first, as the reference's documented chain does), and k > 32 has counters + histogram only
        if (p->k > 32 && (!reduce || cutoff || fused_min_fn)) return NTK_ERR_BAD_K;
        if (!reduce || cutoff || fused_min_fn) return NTK_ERR_UNSUPPORTED;
        // k <= 32: the raw-byte order is the 2-bit order unless a base is lower case - the clean read on which Sequence::normalize returns
        // None (reference src/sequence.rs:57-61).  So the packed-value scan runs (its TIE_RC, !ACCEPT_U build watches every byte it loads for
        // bit 5), the raw-byte kernel is queued behind it and returns at once unless the flag went up, and the fold takes whichever partials
        // are valid: no host round trip, 0.45 ms instead of 5.1 per 1.5 GB of upper-case reads.  One launch only (the raw-byte kernel works on
        // window starts, the scan on window ends: their launch ranges do not line up), the direct route otherwise and under NTK_ROUTE_NO_SPECULATION.
        const uint64_t tiles_all = ((n + 15) / 16 + tile_slots - 1) / tile_slots;
        speculate = p->k <= 32 && tiles_all <= kMaxTilesPerLaunch && !(c->route_off & NTK_ROUTE_NO_SPECULATION);
        if (p->k > 32) return run_wide_reduce(c, d_seq, n, p, zero_first, m.accept_u);
        if (!speculate) return run_raw_bytes_reduce(c, d_seq, n, p, zero_first, false);
    }
    // materialise mode stages 8.7 KiB per wave through LDS: 256-thread blocks, 4 per CU
    const void *fn = fused_min_fn ? fused_min_fn
                   : cutoff ? (reduce ? pick_scan<true, true>(m, p->k) : pick_scan<false, true>(m, p->k))
                            : (reduce ? pick_scan<true, false>(m, p->k) : pick_scan<false, false>(m, p->k));   // (speculate: tie_rc, !accept_u - the SPEC build)
    if (!fn) return NTK_ERR_BAD_ARG;
    // every reduce-mode scan (any path, any k, with or without a quality stream, fused minimizers) is a scan2 build: 768 threads, two
    // blocks per CU = 6 waves per SIMD, which needs <= 80 VGPRs.  A build above that would get ONE 768-thread block per CU; it runs
    // 640-thread blocks instead (two per CU: 5 waves per SIMD).  (No shipped build is above it: the k-mer builds need <= 75, the
    // fused-minimizer builds are compiled under the 6-wave budget, csrc/Makefile MIN_FLAGS.)
    int threads = !reduce ? 256 : (c->launch_threads ? c->launch_threads : 768);
    if (reduce && !c->launch_threads) {
        auto it = c->auto_threads.find(fn);
        if (it != c->auto_threads.end()) threads = it->second;
        else {
            int best_waves = 0;
            for (int t : {768, 640, 512}) {
                int blocks_cu = 0;
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_cu, fn, t, 0));
                if (blocks_cu > 2) blocks_cu = 2;
                if (blocks_cu * t > best_waves) { best_waves = blocks_cu * t; threads = t; }
            }
            c->auto_threads[fn] = threads;
        }
    }
    const int waves_per_block = threads / 64;
    const size_t lds = reduce ? 0 : (size_t)waves_per_block * kStageWaveU64 * sizeof(uint64_t);  // materialise staging
    // auto grid: exactly the blocks that are resident at once (work is pulled, so a second round of blocks would only
    // zero and write out empty histograms: measured +1.5 % at config 2).  Reduce builds: two blocks of `threads` (768, or 640 / 512
    // for a build above 80 VGPRs: chosen above) per CU, each with its 64 KiB LDS histogram; materialise: 4 x 256 (LDS staging).
    int per_cu = 0;
    auto it = c->occupancy.find(std::make_pair(fn, threads));
    if (it != c->occupancy.end()) per_cu = it->second;
    else {
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds));
        if (per_cu < 1) per_cu = 1;
        const int cap = reduce ? 2 * (1024 / threads) : 4;
        if (per_cu > cap) per_cu = cap;
        c->occupancy[std::make_pair(fn, threads)] = per_cu;
    }
    const int blocks_max = c->launch_blocks > 0 ? c->launch_blocks : c->n_cu * per_cu;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    scan_args_set_k(a, p->k);
    a.seq = d_seq;
    a.n_bytes = n;
    a.n_tiles = ((n + 15) / 16 + tile_slots - 1) / tile_slots;
    a.values = d_values; a.valid16 = d_valid16; a.rc16 = d_rc16;
    if (cutoff) { const QualityCut qc = quality_cut(cutoff); a.qual = d_qual; a.q_add = qc.add; a.q_sel = qc.sel; }
    // per launch: a shard holds <= 2^22 tiles so that the per-block u32 histogram cells (a block can at most drain its
    // whole shard: 2^22 * 992 windows) and the u32 work counters cannot overflow
    for (uint64_t tb = 0; tb < a.n_tiles; tb += kMaxTilesPerLaunch) {
        const uint64_t te = tb + kMaxTilesPerLaunch < a.n_tiles ? tb + kMaxTilesPerLaunch : a.n_tiles;
        const uint64_t tiles = te - tb;
        uint64_t chunk = tiles / ((uint64_t)blocks_max * waves_per_block * 4);  // >= ~4 pulls per wave, <= 24 tiles each (8..32 are within 1 %: profiles/r02c)
        chunk = chunk < 1 ? 1 : (chunk > 24 ? 24 : chunk);
        const uint64_t want_blocks = (tiles + chunk * waves_per_block - 1) / (chunk * waves_per_block);
        const int blocks = (int)(want_blocks < (uint64_t)blocks_max ? want_blocks : (uint64_t)blocks_max);
        a.tile_begin = tb; a.tile_end = te;
        {   // tiles t with (t + 1) * 992 > n_bytes touch the end of the input: t >= n_bytes / 992
            const uint64_t first_tail = n / tile_stride;
            a.tail_tile_rel = first_tail < tb ? 0u : (first_tail - tb > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)(first_tail - tb));
        }
        a.n_shards = blocks < kMaxShards ? (uint32_t)blocks : (uint32_t)kMaxShards;
        a.tiles_per_shard = (uint32_t)((tiles + a.n_shards - 1) / a.n_shards);
        a.chunk_tiles = (uint32_t)chunk;
        a.work_counters = c->d_work;
        a.zero_acc = zero_first ? c->d_acc : nullptr; a.zero_words = NTK_ACC_WORDS;
        zero_first = false;
        // the work counters are zero on entry: the fold kernel of the previous reduce scan re-armed them; anything else
        // (first use, a materialise scan, an error on the way) leaves work_dirty set and costs a memset here
        if (c->work_dirty) HIPCHK(hipMemsetAsync(c->d_work, 0, kMaxShards * 64, c->stream));
        c->work_dirty = true;
        const int blocks_raw = speculate ? raw_bytes_blocks(c, n) : 0;
        if (reduce) {
            int rc = ensure_partials(c, blocks > blocks_raw ? blocks : blocks_raw);
            if (rc) return rc;
            a.part_hist = c->d_part_hist; a.part_scalars = c->d_part_scalars;
        }
        uint32_t *flag = nullptr;
        if (speculate) {
            flag = c->d_lower + c->lower_idx;
            c->lower_idx = (c->lower_idx + 1) % kLowerRing;
            a.lower_flag = flag; a.lower_flag_next = c->d_lower + c->lower_idx;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c->timing) {
            int rc = get_event(c, &e0); if (rc) return rc;
            rc = get_event(c, &e1); if (rc) { c->ev_free.push_back(e0); return rc; }
            HIPCHK(hipEventRecord(e0, c->stream));
        }
        void *kargs[] = {(void *)&a};
        HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), kargs, lds, c->stream));
        if (speculate) {   // (inside the timed span: the pair is this route's scan)
            const uint32_t pb = p->k < 6 ? p->k : 6;
            hipLaunchKernelGGL(canonical_bytes_reduce_kernel<false>, dim3(blocks_raw), dim3(kPlThreads), 0, c->stream, d_seq, n, (n + 15) & ~(uint64_t)15, p->k,
                               2u * (p->k - pb), (const uint16_t *)(c->d_lut + 768), c->d_part_hist, c->d_part_scalars, (const uint32_t *)flag, 0u);
            HIPCHK(hipGetLastError());
        }
        if (c->timing) {
            HIPCHK(hipEventRecord(e1, c->stream));
            c->ev_used.emplace_back(e0, e1);
        }
        if (reduce) {
            hipLaunchKernelGGL(fold_kernel, dim3(kFoldBlocks), dim3(kFoldThreads), 0, c->stream,
                               (const uint32_t *)c->d_part_hist, (const uint64_t *)c->d_part_scalars, blocks, c->d_acc, c->d_work, (int)a.n_shards,
                               (const uint32_t *)flag, blocks_raw, 0);
            HIPCHK(hipGetLastError());
            c->work_dirty = false;
        }
    }
    return NTK_OK;
}

// Generic fused windowed minimizers (ntk_kernels.hpp minimizer_scan_kernel): any k <= 31 and w <= 49 of the canonical paths, with or
// without a quality stream; the tile geometry depends on w (2 + ceil((w - 1) / 16) non-emitting lanes).
const void *pick_min_generic(const Mode &m, bool quality, bool f64, uint32_t k)   // f64: k <= 25 (one v_min_f64 per minimum, ntk_tile.hpp)
{
    const int kw = f64 ? 2 : m.kw;   // (the f64 keys are built from the code streams for any k: one instantiation serves both word counts)
    const int mode = min_gen_mode(k, f64);
#define NTK_PICK_MG(KW, T, U, Q, F, MD) if (kw == KW && m.tie_rc == T && m.accept_u == U && quality == Q && f64 == F && mode == MD) return (const void *)&minimizer_scan_kernel<KW, T, U, Q, F, MD>;
#define NTK_PICK_MG4(KW, Q, F, MD) NTK_PICK_MG(KW, false, false, Q, F, MD) NTK_PICK_MG(KW, false, true, Q, F, MD) NTK_PICK_MG(KW, true, false, Q, F, MD) NTK_PICK_MG(KW, true, true, Q, F, MD)
#define NTK_PICK_MG8(KW, F, MD) NTK_PICK_MG4(KW, false, F, MD) NTK_PICK_MG4(KW, true, F, MD)
    NTK_PICK_MG8(2, true, 0) NTK_PICK_MG8(2, true, 1) NTK_PICK_MG8(2, true, 3) NTK_PICK_MG8(2, true, 2)   // f64 keys: k <= 7, 8..18, 19..23, 24..25
    NTK_PICK_MG8(2, false, 2)                                                                             // 26 <= k <= 31
    NTK_PICK_MG8(2, false, 1) NTK_PICK_MG8(1, false, 0) NTK_PICK_MG8(1, false, 1)                         // (only under NTK_ROUTE_NO_F64, the A/B switch)
#undef NTK_PICK_MG8
#undef NTK_PICK_MG4
#undef NTK_PICK_MG
    return nullptr;
}

int run_min_scan(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p, const Mode &m, uint32_t w, const uint8_t *d_qual)
{
    if (!d_seq || ((uintptr_t)d_seq & 15) || ((uintptr_t)d_qual & 15)) return NTK_ERR_BAD_ARG;
    const uint32_t cutoff = d_qual ? quality_cutoff(p) : 0u;
    const bool f64 = p->k <= 25 && !(c->route_off & NTK_ROUTE_NO_F64);
    const void *fn = pick_min_generic(m, cutoff != 0, f64, p->k);
    if (!fn) return NTK_ERR_BAD_ARG;
    const int threads = min_gen_threads(f64);   // 512: two blocks per CU, 768: one (ntk_kernels.hpp), each with its 64 KiB LDS histogram
    int per_cu = 0;
    auto it = c->occupancy.find(std::make_pair(fn, threads));
    if (it != c->occupancy.end()) per_cu = it->second;
    else {
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0));
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        c->occupancy[std::make_pair(fn, threads)] = per_cu;
    }
    const int blocks_max = c->launch_blocks > 0 ? c->launch_blocks : c->n_cu * per_cu;
    const int waves_per_block = threads / 64;
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    scan_args_set_k(a, p->k);
    a.seq = d_seq; a.n_bytes = n;
    if (cutoff) { const QualityCut qc = quality_cut(cutoff); a.qual = d_qual; a.q_add = qc.add; a.q_sel = qc.sel; }
    scan_args_set_window(a, w);   // halo lanes, the validity smear, the overlap of the two power-of-two windows (ntk_tile.hpp)
    const uint64_t slots = 64 - a.min_halo_lanes, stride = slots * 16;
    a.n_tiles = ((n + 15) / 16 + slots - 1) / slots;
    bool zero_first = (p->flags & NTK_FLAG_RESET) != 0;
    const uint64_t kMaxTilesPerLaunch = (uint64_t)8 << 22;
    for (uint64_t tb = 0; tb < a.n_tiles; tb += kMaxTilesPerLaunch) {
        const uint64_t te = tb + kMaxTilesPerLaunch < a.n_tiles ? tb + kMaxTilesPerLaunch : a.n_tiles;
        const uint64_t tiles = te - tb;
        uint64_t chunk = tiles / ((uint64_t)blocks_max * waves_per_block * 4);
        chunk = chunk < 1 ? 1 : (chunk > 16 ? 16 : chunk);
        const uint64_t want_blocks = (tiles + chunk * waves_per_block - 1) / (chunk * waves_per_block);
        const int blocks = (int)(want_blocks < (uint64_t)blocks_max ? want_blocks : (uint64_t)blocks_max);
        a.tile_begin = tb; a.tile_end = te;
        const uint64_t first_tail = n / stride;
        a.tail_tile_rel = first_tail < tb ? 0u : (first_tail - tb > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)(first_tail - tb));
        a.n_shards = blocks < kMaxShards ? (uint32_t)blocks : (uint32_t)kMaxShards;
        a.tiles_per_shard = (uint32_t)((tiles + a.n_shards - 1) / a.n_shards);
        a.chunk_tiles = (uint32_t)chunk;
        a.work_counters = c->d_work;
        a.zero_acc = zero_first ? c->d_acc : nullptr; a.zero_words = NTK_ACC_WORDS;
        zero_first = false;
        if (c->work_dirty) HIPCHK(hipMemsetAsync(c->d_work, 0, kMaxShards * 64, c->stream));
        c->work_dirty = true;
        int rc = ensure_partials(c, blocks);
        if (rc) return rc;
        a.part_hist = c->d_part_hist; a.part_scalars = c->d_part_scalars;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c->timing) {
            rc = get_event(c, &e0); if (rc) return rc;
            rc = get_event(c, &e1); if (rc) { c->ev_free.push_back(e0); return rc; }
            HIPCHK(hipEventRecord(e0, c->stream));
        }
        void *kargs[] = {(void *)&a};
        HIPCHK(hipLaunchKernel(fn, dim3(blocks), dim3(threads), kargs, 0, c->stream));
        if (c->timing) { HIPCHK(hipEventRecord(e1, c->stream)); c->ev_used.emplace_back(e0, e1); }
        hipLaunchKernelGGL(fold_kernel, dim3(kFoldBlocks), dim3(kFoldThreads), 0, c->stream,
                           (const uint32_t *)c->d_part_hist, (const uint64_t *)c->d_part_scalars, blocks, c->d_acc, c->d_work, (int)a.n_shards);
        HIPCHK(hipGetLastError());
        c->work_dirty = false;
    }
    return NTK_OK;
}

void destroy_batch(ntk_batch *b)
{
    if (b->h_seq) (void)hipHostFree(b->h_seq);
    if (b->h_off) (void)hipHostFree(b->h_off);
    if (b->d_seq) (void)hipFree(b->d_seq);
    if (b->h_qual) (void)hipHostFree(b->h_qual);
    if (b->d_qual) (void)hipFree(b->d_qual);
    if (b->ev_copied) (void)hipEventDestroy(b->ev_copied);
    if (b->ev_done) (void)hipEventDestroy(b->ev_done);
    delete b;
}

int init_ctx(ntk_ctx *c, void *stream, bool borrow);

int create_ctx(int device, void *stream, bool borrow, ntk_ctx **out)
{
    if (!out) return NTK_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return NTK_ERR_NO_DEVICE; }
    if (device < 0 || device >= count) return NTK_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return NTK_ERR_NO_DEVICE;  // kernels are built for gfx950 only
    ntk_ctx *c = new (std::nothrow) ntk_ctx();
    if (!c) return NTK_ERR_NOMEM;
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int rc = init_ctx(c, stream, borrow);
    if (rc != NTK_OK) { ntk_ctx_destroy(c); return rc; }   // nothing created so far is leaked (destroy skips what is null)
    *out = c;
    return NTK_OK;
}

int init_ctx(ntk_ctx *c, void *stream, bool borrow)
{
    if (borrow) { c->stream = (hipStream_t)stream; c->owns_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->owns_stream = true; }
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&c->d_acc_own, NTK_ACC_WORDS * sizeof(uint64_t)));
    c->d_acc = c->d_acc_own;
    HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
    HIPCHK(hipMalloc(&c->d_work, kMaxShards * 64 + kLowerRing * sizeof(uint32_t)));
    c->d_lower = c->d_work + kMaxShards * 16;
    HIPCHK(hipMemsetAsync(c->d_lower, 0, kLowerRing * sizeof(uint32_t), c->stream));
    HIPCHK(hipMalloc(&c->d_lut, 4 * 256 * sizeof(uint16_t)));
    HIPCHK(hipHostMalloc(&c->h_pinned, 64 * 1024, hipHostMallocDefault));
    uint16_t *h = (uint16_t *)c->h_pinned;
    build_normalize_lut(h, false);
    build_normalize_lut(h + 256, true);
    build_strip_lut(h + 512);
    build_complement_lut(h + 768);
    HIPCHK(hipMemcpyAsync(c->d_lut, h, 4 * 256 * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

}  // namespace

extern "C" {

const char *ntk_strerror(int s)
{
    switch (s) {
    case NTK_OK: return "ok";
    case NTK_ERR_BAD_K: return "k out of range for this entry point";
    case NTK_ERR_BAD_ARG: return "bad argument (null / misaligned pointer or bad enum)";
    case NTK_ERR_HIP: return "HIP runtime error (see ntk_last_hip_error)";
    case NTK_ERR_NO_DEVICE: return "no usable gfx950 device";
    case NTK_ERR_CAPACITY: return "output or batch capacity too small";
    case NTK_ERR_UNSUPPORTED: return "combination not supported on the device path (no CPU fallback exists)";
    case NTK_ERR_NOMEM: return "out of memory";
    case NTK_ERR_PARSE: return "FASTA/FASTQ parse error (see ntk_reader_error)";
    case NTK_ERR_RCCL: return "RCCL error (see ntk_last_rccl_error)";
    case NTK_EOF: return "end of input";
    default: return "unknown status";
    }
}
int ntk_last_hip_error(void) { return g_last_hip; }
int ntk_last_rccl_error(void) { return g_last_rccl; }
int ntk_abi_version(void) { return NTK_ABI_VERSION; }

int ntk_device_count(int *n)
{
    if (!n) return NTK_ERR_BAD_ARG;
    *n = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return NTK_OK; }
    for (int d = 0; d < count; d++) {   // devices are usable as a prefix 0 .. n-1: stop at the first that is not gfx950
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) { (void)hipGetLastError(); break; }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) break;
        *n = d + 1;
    }
    return NTK_OK;
}

int ntk_ctx_create(int device, ntk_ctx **out) { return create_ctx(device, nullptr, false, out); }
int ntk_ctx_create_on_stream(int device, void *hip_stream, ntk_ctx **out) { return create_ctx(device, hip_stream, true, out); }

void ntk_ctx_destroy(ntk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream || !c->owns_stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->copy_stream2) (void)hipStreamSynchronize(c->copy_stream2);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    for (auto &p : c->ev_used) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto e : c->ev_free) (void)hipEventDestroy(e);
    for (auto b : c->pool) destroy_batch(b);
    c->pool.clear();
    for (auto &s : c->scratch) if (s.p) (void)hipFree(s.p);
    if (c->d_part_hist) (void)hipFree(c->d_part_hist);
    if (c->d_part_scalars) (void)hipFree(c->d_part_scalars);
    if (c->d_acc_own) (void)hipFree(c->d_acc_own);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_work) (void)hipFree(c->d_work);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (CompatBank &b : c->bank) {
        if (b.h_stage) (void)hipHostFree(b.h_stage);
        if (b.h_total) (void)hipHostFree(b.h_total);
        for (auto &s : b.d) if (s.p) (void)hipFree(s.p);
        for (hipEvent_t e : {b.ev_total, b.ev_scattered, b.ev_done}) if (e) (void)hipEventDestroy(e);
    }
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->copy_stream2) (void)hipStreamDestroy(c->copy_stream2);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    delete c;
}

int ntk_ctx_synchronize(ntk_ctx *c)
{
    if (!c) return NTK_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(c->device));   // (a process may drive several devices: every entry point selects its ctx's own)
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->copy_stream2));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

int ntk_ctx_set_launch(ntk_ctx *c, int blocks, int threads)
{
    if (!c || blocks < 0 || threads < 0 || threads > 1024 || (threads & 63)) return NTK_ERR_BAD_ARG;   // threads 0 = automatic
    c->launch_blocks = blocks; c->launch_threads = threads;
    return NTK_OK;
}

int ntk_ctx_get_option(ntk_ctx *c, int option, uint64_t *value)
{
    if (!c || !value) return NTK_ERR_BAD_ARG;
    switch (option) {
    case NTK_OPT_COMPAT_CHUNK_BYTES: *value = c->compat_chunk; return NTK_OK;
    case NTK_OPT_MINIMIZER_CHUNK_BYTES: *value = c->minimizer_chunk; return NTK_OK;
    case NTK_OPT_MINIMIZER_ROUTE: *value = c->route_off; return NTK_OK;
    case NTK_OPT_COMPAT_PACK_THREADS: *value = c->pack_threads; return NTK_OK;
    case NTK_OPT_COPY_STREAMS: *value = c->copy_streams; return NTK_OK;
    case NTK_OPT_BATCH_WAIT_POLL_US: *value = c->wait_poll_us ? c->wait_poll_us : NTK_POLL_BLOCK; return NTK_OK;
    case NTK_OPT_GZ_INMEM_LIMIT_BYTES: *value = c->gz_limit; return NTK_OK;
    case NTK_OPT_GZ_STREAM_WINDOW_BYTES: *value = c->gz_window; return NTK_OK;
    case NTK_OPT_PIPE_STATS: *value = c->pipe_stats; return NTK_OK;
    default: return NTK_ERR_BAD_ARG;
    }
}

int ntk_ctx_set_option(ntk_ctx *c, int option, uint64_t value)
{
    if (!c) return NTK_ERR_BAD_ARG;
    switch (option) {
    case NTK_OPT_COMPAT_CHUNK_BYTES:      // 0 = default; below 64 bytes is taken as 64 (a stray 1 would cost a launch and an event wait per record)
        c->compat_chunk = value == 0 ? ((uint64_t)16 << 20) : (value < 64 ? 64 : value);
        return NTK_OK;
    case NTK_OPT_MINIMIZER_CHUNK_BYTES:   // 0 = default; a multiple of 4096, at least 4096
        if (value && value < 4096) return NTK_ERR_BAD_ARG;
        c->minimizer_chunk = value == 0 ? ((uint64_t)256 << 20) : (value & ~(uint64_t)4095);
        return NTK_OK;
    case NTK_OPT_MINIMIZER_ROUTE:
        if (value & ~(uint64_t)(NTK_ROUTE_NO_REGFUSED | NTK_ROUTE_NO_GENERIC | NTK_ROUTE_NO_F64 | NTK_ROUTE_NO_SPECULATION)) return NTK_ERR_BAD_ARG;
        c->route_off = (uint32_t)value;
        return NTK_OK;
    case NTK_OPT_BATCH_WAIT_POLL_US:      // 0 = default (50); NTK_POLL_BLOCK = block in hipEventSynchronize
        c->wait_poll_us = value == 0 ? 50 : (value == NTK_POLL_BLOCK ? 0 : (value > 10000000 ? 10000000 : value));
        return NTK_OK;
    case NTK_OPT_GZ_INMEM_LIMIT_BYTES: c->gz_limit = value; return NTK_OK;
    case NTK_OPT_GZ_STREAM_WINDOW_BYTES: c->gz_window = value; return NTK_OK;
    case NTK_OPT_PIPE_STATS: c->pipe_stats = value; return NTK_OK;
    case NTK_OPT_COPY_STREAMS:            // 0 = default (2); 1 or 2 HIP streams take the pinned batches' H2D copies in turn
        if (value > 2) return NTK_ERR_BAD_ARG;
        c->copy_streams = value ? (uint32_t)value : 2u;
        return NTK_OK;
    case NTK_OPT_COMPAT_PACK_THREADS: {   // 0 = default (8); capped by the hardware threads and 64
        uint64_t v = value ? value : 8;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && v > hw) v = hw;
        c->pack_threads = (uint32_t)(v > 64 ? 64 : v);
        return NTK_OK;
    }
    default: return NTK_ERR_BAD_ARG;
    }
}

int ntk_ctx_enable_timing(ntk_ctx *c, int on)
{
    if (!c) return NTK_ERR_BAD_ARG;
    c->timing = on != 0;
    return NTK_OK;
}

int ntk_ctx_scan_time_ms(ntk_ctx *c, double *total_ms, uint64_t *launches)
{
    if (!c || !total_ms) return NTK_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double t = 0;
    for (auto &p : c->ev_used) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
        t += ms;
        c->ev_free.push_back(p.first); c->ev_free.push_back(p.second);
    }
    if (launches) *launches = c->ev_used.size();
    c->ev_used.clear();
    *total_ms = t;
    return NTK_OK;
}

int ntk_accum_reset(ntk_ctx *c)
{
    if (!c) return NTK_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
    return NTK_OK;
}

static int minimizers_reduce_impl(ntk_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, const ntk_params *p, uint32_t w);

int ntk_reduce_device_quality(ntk_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, const ntk_params *p)
{
    if (!c) return NTK_ERR_BAD_ARG;
    Mode m;
    int rc = resolve_mode(p, true, &m);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if (p->flags & 0xFFu) return minimizers_reduce_impl(c, d_seq, d_qual, n, p, p->flags & 0xFFu);
    return run_scan(c, d_seq, n, p, m, true, nullptr, nullptr, nullptr, d_qual);
}

int ntk_reduce_device(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p)
{
    return ntk_reduce_device_quality(c, d_seq, nullptr, n, p);
}

int ntk_accum_read(ntk_ctx *c, ntk_result *out)
{
    if (!c || !out) return NTK_ERR_BAD_ARG;
    uint64_t *h = (uint64_t *)c->h_pinned;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(h, c->d_acc, NTK_ACC_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    out->n_total = h[NTK_ACC_N_TOTAL]; out->n_fwd = h[NTK_ACC_N_FWD]; out->n_rc = h[NTK_ACC_N_RC];
    out->sum = h[NTK_ACC_SUM]; out->xr = h[NTK_ACC_XOR];
    memcpy(out->hist, h + NTK_ACC_HIST, sizeof(out->hist));
    out->n_undigested = h[NTK_ACC_UNDIGESTED];
    return NTK_OK;
}

int ntk_accum_device_ptr(ntk_ctx *c, uint64_t **d_words)
{
    if (!c || !d_words) return NTK_ERR_BAD_ARG;
    *d_words = c->d_acc;
    return NTK_OK;
}

int ntk_accum_bind_device(ntk_ctx *c, uint64_t *d_words)
{
    if (!c || ((uintptr_t)d_words & 7)) return NTK_ERR_BAD_ARG;
    c->d_acc = d_words ? d_words : c->d_acc_own;
    return NTK_OK;
}

int ntk_materialize_device_quality(ntk_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, const ntk_params *p,
                                   uint64_t *d_values, uint16_t *d_valid16, uint16_t *d_rc16)
{
    if (!c || !d_valid16 || !d_rc16) return NTK_ERR_BAD_ARG;
    Mode m;
    int rc = resolve_mode(p, true, &m);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    return run_scan(c, d_seq, n, p, m, false, d_values, d_valid16, d_rc16, d_qual);
}

int ntk_materialize_device(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p,
                           uint64_t *d_values, uint16_t *d_valid16, uint16_t *d_rc16)
{
    return ntk_materialize_device_quality(c, d_seq, nullptr, n, p, d_values, d_valid16, d_rc16);
}

/* ---- pinned batches ------------------------------------------------------------------------ */

int ntk_batch_acquire(ntk_ctx *c, uint64_t max_bytes, uint64_t max_records, ntk_batch **out)
{
    if (!c || !out || max_bytes == 0) return NTK_ERR_BAD_ARG;
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    const uint64_t want_bytes = (max_bytes + 1023) & ~(uint64_t)1023, want_records = max_records ? max_records : 1;
    {   // a pooled batch that is large enough, and not more than twice as large as asked for
        std::lock_guard<std::mutex> g(c->pool_mu);
        for (size_t i = 0; i < c->pool.size(); i++) {
            ntk_batch *p = c->pool[i];
            if (p->cap_bytes >= want_bytes && p->cap_bytes <= 2 * want_bytes && p->cap_records >= want_records) {
                c->pool[i] = c->pool.back(); c->pool.pop_back();
                c->pool_bytes -= p->cap_bytes;
                *out = p;
                return NTK_OK;
            }
        }
    }
    ntk_batch *b = new (std::nothrow) ntk_batch();
    if (!b) return NTK_ERR_NOMEM;
    b->cap_bytes = want_bytes;
    b->cap_records = want_records;
    if (hipHostMalloc((void **)&b->h_seq, b->cap_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&b->h_off, (b->cap_records + 1) * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&b->d_seq, b->cap_bytes) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming) != hipSuccess) {
        g_last_hip = (int)hipGetLastError();
        destroy_batch(b);
        return NTK_ERR_HIP;
    }
    b->h_off[0] = 0;
    b->device = c->device;
    *out = b;
    return NTK_OK;
}

static int batch_append_impl(ntk_batch *b, const uint8_t *seq, const uint8_t *qual, uint64_t n, uint32_t pre, uint32_t cutoff)
{
    if (!b || (!seq && n) || pre > NTK_PRE_NORMALIZE_IUPAC || b->in_flight) return NTK_ERR_BAD_ARG;
    if (b->n_records >= b->cap_records || b->n_bytes + n + 1 > b->cap_bytes) return NTK_ERR_CAPACITY;
    if (qual && !b->h_qual) {
        // quality stream: same capacity and offsets as the sequence stream; records appended without qualities read
        // as 0xFF (never below a cutoff)
        HIPCHK(hipSetDevice(b->device));
        HIPCHK(hipHostMalloc((void **)&b->h_qual, b->cap_bytes, hipHostMallocDefault));
        if (hipMalloc((void **)&b->d_qual, b->cap_bytes) != hipSuccess) {
            g_last_hip = (int)hipGetLastError();
            (void)hipHostFree(b->h_qual); b->h_qual = nullptr;
            return NTK_ERR_HIP;
        }
        memset(b->h_qual, 0xFF, b->n_bytes);
    }
    uint8_t *o = b->h_seq + b->n_bytes;
    uint8_t *oq = b->h_qual ? b->h_qual + b->n_bytes : nullptr;
    uint64_t w = 0;
    // the pre-step's deleted class is dropped here together with its quality byte - unless that quality is below the
    // cutoff: the reference masks (base, quality) pairs BEFORE normalize deletes anything (src/sequence.rs:285-296), so
    // such a byte has become an N by then and stays, as a break.  It is kept here and masked on the device.
    // Line by line (memchr for LF): a line without any other byte of the deleted class - the usual case, checked by a
    // branch-free reduction the compiler vectorises - is one memcpy; wrapped FASTA contigs cost a memchr + memcpy per line.
    const bool ws = pre != NTK_PRE_STRIP_RETURNS;
    bool plain = pre == NTK_PRE_NONE;
    if (!plain && n <= 4096) {
        // a short record (a read): ONE branch-free pass says whether any byte of the deleted class is in it at all - the usual answer is no,
        // and the record is one memcpy (the line-by-line walk below costs a memchr and a second pass per line)
        unsigned any = 0;
        for (uint64_t j = 0; j < n; j++) {
            const uint8_t ch = seq[j];
            any |= (unsigned)(ch == '\n') | (unsigned)(ch == '\r') | ((unsigned)ws & ((unsigned)(ch == ' ') | (unsigned)(ch == '\t')));
        }
        plain = !any;
    }
    if (plain) {
        memcpy(o, seq, n); w = n;
        if (oq) { if (qual) memcpy(oq, qual, n); else memset(oq, 0xFF, n); }
    } else {
        uint64_t i = 0;
        while (i < n) {
            const uint8_t *nl = (const uint8_t *)memchr(seq + i, '\n', n - i);
            const uint64_t end = nl ? (uint64_t)(nl - seq) : n;   // line = [i, end), then the LF (if any)
            unsigned any = 0;
            for (uint64_t j = i; j < end; j++) {
                const uint8_t ch = seq[j];
                any |= (unsigned)(ch == '\r') | ((unsigned)ws & ((unsigned)(ch == ' ') | (unsigned)(ch == '\t')));
            }
            if (!any) {
                memcpy(o + w, seq + i, end - i);
                if (oq) { if (qual) memcpy(oq + w, qual + i, end - i); else memset(oq + w, 0xFF, end - i); }
                w += end - i;
            } else {
                for (uint64_t j = i; j < end; j++) {
                    const uint8_t ch = seq[j];
                    if ((ch == '\r' || (ws && (ch == ' ' || ch == '\t'))) && !(qual && qual[j] < cutoff)) continue;
                    if (oq) oq[w] = qual ? qual[j] : (uint8_t)0xFF;
                    o[w++] = ch;
                }
            }
            if (nl && qual && qual[end] < cutoff) {   // a low-quality LF has become an N: it stays (see above)
                if (oq) oq[w] = qual[end];
                o[w++] = '\n';
            }
            i = end + 1;
        }
    }
    if (oq) oq[w] = 0xFF;
    o[w++] = '\n';
    b->n_bytes += w;
    b->n_records += 1;
    b->h_off[b->n_records] = b->n_bytes;
    if (qual) { b->has_qual = true; b->qual_cutoff = cutoff; }
    return NTK_OK;
}

int ntk_batch_append(ntk_batch *b, const uint8_t *seq, uint64_t n, uint32_t pre)
{
    return batch_append_impl(b, seq, nullptr, n, pre, 0);
}

int ntk_batch_append_quality(ntk_batch *b, const uint8_t *seq, const uint8_t *qual, uint64_t n, uint32_t pre, uint32_t cutoff)
{
    if ((!qual && n) || cutoff < 1 || cutoff > 255) return NTK_ERR_BAD_ARG;
    if (b && b->has_qual && b->qual_cutoff != cutoff) return NTK_ERR_BAD_ARG;  // one cutoff per fill
    return batch_append_impl(b, seq, qual ? qual : (const uint8_t *)"", n, pre, cutoff);
}

int ntk_batch_buffers(ntk_batch *b, uint8_t **seq, uint64_t **offsets, uint64_t *n_bytes, uint64_t *n_records)
{
    if (!b) return NTK_ERR_BAD_ARG;
    if (seq) *seq = b->h_seq;
    if (offsets) *offsets = b->h_off;
    if (n_bytes) *n_bytes = b->n_bytes;
    if (n_records) *n_records = b->n_records;
    return NTK_OK;
}

int ntk_batch_submit(ntk_ctx *c, ntk_batch *b, const ntk_params *p)
{
    if (!c || !b || b->in_flight) return NTK_ERR_BAD_ARG;
    Mode m;
    int rc = resolve_mode(p, true, &m);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if (b->n_bytes) {
        const uint64_t padded = (b->n_bytes + 15) & ~(uint64_t)15;
        for (uint64_t i = b->n_bytes; i < padded; i++) b->h_seq[i] = '\n';
        hipStream_t cs = (c->copy_streams > 1 && (c->copy_rr++ & 1u)) ? c->copy_stream2 : c->copy_stream;
        HIPCHK(hipMemcpyAsync(b->d_seq, b->h_seq, padded, hipMemcpyHostToDevice, cs));
        // the quality stream travels only when a cutoff is set and some record of this fill carries qualities
        const uint8_t *d_qual = nullptr;
        if (b->has_qual && quality_cutoff(p) != b->qual_cutoff) return NTK_ERR_BAD_ARG;
        if (quality_cutoff(p) && b->has_qual) {
            for (uint64_t i = b->n_bytes; i < padded; i++) b->h_qual[i] = 0xFF;
            HIPCHK(hipMemcpyAsync(b->d_qual, b->h_qual, padded, hipMemcpyHostToDevice, cs));
            d_qual = b->d_qual;
        }
        HIPCHK(hipEventRecord(b->ev_copied, cs));
        HIPCHK(hipStreamWaitEvent(c->stream, b->ev_copied, 0));
        rc = (p->flags & 0xFFu) ? minimizers_reduce_impl(c, b->d_seq, d_qual, b->n_bytes, p, p->flags & 0xFFu)
                               : run_scan(c, b->d_seq, b->n_bytes, p, m, true, nullptr, nullptr, nullptr, d_qual);
        if (rc) return rc;
    }
    else if (p->flags & NTK_FLAG_RESET) HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
    HIPCHK(hipEventRecord(b->ev_done, c->stream));
    b->in_flight = true;
    return NTK_OK;
}

int ntk_batch_wait(ntk_ctx *c, ntk_batch *b)
{
    if (!c || !b) return NTK_ERR_BAD_ARG;
    if (b->in_flight) {
        // query + sleep rather than hipEventSynchronize: with dozens of parser threads blocked inside the runtime at once, the
        // submitting thread's enqueue calls slowed down 2 - 3 x (48+ threads: 16 instead of 39 Gbases/s, profiles/r02e/pipeline.txt).
        // NTK_OPT_BATCH_WAIT_POLL_US = NTK_POLL_BLOCK restores the blocking wait (values are clamped to 10 s).
        const long poll_us = (long)c->wait_poll_us;
        if (poll_us > 0) {
            for (;;) {
                const hipError_t q = hipEventQuery(b->ev_done);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { g_last_hip = (int)q; return NTK_ERR_HIP; }
                struct timespec ts = {poll_us / 1000000L, (poll_us % 1000000L) * 1000L};   // tv_nsec stays below 1e9
                nanosleep(&ts, nullptr);
            }
        } else HIPCHK(hipEventSynchronize(b->ev_done));
        b->in_flight = false;
    }
    b->n_bytes = 0; b->n_records = 0; b->h_off[0] = 0; b->has_qual = false;
    return NTK_OK;
}

void ntk_batch_release(ntk_ctx *c, ntk_batch *b)
{
    if (!b) return;
    if (c) (void)hipSetDevice(c->device);
    if (b->in_flight && b->ev_done) (void)hipEventSynchronize(b->ev_done);
    b->in_flight = false;
    if (c && b->h_seq && b->h_off && b->d_seq) {  // keep it for the next acquire (bounded: 256 batches / 8 GiB pinned)
        std::lock_guard<std::mutex> g(c->pool_mu);
        if (c->pool.size() < 256 && c->pool_bytes + b->cap_bytes <= ((uint64_t)8 << 30)) {
            b->n_bytes = 0; b->n_records = 0; b->h_off[0] = 0; b->has_qual = false; b->qual_cutoff = 0;
            c->pool.push_back(b);
            c->pool_bytes += b->cap_bytes;
            return;
        }
    }
    destroy_batch(b);
}

/* ---- compat face ------------------------------------------------------------------------------ */

static int compact_with_lut(ntk_ctx *c, const uint8_t *seq, uint64_t n, const uint16_t *d_lut,
                            uint8_t *out, uint64_t *out_len, uint32_t *flags_out)
{
    flags_out[0] = flags_out[1] = 0;
    *out_len = 0;
    if (n == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    const uint32_t nblocks = (uint32_t)((n + kCompactBlockBytes - 1) / kCompactBlockBytes);
    int rc;
    if ((rc = ensure_scratch(c, 0, n))) return rc;                                   // input
    if ((rc = ensure_scratch(c, 1, n))) return rc;                                   // output
    if ((rc = ensure_scratch(c, 2, (size_t)nblocks * 4 + 16))) return rc;            // kept per block
    if ((rc = ensure_scratch(c, 3, (size_t)nblocks * 8 + 64))) return rc;            // offsets + total + flags
    uint8_t *d_in = (uint8_t *)c->scratch[0].p, *d_out = (uint8_t *)c->scratch[1].p;
    uint32_t *d_kept = (uint32_t *)c->scratch[2].p;
    uint64_t *d_off = (uint64_t *)c->scratch[3].p;
    uint64_t *d_total = d_off + nblocks;
    uint32_t *d_flags = (uint32_t *)(d_total + 1);
    HIPCHK(hipMemcpyAsync(d_in, seq, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_total, 0, 16, c->stream));
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(kCompactThreads), 0, c->stream, (const uint8_t *)d_in, n, d_lut, d_kept, d_flags);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)d_kept, d_off, nblocks, d_total);
    hipLaunchKernelGGL(compact_write_kernel, dim3(nblocks), dim3(kCompactThreads), 0, c->stream, (const uint8_t *)d_in, n, d_lut, (const uint64_t *)d_off, d_out);
    HIPCHK(hipGetLastError());
    uint64_t *h = (uint64_t *)c->h_pinned;
    HIPCHK(hipMemcpyAsync(h, d_total, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint64_t total = h[0];
    memcpy(flags_out, h + 1, 8);
    if (total) {
        HIPCHK(hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *out_len = total;
    return NTK_OK;
}

int ntk_normalize(ntk_ctx *c, const uint8_t *seq, uint64_t n, int allow_iupac, uint8_t *out, uint64_t *out_len, int *changed)
{
    if (!c || (!seq && n) || (!out && n) || !out_len) return NTK_ERR_BAD_ARG;
    uint32_t flags[2];
    int rc = compact_with_lut(c, seq, n, c->d_lut + (allow_iupac ? 256 : 0), out, out_len, flags);
    if (rc) return rc;
    if (changed) *changed = flags[0] ? 1 : 0;
    return NTK_OK;
}

int ntk_strip_returns(ntk_ctx *c, const uint8_t *seq, uint64_t n, uint8_t *out, uint64_t *out_len, int *borrowed)
{
    if (!c || (!seq && n) || (!out && n) || !out_len) return NTK_ERR_BAD_ARG;
    uint32_t flags[2];
    int rc = compact_with_lut(c, seq, n, c->d_lut + 512, out, out_len, flags);
    if (rc) return rc;
    if (borrowed) *borrowed = flags[1] ? 0 : 1;
    return NTK_OK;
}

int ntk_reverse_complement(ntk_ctx *c, const uint8_t *seq, uint64_t n, uint8_t *out)
{
    if (!c || (!seq && n) || (!out && n)) return NTK_ERR_BAD_ARG;
    if (n == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, n))) return rc;
    if ((rc = ensure_scratch(c, 1, n))) return rc;
    uint8_t *d_in = (uint8_t *)c->scratch[0].p, *d_out = (uint8_t *)c->scratch[1].p;
    HIPCHK(hipMemcpyAsync(d_in, seq, n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(map_reverse_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream,
                       (const uint8_t *)d_in, d_out, n, (const uint16_t *)(c->d_lut + 768));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

int ntk_canonical_kmers(ntk_ctx *c, const uint8_t *seq, uint64_t n, uint32_t k, uint64_t *pos_out, uint8_t *is_rc_out,
                        uint64_t cap, uint64_t *count)
{
    // one record through the batched entry point: the items are compacted on the device, only they come back
    if (!c || (!seq && n) || !count) return NTK_ERR_BAD_ARG;
    const uint64_t offsets[2] = {0, n};
    return ntk_canonical_kmers_batch(c, seq, offsets, 1, k, nullptr, pos_out, is_rc_out, cap, count);
}

int ntk_bit_kmers(ntk_ctx *c, const uint8_t *seq, uint64_t n, uint32_t k, int canonical, uint64_t *pos_out,
                  uint64_t *val_out, uint8_t *was_rc_out, uint64_t cap, uint64_t *count)
{
    if (!c || (!seq && n) || !count) return NTK_ERR_BAD_ARG;
    const uint64_t offsets[2] = {0, n};
    return ntk_bit_kmers_batch(c, seq, offsets, 1, k, canonical, nullptr, pos_out, val_out, was_rc_out, cap, count);
}

/* ---- batched compat face ------------------------------------------------------------------------------------------- */
namespace {
// The items of Sequence::canonical_kmers / bit_kmers (reference src/sequence.rs:237-252) for a whole batch of records in ONE call.
// The batch is cut into chunks of <= kCompatChunkBytes packed bytes that run through a two-deep pipeline, each chunk owning one
// CompatBank (pinned staging + device buffers):
//     A(c):  pack chunk c into pinned memory (one break byte after each record) -> H2D -> scan -> valid / is_rc planes ->
//            count + scan of the items per 16 384-position block -> the chunk's item total to pinned memory        [ctx stream]
//     B(c):  (total known) scatter into dense (pos, value, flag) arrays + per-record counts                        [ctx stream]
//            -> D2H straight into the caller's arrays at the chunk's item offset                                   [copy stream]
// and the host issues A(c+1) before B(c): while the copy engine drains chunk c's items (the 9 B per item going back over PCIe are
// the bound of this face), the CPU packs chunk c+1 and the GPU scans it.  THREE banks rotate: a bank is free again when its
// chunk's D2H has finished, and with two the host would wait for D2H(c-1) - enqueued a moment ago - before it could pack chunk
// c+1 (measured: 3.85 ms per chunk = 2.35 ms of copies + 1.5 ms of packing in series, profiles/r03b/compat_trace.txt).
// Caller arrays may be pageable or pinned (ntk_pinned_alloc).
// (chunk size: ntk_ctx::compat_chunk, 16 MiB; NTK_OPT_COMPAT_CHUNK_BYTES lets the suites force many chunks out of small batches)
// The banks keep their staging and device buffers between calls (re-allocation costs milliseconds); what a call leaves behind above
// this many bytes per bank is released when it returns (a single 10-kb-record batch used to pin ~1.2 GiB until ctx_destroy).
constexpr size_t kBankKeepBytes = (size_t)512 << 20;   // a 16 MiB chunk of the bit path needs ~420 MiB per bank: kept; more than that: released
void bank_trim(ntk_ctx *c, size_t keep = kBankKeepBytes)
{
    for (CompatBank &b : c->bank) {
        size_t dev = 0;
        for (const Scratch &s : b.d) dev += s.bytes;
        if (dev > keep) for (Scratch &s : b.d) { if (s.p) (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
        if (b.h_stage_bytes > keep) { (void)hipHostFree(b.h_stage); b.h_stage = nullptr; b.h_stage_bytes = 0; }
    }
}

int bank_scratch(ntk_ctx *c, CompatBank &b, int slot, size_t bytes)
{
    if (bytes < 256) bytes = 256;
    Scratch &s = b.d[slot];
    if (s.bytes >= bytes) return NTK_OK;
    // (only this bank's own earlier work can still use the buffer, and that was waited for before the bank was re-used)
    if (s.p) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipStreamSynchronize(c->copy_stream)); HIPCHK(hipStreamSynchronize(c->down_stream)); HIPCHK(hipFree(s.p)); s.p = nullptr; s.bytes = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc(&s.p, want));
    s.bytes = want;
    return NTK_OK;
}

int bank_init(ntk_ctx *c, CompatBank &b)
{
    (void)c;
    if (b.ev_total) return NTK_OK;
    HIPCHK(hipEventCreateWithFlags(&b.ev_total, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b.ev_scattered, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void **)&b.h_total, 64, hipHostMallocDefault));
    return NTK_OK;
}

struct CompatJob {
    const uint8_t *seq; const uint64_t *offsets;
    uint32_t k; int kind;   // 0: byte path (raw-byte CanonicalKmers, any k <= 255), 1: bit path canonical, 2: bit path forward
    uint64_t *counts, *pos_out, *val_out; uint8_t *flag_out;
    uint64_t cap;
};

// A: records [r0, r0 + nrec) of the job into bank b, up to the item total
int compat_stage_a(ntk_ctx *c, CompatBank &b, const CompatJob &j, uint64_t r0, uint64_t nrec)
{
    int rc;
    const uint64_t n = j.offsets[r0 + nrec] - j.offsets[r0] + nrec;
    const size_t stage_bytes = (size_t)n + (size_t)(nrec + 1) * 8 + 64;
    if (b.h_stage_bytes < stage_bytes) {
        if (b.h_stage) { HIPCHK(hipHostFree(b.h_stage)); b.h_stage = nullptr; b.h_stage_bytes = 0; }
        HIPCHK(hipHostMalloc(&b.h_stage, stage_bytes + stage_bytes / 4, hipHostMallocDefault));
        b.h_stage_bytes = stage_bytes + stage_bytes / 4;
    }
    uint64_t *h_start = (uint64_t *)b.h_stage;                       // 8-byte aligned head
    uint8_t *h_seq = (uint8_t *)b.h_stage + (size_t)(nrec + 1) * 8;
    // record r of the chunk lands at offsets[r0 + r] - offsets[r0] + r (one break byte after every record before it): the
    // packing splits over threads without a prefix pass (one core packs ~5 GB/s of 150-byte records - less than the PCIe link
    // takes back - so a chunk is packed by up to ntk_ctx::pack_threads threads, default 8)
    const uint64_t o0 = j.offsets[r0];
    auto pack = [&](uint64_t ra, uint64_t rb) {
        for (uint64_t r = ra; r < rb; r++) {
            const uint64_t o = j.offsets[r0 + r], len = j.offsets[r0 + r + 1] - o, w = o - o0 + r;
            h_start[r] = w;
            if (len) memcpy(h_seq + w, j.seq + o, len);
            h_seq[w + len] = '\n';
        }
    };
    const unsigned nth = (unsigned)std::min<uint64_t>(c->pack_threads, n / (256 << 10) + 1);   // a thread per 256 KiB at least
    if (nth <= 1) pack(0, nrec);
    else {
        // a thread that cannot be created (thread / cgroup limit: std::system_error, bad_alloc) must not unwind across the C ABI:
        // the records it would have packed are packed here instead
        std::vector<std::thread> th;
        uint64_t done_to = nrec / nth;            // [0, done_to) is this thread's own share
        uint64_t spawned_from = nrec;             // [spawned_from, nrec) went to threads
        try {
            th.reserve(nth - 1);
            for (unsigned t = nth - 1; t >= 1; t--) {   // from the back, so that what is left over stays one contiguous range
                th.emplace_back(pack, nrec * t / nth, nrec * (t + 1) / nth);
                spawned_from = nrec * t / nth;
            }
        } catch (...) {}
        pack(0, spawned_from < done_to ? spawned_from : done_to);
        if (spawned_from > done_to) pack(done_to, spawned_from);
        for (auto &t : th) t.join();
    }
    h_start[nrec] = n;
    b.r0 = r0; b.nrec = nrec; b.n = n;
    const uint64_t nt = (n + 15) / 16 * 16;
    if ((rc = bank_scratch(c, b, 0, nt + 16))) return rc;
    if ((rc = bank_scratch(c, b, 5, (size_t)(nrec + 1) * 8))) return rc;
    if ((rc = bank_scratch(c, b, 1, j.kind == 0 ? nt : nt * 8))) return rc;
    if ((rc = bank_scratch(c, b, 2, nt / 8 + 16))) return rc;
    if ((rc = bank_scratch(c, b, 3, nt / 8 + 16))) return rc;
    b.n_words = (n + 15) >> 4;
    b.nblocks = (b.n_words + kCpBlockWords - 1) / kCpBlockWords;
    // one slot holds: block_items u32[nblocks] | block_off u64[nblocks] | total u64 | counts u64[nrec]
    b.o_off = ((size_t)b.nblocks * 4 + 7) & ~(size_t)7; b.o_total = b.o_off + (size_t)b.nblocks * 8; b.o_counts = b.o_total + 8;
    if ((rc = bank_scratch(c, b, 4, b.o_counts + (size_t)nrec * 8))) return rc;
    HIPCHK(hipMemcpyAsync(b.d[0].p, h_seq, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b.d[5].p, h_start, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, c->stream));
    uint16_t *d_v16 = (uint16_t *)b.d[2].p, *d_r16 = (uint16_t *)b.d[3].p;
    if (j.kind == 0) {
        // raw-byte comparison exactly as the reference (src/kmer.rs:84-129): any k <= 255, mixed case compares as bytes
        uint8_t *d_flags = (uint8_t *)b.d[1].p;
        hipLaunchKernelGGL(canonical_bytes_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream,
                           (const uint8_t *)b.d[0].p, n, j.k, (const uint16_t *)(c->d_lut + 768), d_flags, (const uint32_t *)nullptr);
        hipLaunchKernelGGL(pack_flags8_kernel, dim3(grid_for((n + 15) / 16, 256)), dim3(256), 0, c->stream, (const uint8_t *)d_flags, n, d_v16, d_r16);
        HIPCHK(hipGetLastError());
    } else {
        ntk_params p = {j.k, (uint32_t)(j.kind == 1 ? NTK_PATH_BITS_CANONICAL : NTK_PATH_BITS), NTK_PRE_NONE, 0};
        Mode m;
        if ((rc = resolve_mode(&p, true, &m))) return rc;
        if ((rc = run_scan(c, (const uint8_t *)b.d[0].p, n, &p, m, false, (uint64_t *)b.d[1].p, d_v16, d_r16))) return rc;
    }
    uint8_t *base = (uint8_t *)b.d[4].p;
    hipLaunchKernelGGL(cp_count_kernel, dim3((unsigned)b.nblocks), dim3(kCpThreads), 0, c->stream, d_v16, b.n_words, (uint32_t *)base);
    hipLaunchKernelGGL(cp_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)base, (uint64_t *)(base + b.o_off), b.nblocks,
                       (uint64_t *)(base + b.o_total));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(b.h_total, base + b.o_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(b.ev_total, c->stream));
    return NTK_OK;
}

// B: the chunk's items leave for the caller's arrays at item offset *base (capacity permitting); *base += the chunk's item total
int compat_stage_b(ntk_ctx *c, CompatBank &b, const CompatJob &j, uint64_t *base_items)
{
    int rc;
    HIPCHK(hipEventSynchronize(b.ev_total));
    const uint64_t m = *b.h_total, at = *base_items;
    const uint64_t take = at >= j.cap ? 0 : (m < j.cap - at ? m : j.cap - at);
    const bool values = j.kind != 0;
    if ((rc = bank_scratch(c, b, 6, (size_t)take * (8 + (values ? 8 : 0) + 1) + 64))) return rc;
    uint64_t *d_pos = (uint64_t *)b.d[6].p, *d_val = values ? d_pos + take : nullptr;
    uint8_t *d_flag = (uint8_t *)(d_pos + take + (values ? take : 0));
    uint8_t *sb = (uint8_t *)b.d[4].p;
    uint64_t *d_counts = (uint64_t *)(sb + b.o_counts);
    HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)b.nrec * 8, c->stream));
    hipLaunchKernelGGL(cp_scatter_kernel, dim3((unsigned)b.nblocks), dim3(kCpThreads), 0, c->stream, (const uint16_t *)b.d[2].p, (const uint16_t *)b.d[3].p,
                       values ? (const uint64_t *)b.d[1].p : nullptr, b.n_words, (const uint64_t *)(sb + b.o_off), (const uint64_t *)b.d[5].p, b.nrec,
                       j.kind == 0 ? 0u : j.k - 1, take, d_pos, d_val, d_flag, (unsigned long long *)d_counts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b.ev_scattered, c->stream));
    HIPCHK(hipStreamWaitEvent(c->copy_stream, b.ev_scattered, 0));
    if (j.counts) HIPCHK(hipMemcpyAsync(j.counts + b.r0, d_counts, (size_t)b.nrec * 8, hipMemcpyDeviceToHost, c->copy_stream));
    if (take && j.pos_out) HIPCHK(hipMemcpyAsync(j.pos_out + at, d_pos, (size_t)take * 8, hipMemcpyDeviceToHost, c->copy_stream));
    if (take && j.val_out && d_val) HIPCHK(hipMemcpyAsync(j.val_out + at, d_val, (size_t)take * 8, hipMemcpyDeviceToHost, c->copy_stream));
    if (take && j.flag_out) HIPCHK(hipMemcpyAsync(j.flag_out + at, d_flag, (size_t)take, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipEventRecord(b.ev_done, c->copy_stream));
    b.busy = true;
    *base_items = at + m;
    return NTK_OK;
}

int compat_batch(ntk_ctx *c, const CompatJob &j, uint64_t n_records, uint64_t *total)
{
    for (uint64_t r = 0; r < n_records; r++) if (j.offsets[r] > j.offsets[r + 1]) return NTK_ERR_BAD_ARG;
    int rc = NTK_OK;
    for (CompatBank &b : c->bank) if ((rc = bank_init(c, b))) return rc;
    uint64_t items = 0, r0 = 0;
    int cur = 0, prev = -1;
    const uint64_t chunk_bytes = c->compat_chunk;
    constexpr int kBanks = (int)(sizeof(c->bank) / sizeof(c->bank[0]));
    while (r0 < n_records && rc == NTK_OK) {
        // the chunk: records up to chunk_bytes packed bytes (at least one record)
        uint64_t r1 = r0 + 1;
        while (r1 < n_records && j.offsets[r1 + 1] - j.offsets[r0] + (r1 + 1 - r0) <= chunk_bytes) r1++;
        CompatBank &b = c->bank[cur];
        if (b.busy) { if (hipEventSynchronize(b.ev_done) != hipSuccess) { g_last_hip = (int)hipGetLastError(); rc = NTK_ERR_HIP; break; } b.busy = false; }
        if ((rc = compat_stage_a(c, b, j, r0, r1 - r0))) break;
        if (prev >= 0 && (rc = compat_stage_b(c, c->bank[prev], j, &items))) break;
        prev = cur; cur = (cur + 1) % kBanks; r0 = r1;
    }
    if (rc == NTK_OK && prev >= 0) rc = compat_stage_b(c, c->bank[prev], j, &items);
    // the call is synchronous: everything of it has landed (or, after an error, nothing of it is still reading the pinned
    // staging buffers or writing the caller's arrays) when it returns
    const hipError_t e1 = hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(c->copy_stream);
    for (CompatBank &b : c->bank) b.busy = false;
    bank_trim(c);
    if (rc != NTK_OK) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess) { g_last_hip = (int)(e1 != hipSuccess ? e1 : e2); return NTK_ERR_HIP; }
    *total = items;
    return items > j.cap ? NTK_ERR_CAPACITY : NTK_OK;
}

// ---- bit-plane form of the byte-path items -----------------------------------------------------------------------------------
// Sequence::canonical_kmers yields (pos, slice, is_rc) (reference src/kmer.rs:114-129); slice and pos follow from the window's place,
// so all the device has to return is, per window start, "emitted?" and "is_rc?": TWO BITS per sequence byte instead of nine bytes per
// item (the bound of ntk_canonical_kmers_batch: 48 GB/s of items over PCIe = 6.3 Gbases/s).  The caller's bytes are uploaded as they
// lie (no packing pass, no break bytes: record starts travel as a bitmap built on the device from the offsets), the raw-byte kernel
// of the compat face marks the windows, and the two planes come back: 1/4 byte per input byte.  Chunks of <= 16 MiB, the same three
// banks: upload and kernels of chunk i + 1 overlap the download of chunk i.  Each chunk starts on a 16-position word boundary of the
// planes; rec_bit[r] is the plane position of record r's first byte.
// kind 0: the byte path (CanonicalKmers, raw-byte compare, any k <= 255); 1 / 2: the bit path, canonical / forward-only (BitNuclKmer, k <= 32),
// with values != nullptr also the items' packed values, one u64 per plane position (0 where nothing is emitted): 8 more bytes per input byte
// come back then, and the download is the call's bound.
int compat_planes(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k, uint64_t *rec_bit,
                  uint16_t *valid16, uint16_t *rc16, uint64_t cap_words, uint64_t *n_words, uint64_t *total, int kind = 0, uint64_t *values = nullptr)
{
    for (uint64_t r = 0; r < n_records; r++) if (offsets[r] > offsets[r + 1]) return NTK_ERR_BAD_ARG;
    const uint64_t chunk_bytes = c->compat_chunk;
    constexpr int kBanks = (int)(sizeof(c->bank) / sizeof(c->bank[0]));
    // pass 1 (host): the chunks and the words they need
    struct Chunk { uint64_t r0, r1, wbase, nb; };
    std::vector<Chunk> chunks;
    uint64_t words = 0;
    try {
        for (uint64_t r0 = 0; r0 < n_records;) {
            uint64_t r1 = r0 + 1;
            while (r1 < n_records && offsets[r1 + 1] - offsets[r0] <= chunk_bytes) r1++;
            const uint64_t nb = offsets[r1] - offsets[r0];
            chunks.push_back({r0, r1, words, nb});
            words += (nb + 15) >> 4;
            r0 = r1;
        }
    } catch (...) { return NTK_ERR_NOMEM; }
    *n_words = words;
    if (words > cap_words) return NTK_ERR_CAPACITY;
    int rc = NTK_OK;
    for (CompatBank &b : c->bank) if ((rc = bank_init(c, b))) return rc;
    uint64_t items = 0;
    auto retire = [&](CompatBank &b) -> int {   // the bank's chunk has fully landed: its item total joins the sum
        if (!b.busy) return NTK_OK;
        if (hipEventSynchronize(b.ev_done) != hipSuccess) { g_last_hip = (int)hipGetLastError(); return NTK_ERR_HIP; }
        items += *b.h_total;
        b.busy = false;
        return NTK_OK;
    };
    // Three stages per chunk on three streams: upload [copy stream] -> bitmap + planes kernel [ctx stream] -> download [down stream]:
    // the uploads run back to back (the call is bound by them), the kernel and the download of chunk i hide behind the upload of
    // chunk i + 1.  A bank is re-used every third chunk, after its download has landed (retire()).
    auto hip_fail = [&](hipError_t e) { g_last_hip = (int)e; return NTK_ERR_HIP; };
    auto upload = [&](size_t ci) -> int {
        const Chunk &ch = chunks[ci];
        CompatBank &b = c->bank[ci % kBanks];
        int r;
        if ((r = retire(b))) return r;
        const uint64_t nrec = ch.r1 - ch.r0, nt = (ch.nb + 15) / 16 * 16;
        for (uint64_t q = ch.r0; q < ch.r1; q++) rec_bit[q] = ch.wbase * 16 + (offsets[q] - offsets[ch.r0]);
        if (ch.nb == 0) return NTK_OK;   // only empty records
        if ((r = bank_scratch(c, b, 0, nt + 16))) return r;
        if ((r = bank_scratch(c, b, 5, (size_t)(nrec + 1) * 8))) return r;
        if ((r = bank_scratch(c, b, 2, nt / 8 + 16))) return r;
        if ((r = bank_scratch(c, b, 3, nt / 8 + 16))) return r;
        if ((r = bank_scratch(c, b, 4, 64))) return r;
        if ((r = bank_scratch(c, b, 6, (size_t)((nt >> 5) + 2) * 4))) return r;
        if (values && (r = bank_scratch(c, b, 1, (size_t)nt * 8))) return r;
        hipError_t e;
        if ((e = hipMemcpyAsync(b.d[0].p, seq + offsets[ch.r0], ch.nb, hipMemcpyHostToDevice, c->copy_stream))) return hip_fail(e);
        if ((e = hipMemcpyAsync(b.d[5].p, offsets + ch.r0, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, c->copy_stream))) return hip_fail(e);
        if ((e = hipEventRecord(b.ev_total, c->copy_stream))) return hip_fail(e);   // "uploaded"
        return NTK_OK;
    };
    auto compute_and_download = [&](size_t ci) -> int {
        const Chunk &ch = chunks[ci];
        CompatBank &b = c->bank[ci % kBanks];
        if (ch.nb == 0) return NTK_OK;
        const uint64_t nrec = ch.r1 - ch.r0, nb = ch.nb, nt = (nb + 15) / 16 * 16, nw = nt >> 4, sb_words = (nt >> 5) + 2;
        hipError_t e;
        if ((e = hipStreamWaitEvent(c->stream, b.ev_total, 0))) return hip_fail(e);
        if ((e = hipMemsetAsync(b.d[6].p, 0, (size_t)sb_words * 4, c->stream))) return hip_fail(e);
        if ((e = hipMemsetAsync(b.d[4].p, 0, 8, c->stream))) return hip_fail(e);
        hipLaunchKernelGGL(mark_record_starts_kernel, dim3(grid_for(nrec, 256)), dim3(256), 0, c->stream, (const uint64_t *)b.d[5].p, nrec, nb,
                           (uint32_t *)b.d[6].p);
        const uint64_t tiles = (nb + kPlTile - 1) / kPlTile;
        const dim3 grid((unsigned)(tiles < (uint64_t)c->n_cu * 8 ? tiles : (uint64_t)c->n_cu * 8));
        if (kind == 0)
            hipLaunchKernelGGL(canonical_bytes_planes_kernel, grid, dim3(kPlThreads), 0, c->stream, (const uint8_t *)b.d[0].p, nb, nt + 16, k,
                               (const uint16_t *)(c->d_lut + 768), (const uint32_t *)b.d[6].p, sb_words, (uint16_t *)b.d[2].p, (uint16_t *)b.d[3].p,
                               (unsigned long long *)b.d[4].p);
        else if (kind == 1)
            hipLaunchKernelGGL(bit_kmers_planes_kernel<true>, grid, dim3(kPlThreads), 0, c->stream, (const uint8_t *)b.d[0].p, nb, nt + 16, k,
                               (const uint32_t *)b.d[6].p, sb_words, (uint16_t *)b.d[2].p, (uint16_t *)b.d[3].p, values ? (uint64_t *)b.d[1].p : nullptr,
                               (unsigned long long *)b.d[4].p);
        else
            hipLaunchKernelGGL(bit_kmers_planes_kernel<false>, grid, dim3(kPlThreads), 0, c->stream, (const uint8_t *)b.d[0].p, nb, nt + 16, k,
                               (const uint32_t *)b.d[6].p, sb_words, (uint16_t *)b.d[2].p, (uint16_t *)b.d[3].p, values ? (uint64_t *)b.d[1].p : nullptr,
                               (unsigned long long *)b.d[4].p);
        if ((e = hipGetLastError())) return hip_fail(e);
        if ((e = hipMemcpyAsync(b.h_total, b.d[4].p, 8, hipMemcpyDeviceToHost, c->stream))) return hip_fail(e);
        if ((e = hipEventRecord(b.ev_scattered, c->stream))) return hip_fail(e);
        if ((e = hipStreamWaitEvent(c->down_stream, b.ev_scattered, 0))) return hip_fail(e);
        if ((e = hipMemcpyAsync(valid16 + ch.wbase, b.d[2].p, (size_t)nw * 2, hipMemcpyDeviceToHost, c->down_stream))) return hip_fail(e);
        if ((e = hipMemcpyAsync(rc16 + ch.wbase, b.d[3].p, (size_t)nw * 2, hipMemcpyDeviceToHost, c->down_stream))) return hip_fail(e);
        if (values && (e = hipMemcpyAsync(values + ch.wbase * 16, b.d[1].p, (size_t)nw * 16 * 8, hipMemcpyDeviceToHost, c->down_stream))) return hip_fail(e);
        if ((e = hipEventRecord(b.ev_done, c->down_stream))) return hip_fail(e);
        b.busy = true;
        return NTK_OK;
    };
    if (!chunks.empty()) rc = upload(0);
    for (size_t ci = 0; ci < chunks.size() && rc == NTK_OK; ci++) {
        if (ci + 1 < chunks.size() && (rc = upload(ci + 1))) break;
        rc = compute_and_download(ci);
    }
    rec_bit[n_records] = words * 16;
    const hipError_t e0 = hipStreamSynchronize(c->down_stream);
    const hipError_t e1 = e0 != hipSuccess ? e0 : hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(c->copy_stream);
    if (rc == NTK_OK) for (CompatBank &b : c->bank) if ((rc = retire(b))) break;
    for (CompatBank &b : c->bank) b.busy = false;
    bank_trim(c);
    if (rc != NTK_OK) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess) { g_last_hip = (int)(e1 != hipSuccess ? e1 : e2); return NTK_ERR_HIP; }
    *total = items;
    return NTK_OK;
}
// sequence::minimizer per record (ntk_minimizer_batch): the same three stages on three streams and three banks as compat_planes - upload
// [copy stream] -> wave-per-record kernel [ctx stream] -> download of the chunk's minimizers / window starts / strands [down stream].
int minimizer_batch_impl(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t m, uint8_t *out, uint64_t *pos_out,
                         uint8_t *is_rc_out)
{
    constexpr uint64_t kLongRecord = 1ull << 16;
    constexpr int kBanks = (int)(sizeof(c->bank) / sizeof(c->bank[0]));
    const uint64_t chunk_bytes = c->compat_chunk;
    struct Chunk { uint64_t r0, r1, nb; };
    std::vector<Chunk> chunks;
    try {
        for (uint64_t r0 = 0; r0 < n_records;) {
            uint64_t r1 = r0 + 1;
            while (r1 < n_records && offsets[r1 + 1] - offsets[r0] <= chunk_bytes) r1++;
            chunks.push_back({r0, r1, offsets[r1] - offsets[r0]});
            r0 = r1;
        }
    } catch (...) { return NTK_ERR_NOMEM; }
    int rc = NTK_OK;
    for (CompatBank &b : c->bank) if ((rc = bank_init(c, b))) return rc;
    const uint16_t *comp = (const uint16_t *)(c->d_lut + 768);
    auto hip_fail = [&](hipError_t e) { g_last_hip = (int)e; return NTK_ERR_HIP; };
    auto retire = [&](CompatBank &b) -> int {
        if (!b.busy) return NTK_OK;
        if (hipEventSynchronize(b.ev_done) != hipSuccess) { g_last_hip = (int)hipGetLastError(); return NTK_ERR_HIP; }
        b.busy = false;
        return NTK_OK;
    };
    auto upload = [&](size_t ci) -> int {
        const Chunk &ch = chunks[ci];
        CompatBank &b = c->bank[ci % kBanks];
        const uint64_t nrec = ch.r1 - ch.r0;
        int r;
        if ((r = retire(b))) return r;
        if ((r = bank_scratch(c, b, 0, ch.nb + 16))) return r;
        if ((r = bank_scratch(c, b, 5, (size_t)(nrec + 1) * 8))) return r;
        if ((r = bank_scratch(c, b, 2, (size_t)nrec * m))) return r;
        if ((r = bank_scratch(c, b, 3, (size_t)nrec * 8))) return r;
        if ((r = bank_scratch(c, b, 1, (size_t)nrec))) return r;
        if ((r = bank_scratch(c, b, 4, 64))) return r;
        hipError_t e;
        if ((e = hipMemcpyAsync(b.d[0].p, seq + offsets[ch.r0], ch.nb, hipMemcpyHostToDevice, c->copy_stream))) return hip_fail(e);
        if ((e = hipMemcpyAsync(b.d[5].p, offsets + ch.r0, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, c->copy_stream))) return hip_fail(e);
        if ((e = hipEventRecord(b.ev_total, c->copy_stream))) return hip_fail(e);
        return NTK_OK;
    };
    auto compute_and_download = [&](size_t ci) -> int {
        const Chunk &ch = chunks[ci];
        CompatBank &b = c->bank[ci % kBanks];
        const uint64_t nrec = ch.r1 - ch.r0;
        hipError_t e;
        if ((e = hipStreamWaitEvent(c->stream, b.ev_total, 0))) return hip_fail(e);
        if ((e = hipMemsetAsync(b.d[4].p, 0xFF, 8, c->stream))) return hip_fail(e);
        const uint64_t blocks = (nrec + 3) / 4;
        hipLaunchKernelGGL(minimizer_batch_kernel, dim3((unsigned)(blocks < (uint64_t)c->n_cu * 16 ? blocks : (uint64_t)c->n_cu * 16)), dim3(256), 0, c->stream,
                           (const uint8_t *)b.d[0].p, (const uint64_t *)b.d[5].p, nrec, m, kLongRecord, comp, (uint8_t *)b.d[2].p, (uint64_t *)b.d[3].p,
                           (uint8_t *)b.d[1].p, (unsigned long long *)b.d[4].p);
        for (uint64_t r = ch.r0; r < ch.r1; r++) {   // the rare long record: the one-block kernel of ntk_minimizer, on the uploaded bytes
            const uint64_t n = offsets[r + 1] - offsets[r];
            if (n <= kLongRecord) continue;
            const uint8_t *rec = (const uint8_t *)b.d[0].p + (offsets[r] - offsets[ch.r0]);
            uint64_t *d_best = (uint64_t *)b.d[4].p + 1;
            hipLaunchKernelGGL(minimizer_bytes_kernel, dim3(1), dim3(1024), 0, c->stream, rec, n, m, comp, d_best);
            hipLaunchKernelGGL(minimizer_emit_record_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, rec, n, m, comp, (const uint64_t *)d_best, r - ch.r0,
                               (uint8_t *)b.d[2].p, (uint64_t *)b.d[3].p, (uint8_t *)b.d[1].p);
        }
        if ((e = hipGetLastError())) return hip_fail(e);
        if ((e = hipEventRecord(b.ev_scattered, c->stream))) return hip_fail(e);
        if ((e = hipStreamWaitEvent(c->down_stream, b.ev_scattered, 0))) return hip_fail(e);
        if ((e = hipMemcpyAsync(out + ch.r0 * m, b.d[2].p, (size_t)nrec * m, hipMemcpyDeviceToHost, c->down_stream))) return hip_fail(e);
        if (pos_out && (e = hipMemcpyAsync(pos_out + ch.r0, b.d[3].p, (size_t)nrec * 8, hipMemcpyDeviceToHost, c->down_stream))) return hip_fail(e);
        if (is_rc_out && (e = hipMemcpyAsync(is_rc_out + ch.r0, b.d[1].p, (size_t)nrec, hipMemcpyDeviceToHost, c->down_stream))) return hip_fail(e);
        if ((e = hipEventRecord(b.ev_done, c->down_stream))) return hip_fail(e);
        b.busy = true;
        return NTK_OK;
    };
    if (!chunks.empty()) rc = upload(0);
    for (size_t ci = 0; ci < chunks.size() && rc == NTK_OK; ci++) {
        if (ci + 1 < chunks.size() && (rc = upload(ci + 1))) break;
        rc = compute_and_download(ci);
    }
    const hipError_t e0 = hipStreamSynchronize(c->down_stream);
    const hipError_t e1 = e0 != hipSuccess ? e0 : hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(c->copy_stream);
    for (CompatBank &b : c->bank) b.busy = false;
    bank_trim(c);
    if (rc != NTK_OK) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess) { g_last_hip = (int)(e1 != hipSuccess ? e1 : e2); return NTK_ERR_HIP; }
    return NTK_OK;
}
}  // namespace

int ntk_canonical_kmers_batch_planes(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k,
                                     uint64_t *rec_bit, uint16_t *valid16, uint16_t *rc16, uint64_t cap_words, uint64_t *n_words,
                                     uint64_t *total)
{
    if (!c || !offsets || !rec_bit || !n_words || !total || (!seq && offsets[n_records] > offsets[0])) return NTK_ERR_BAD_ARG;
    if (k < 1 || k > 255) return NTK_ERR_BAD_K;
    *total = 0; *n_words = 0;
    rec_bit[0] = 0;
    if (n_records == 0) return NTK_OK;
    if ((!valid16 || !rc16) && cap_words) return NTK_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(c->device));
    return compat_planes(c, seq, offsets, n_records, k, rec_bit, valid16, rc16, cap_words, n_words, total);
}

int ntk_bit_kmers_batch_planes(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k, int canonical,
                               uint64_t *rec_bit, uint16_t *valid16, uint16_t *rc16, uint64_t *values, uint64_t cap_words, uint64_t *n_words,
                               uint64_t *total)
{
    if (!c || !offsets || !rec_bit || !n_words || !total || (!seq && offsets[n_records] > offsets[0])) return NTK_ERR_BAD_ARG;
    if (k < 1 || k > 32) return NTK_ERR_BAD_K;
    *total = 0; *n_words = 0;
    rec_bit[0] = 0;
    if (n_records == 0) return NTK_OK;
    if ((!valid16 || !rc16) && cap_words) return NTK_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(c->device));
    return compat_planes(c, seq, offsets, n_records, k, rec_bit, valid16, rc16, cap_words, n_words, total, canonical ? 1 : 2, values);
}

int ntk_bit_kmers_batch(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k, int canonical,
                        uint64_t *counts, uint64_t *pos_out, uint64_t *val_out, uint8_t *was_rc_out, uint64_t cap, uint64_t *total)
{
    if (!c || !offsets || !total || (!seq && offsets[n_records] > offsets[0])) return NTK_ERR_BAD_ARG;
    if (k < 1 || k > 32) return NTK_ERR_BAD_K;
    *total = 0;
    if (counts) memset(counts, 0, (size_t)n_records * 8);
    if (n_records == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    const CompatJob j = {seq, offsets, k, canonical ? 1 : 2, counts, pos_out, val_out, was_rc_out, cap};
    return compat_batch(c, j, n_records, total);
}

int ntk_canonical_kmers_batch(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k,
                              uint64_t *counts, uint64_t *pos_out, uint8_t *is_rc_out, uint64_t cap, uint64_t *total)
{
    if (!c || !offsets || !total || (!seq && offsets[n_records] > offsets[0])) return NTK_ERR_BAD_ARG;
    if (k < 1 || k > 255) return NTK_ERR_BAD_K;
    *total = 0;
    if (counts) memset(counts, 0, (size_t)n_records * 8);
    if (n_records == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    const CompatJob j = {seq, offsets, k, 0, counts, pos_out, nullptr, is_rc_out, cap};
    return compat_batch(c, j, n_records, total);
}

/* Page-locked host memory for the arrays the batched calls read and fill (what a Rust host would back its Vecs with): copies to
 * and from it run at the PCIe rate without a staging pass. */
int ntk_pinned_alloc(uint64_t bytes, void **out)
{
    if (!out || bytes == 0) return NTK_ERR_BAD_ARG;
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return NTK_OK;
}
void ntk_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

/* Gives back what the batched compat calls keep between calls (three banks of pinned staging and device buffers: up to ~180 MiB each
 * on the byte path, ~420 MiB on the bit path) and the ctx's scratch buffers.  The next call re-allocates. */
int ntk_ctx_trim(ntk_ctx *c)
{
    if (!c) return NTK_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->down_stream));
    bank_trim(c, 0);
    for (Scratch &s : c->scratch) { if (s.p) (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
    return NTK_OK;
}

/* ---- minimizers, quality mask --------------------------------------------------------------------------------- */

static int minimizers_reduce_impl(ntk_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, const ntk_params *p, uint32_t w);

int ntk_minimizers_reduce_device(ntk_ctx *c, const uint8_t *d_seq, uint64_t n, const ntk_params *p, uint32_t w)
{
    return minimizers_reduce_impl(c, d_seq, nullptr, n, p, w);
}

static int minimizers_reduce_impl(ntk_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, const ntk_params *p, uint32_t w)
{
    if (!c || w < 1 || w > 256) return NTK_ERR_BAD_ARG;
    Mode m;
    int rc = resolve_mode(p, true, &m);
    if (rc) return rc;
    if (!m.canon) return NTK_ERR_BAD_ARG;
    if (m.raw_bytes) return p->k > 32 ? NTK_ERR_BAD_K : NTK_ERR_UNSUPPORTED;   // windowed minimizers are defined on normalised input (values, not raw bytes)
    HIPCHK(hipSetDevice(c->device));
    // fused build (one pass, nothing written to HBM) where one exists - with a quality stream: the quality-masked builds
    if (n) {
        const bool masked = d_qual && quality_cutoff(p);
        const void *fn = (c->route_off & NTK_ROUTE_NO_REGFUSED) ? nullptr : pick_scan_min(m, p->k, w, masked);   // (route bits: ntk_ctx_set_option)
        if (fn)
            return run_scan(c, d_seq, n, p, m, true, nullptr, nullptr, nullptr, masked ? d_qual : nullptr, fn, p->k + w - 1 > 32 ? 3 : 2);
        // every other (k <= 31, w <= 49): the generic fused kernel (one pass as well, run-time k and w)
        if (p->k <= 31 && w <= 49 && !(c->route_off & NTK_ROUTE_NO_GENERIC))
            return run_min_scan(c, d_seq, n, p, m, w, masked ? d_qual : nullptr);
    }
    if (p->flags & NTK_FLAG_RESET) HIPCHK(hipMemsetAsync(c->d_acc, 0, NTK_ACC_WORDS * sizeof(uint64_t), c->stream));
    if (n == 0) return NTK_OK;
    // Long inputs are scanned in chunks so that the scratch planes stay bounded (8 B per position: 2 GiB for the default
    // 256 MiB chunk instead of 80 GB for a 10 GB batch).  A chunk is scanned together with the w+k-2 bytes of left context
    // before it (start rounded down to the 16-byte alignment of the scan); only windows ENDING inside the chunk are counted.
    const uint64_t chunk = c->minimizer_chunk;
    const uint64_t back = (uint64_t)w + p->k - 2;
    const int blocks = c->n_cu * 4;  // ~35 KiB of LDS per block
    if ((rc = ensure_partials(c, blocks))) return rc;
    ScanArgs a; memset(&a, 0, sizeof(a)); scan_args_set_k(a, p->k);
    for (uint64_t c0 = 0; c0 < n; c0 += chunk) {
        const uint64_t c1 = c0 + chunk < n ? c0 + chunk : n;
        const uint64_t start = c0 > back ? (c0 - back) & ~(uint64_t)15 : 0;
        const uint64_t n_sub = c1 - start, nt = (n_sub + 15) / 16 * 16;
        if ((rc = ensure_scratch(c, 3, nt * 8))) return rc;
        if ((rc = ensure_scratch(c, 4, nt / 8 + 16))) return rc;
        if ((rc = ensure_scratch(c, 5, nt / 8 + 16))) return rc;
        uint64_t *d_val = (uint64_t *)c->scratch[3].p;
        uint16_t *d_v16 = (uint16_t *)c->scratch[4].p, *d_r16 = (uint16_t *)c->scratch[5].p;
        if ((rc = run_scan(c, d_seq + start, n_sub, p, m, false, d_val, d_v16, d_r16, d_qual ? d_qual + start : nullptr))) return rc;
#define NTK_WM(WW)                                                                                                    \
    case WW:                                                                                                          \
        hipLaunchKernelGGL(window_min_reduce_kernel<WW>, dim3(blocks), dim3(kWmThreads), 0, c->stream, (const uint64_t *)d_val, \
                           (const uint16_t *)d_v16, (const uint16_t *)d_r16, n_sub, w, a.bin_shift, c->d_part_hist,    \
                           c->d_part_scalars, c0 - start);                                                            \
        break;
        switch (w <= 16 ? w : 0u) {   // compile-time windows up to 16 (register van Herk), the run-time walk otherwise
            NTK_WM(2) NTK_WM(3) NTK_WM(4) NTK_WM(5) NTK_WM(6) NTK_WM(7) NTK_WM(8) NTK_WM(9) NTK_WM(10) NTK_WM(11) NTK_WM(12)
            NTK_WM(13) NTK_WM(14) NTK_WM(15) NTK_WM(16)
        default:
            hipLaunchKernelGGL(window_min_reduce_kernel<0>, dim3(blocks), dim3(kWmThreads), 0, c->stream, (const uint64_t *)d_val,
                               (const uint16_t *)d_v16, (const uint16_t *)d_r16, n_sub, w, a.bin_shift, c->d_part_hist,
                               c->d_part_scalars, c0 - start);
        }
#undef NTK_WM
        hipLaunchKernelGGL(fold_kernel, dim3(kFoldBlocks), dim3(kFoldThreads), 0, c->stream,
                           (const uint32_t *)c->d_part_hist, (const uint64_t *)c->d_part_scalars, blocks, c->d_acc);
        HIPCHK(hipGetLastError());
    }
    return NTK_OK;
}

int ntk_minimizer(ntk_ctx *c, const uint8_t *seq, uint64_t n, uint32_t m, uint8_t *out)
{
    if (!c || !seq || !out || m < 1 || n < m) return NTK_ERR_BAD_ARG;  // the reference panics on n < m (sequence.rs:141)
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, n))) return rc;
    if ((rc = ensure_scratch(c, 1, (size_t)m + 64))) return rc;
    uint8_t *d_in = (uint8_t *)c->scratch[0].p, *d_out = (uint8_t *)c->scratch[1].p;
    uint64_t *d_best = (uint64_t *)(d_out + ((m + 15) & ~15u));
    HIPCHK(hipMemcpyAsync(d_in, seq, n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(minimizer_bytes_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint8_t *)d_in, n, m,
                       (const uint16_t *)(c->d_lut + 768), d_best);
    hipLaunchKernelGGL(minimizer_emit_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, (const uint8_t *)d_in, n, m,
                       (const uint16_t *)(c->d_lut + 768), (const uint64_t *)d_best, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

/* sequence::minimizer for every record of a reader batch in one call: chunks of <= 16 MiB uploaded back to back, one wave per record (one
 * block for a record beyond 64 KiB), the chunk's n x m bytes (+ the optional window starts and strands) downloaded behind the next upload. */
int ntk_minimizer_batch(ntk_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t m, uint8_t *out, uint64_t *pos_out,
                        uint8_t *is_rc_out, uint64_t *bad_record)
{
    if (!c || !offsets || m < 1 || (n_records && !out)) return NTK_ERR_BAD_ARG;
    if (bad_record) *bad_record = ~0ull;
    if (n_records == 0) return NTK_OK;
    for (uint64_t r = 0; r < n_records; r++) {
        if (offsets[r] > offsets[r + 1]) return NTK_ERR_BAD_ARG;
        if (offsets[r + 1] - offsets[r] < m) {   // the reference panics on a sequence shorter than the length asked for (src/sequence.rs:141)
            if (bad_record) *bad_record = r;
            return NTK_ERR_BAD_ARG;
        }
    }
    if (!seq) return NTK_ERR_BAD_ARG;   // (every record holds at least m >= 1 bytes)
    HIPCHK(hipSetDevice(c->device));
    return minimizer_batch_impl(c, seq, offsets, n_records, m, out, pos_out, is_rc_out);
}

int ntk_canonical(ntk_ctx *c, const uint8_t *seq, uint64_t n, uint8_t *out, int *was_rc)
{
    if (!c || (!seq && n) || (!out && n)) return NTK_ERR_BAD_ARG;
    if (was_rc) *was_rc = 0;
    if (n == 0) return NTK_OK;
    if (n > 0xFFFFFFFFull) return NTK_ERR_UNSUPPORTED;
    // the smallest length-n substring over both strands IS the canonical form (one candidate per strand)
    const int rc = ntk_minimizer(c, seq, n, (uint32_t)n, out);
    if (rc == NTK_OK && was_rc) *was_rc = memcmp(out, seq, n) != 0;
    return rc;
}

int ntk_bit_minimizers(ntk_ctx *c, const uint64_t *values, uint64_t n, uint32_t k, uint32_t m, uint64_t *out)
{
    if (!c || (!values && n) || (!out && n)) return NTK_ERR_BAD_ARG;
    if (k < 1 || k > 32 || m < 1 || m > k) return NTK_ERR_BAD_K;
    if (n == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, n * 8))) return rc;
    if ((rc = ensure_scratch(c, 1, n * 8))) return rc;
    uint64_t *d_in = (uint64_t *)c->scratch[0].p, *d_out = (uint64_t *)c->scratch[1].p;
    HIPCHK(hipMemcpyAsync(d_in, values, n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bit_minimizer_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const uint64_t *)d_in, n, k, m, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

int ntk_bit_canonical(ntk_ctx *c, const uint64_t *values, uint64_t n, uint32_t k, int canonical, uint64_t *out, uint8_t *was_rc_out)
{
    if (!c || (n && (!values || !out || (canonical && !was_rc_out)))) return NTK_ERR_BAD_ARG;
    if (k < 1 || k > 32) return NTK_ERR_BAD_K;
    if (n == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, n * 8))) return rc;
    if ((rc = ensure_scratch(c, 1, n * 8))) return rc;
    if ((rc = ensure_scratch(c, 2, n))) return rc;
    uint64_t *d_in = (uint64_t *)c->scratch[0].p, *d_out = (uint64_t *)c->scratch[1].p;
    uint8_t *d_f = (uint8_t *)c->scratch[2].p;
    HIPCHK(hipMemcpyAsync(d_in, values, n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bit_canonical_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const uint64_t *)d_in, n, k, canonical, d_out, d_f);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, c->stream));
    if (canonical) HIPCHK(hipMemcpyAsync(was_rc_out, d_f, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

int ntk_quality_mask(ntk_ctx *c, const uint8_t *seq, const uint8_t *qual, uint64_t n, uint8_t score, uint8_t *out)
{
    if (!c || (n && (!seq || !qual || !out))) return NTK_ERR_BAD_ARG;
    if (n == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_scratch(c, 0, n))) return rc;
    if ((rc = ensure_scratch(c, 1, n))) return rc;
    if ((rc = ensure_scratch(c, 2, n))) return rc;
    uint8_t *d_s = (uint8_t *)c->scratch[0].p, *d_q = (uint8_t *)c->scratch[1].p, *d_o = (uint8_t *)c->scratch[2].p;
    HIPCHK(hipMemcpyAsync(d_s, seq, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_q, qual, n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(quality_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const uint8_t *)d_s, (const uint8_t *)d_q, n, score, d_o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_o, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTK_OK;
}

/* ---- device utilities ---------------------------------------------------------------------------- */

int ntk_synth_reads_device(ntk_ctx *c, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                           uint32_t n_per_1024, uint8_t *d_out)
{
    if (!c || !d_out || read_len == 0) return NTK_ERR_BAD_ARG;
    if (n_reads == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    const uint64_t total = n_reads * ((uint64_t)read_len + 1);
    const uint64_t threads = (total + 15) / 16;
    hipLaunchKernelGGL(synth_reads_kernel, dim3(grid_for(threads, 256)), dim3(256), 0, c->stream,
                       seed, first_read, n_reads, read_len, n_per_1024, d_out);
    HIPCHK(hipGetLastError());
    return NTK_OK;
}

int ntk_reverse_complement_records_device(ntk_ctx *c, const uint8_t *d_in, uint8_t *d_out, uint64_t n_records,
                                          uint32_t record_len, uint32_t stride)
{
    if (!c || !d_in || !d_out || stride == 0 || record_len > stride) return NTK_ERR_BAD_ARG;
    if (n_records == 0) return NTK_OK;
    HIPCHK(hipSetDevice(c->device));
    const uint64_t total = n_records * stride;
    hipLaunchKernelGGL(revcomp_records_kernel, dim3(grid_for(total, 256)), dim3(256), 0, c->stream,
                       d_in, d_out, n_records, record_len, stride, (const uint16_t *)(c->d_lut + 768));
    HIPCHK(hipGetLastError());
    return NTK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// multi-GPU: one RCCL all-reduce of the accumulators
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    bool ok = false;
};
const RcclApi &load_rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already mapped into the process first (a torch process brings its own librccl.so), then the ROCm one
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!h) for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
#define NTK_RCCL_SYM(f) api.f = (decltype(api.f))dlsym(h, "nccl" #f); if (!api.f) return;
        NTK_RCCL_SYM(GetUniqueId) NTK_RCCL_SYM(CommInitRank) NTK_RCCL_SYM(CommInitAll) NTK_RCCL_SYM(CommDestroy)
        NTK_RCCL_SYM(AllReduce) NTK_RCCL_SYM(GroupStart) NTK_RCCL_SYM(GroupEnd)
#undef NTK_RCCL_SYM
        api.ok = true;
    });
    return api;
}
#define RCCLCHK(x)                                                  \
    do {                                                            \
        ncclResult_t r__ = (x);                                     \
        if (r__ != ncclSuccess) { g_last_rccl = (int)r__; return NTK_ERR_RCCL; } \
    } while (0)

// after the all-reduce the xor word holds a SUM of xors: rebuild it from the 64 one-bit counters (parity of each sum)
__global__ void xor_from_bit_counters_kernel(uint64_t *acc)
{
    const uint64_t bit = acc[NTK_ACC_XOR_BITS + threadIdx.x] & 1ull;
    const uint64_t word = __builtin_amdgcn_ballot_w64(bit != 0);
    if (threadIdx.x == 0) acc[NTK_ACC_XOR] = word;
}
}  // namespace

extern "C" {

struct ntk_comm {
    std::vector<ntk_ctx *> ctxs;     // local contexts (1 for init_rank, n for init_all)
    std::vector<ncclComm_t> comms;   // one communicator handle per local context
    int size = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used;   // around each all-reduce on ctxs[0]'s stream while that ctx has timing on
};

int ntk_comm_unique_id(uint8_t id[NTK_COMM_ID_BYTES])
{
    if (!id) return NTK_ERR_BAD_ARG;
    const RcclApi &R = load_rccl();
    if (!R.ok) { g_last_rccl = -1; return NTK_ERR_RCCL; }
    static_assert(sizeof(ncclUniqueId) == NTK_COMM_ID_BYTES, "NTK_COMM_ID_BYTES follows ncclUniqueId");
    ncclUniqueId u;
    RCCLCHK(R.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return NTK_OK;
}

int ntk_comm_init_rank(ntk_ctx *c, int n_ranks, int rank, const uint8_t id[NTK_COMM_ID_BYTES], ntk_comm **out)
{
    if (!c || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return NTK_ERR_BAD_ARG;
    *out = nullptr;
    const RcclApi &R = load_rccl();
    if (!R.ok) { g_last_rccl = -1; return NTK_ERR_RCCL; }
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t h = nullptr;
    RCCLCHK(R.CommInitRank(&h, n_ranks, u, rank));
    ntk_comm *m = new (std::nothrow) ntk_comm();
    if (!m) { (void)R.CommDestroy(h); return NTK_ERR_NOMEM; }
    m->ctxs.push_back(c); m->comms.push_back(h); m->size = n_ranks;
    *out = m;
    return NTK_OK;
}

int ntk_comm_init_all(ntk_ctx *const *ctxs, int n, ntk_comm **out)
{
    if (!ctxs || !out || n < 1) return NTK_ERR_BAD_ARG;
    *out = nullptr;
    std::vector<int> devs;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return NTK_ERR_BAD_ARG;
        for (int d : devs) if (d == ctxs[i]->device) return NTK_ERR_BAD_ARG;   // one context per device
        devs.push_back(ctxs[i]->device);
    }
    const RcclApi &R = load_rccl();
    if (!R.ok) { g_last_rccl = -1; return NTK_ERR_RCCL; }
    std::vector<ncclComm_t> hs((size_t)n, nullptr);
    RCCLCHK(R.CommInitAll(hs.data(), n, devs.data()));
    ntk_comm *m = new (std::nothrow) ntk_comm();
    if (!m) { for (ncclComm_t h : hs) (void)R.CommDestroy(h); return NTK_ERR_NOMEM; }
    m->ctxs.assign(ctxs, ctxs + n); m->comms = hs; m->size = n;
    *out = m;
    return NTK_OK;
}

int ntk_comm_size(const ntk_comm *m) { return m ? m->size : 0; }

int ntk_allreduce_accumulators(ntk_comm *m)
{
    if (!m) return NTK_ERR_BAD_ARG;
    const RcclApi &R = load_rccl();
    if (!R.ok) { g_last_rccl = -1; return NTK_ERR_RCCL; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ntk_ctx *c0 = m->ctxs.empty() ? nullptr : m->ctxs[0];
    if (c0 && c0->timing) {   // the collective's own duration on this rank (ntk_comm_allreduce_time_ms): events on the stream it runs on
        HIPCHK(hipSetDevice(c0->device));
        int rc = get_event(c0, &e0); if (rc) return rc;
        rc = get_event(c0, &e1); if (rc) { c0->ev_free.push_back(e0); return rc; }
        HIPCHK(hipEventRecord(e0, c0->stream));
    }
    // from here on a failure hands the event pair back to the ctx's pool (the timing of this call is lost, the events are not)
    auto fail = [&](int status) { if (e0) { c0->ev_free.push_back(e0); c0->ev_free.push_back(e1); } return status; };
    ncclResult_t r = R.GroupStart();
    if (r != ncclSuccess) { g_last_rccl = (int)r; return fail(NTK_ERR_RCCL); }
    for (size_t i = 0; i < m->ctxs.size(); i++) {
        ntk_ctx *c = m->ctxs[i];
        r = R.AllReduce(c->d_acc, c->d_acc, NTK_ACC_WORDS, ncclUint64, ncclSum, m->comms[i], c->stream);
        if (r != ncclSuccess) { (void)R.GroupEnd(); g_last_rccl = (int)r; return fail(NTK_ERR_RCCL); }
    }
    if ((r = R.GroupEnd()) != ncclSuccess) { g_last_rccl = (int)r; return fail(NTK_ERR_RCCL); }
    for (ntk_ctx *c : m->ctxs) {
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) { hipLaunchKernelGGL(xor_from_bit_counters_kernel, dim3(1), dim3(64), 0, c->stream, c->d_acc); e = hipGetLastError(); }
        if (e != hipSuccess) { g_last_hip = (int)e; return fail(NTK_ERR_HIP); }
    }
    if (e0) {
        hipError_t e = hipSetDevice(c0->device);
        if (e == hipSuccess) e = hipEventRecord(e1, c0->stream);
        if (e != hipSuccess) { g_last_hip = (int)e; (void)hipGetLastError(); return fail(NTK_ERR_HIP); }
        m->ev_used.emplace_back(e0, e1);
    }
    return NTK_OK;
}

/* Sum of the durations of the all-reduces (collective + xor rebuild, on the first local ctx's stream) issued while that ctx had
 * ntk_ctx_enable_timing on, since the last call; synchronises that stream. */
int ntk_comm_allreduce_time_ms(ntk_comm *m, double *total_ms, uint64_t *calls)
{
    if (!m || !total_ms || m->ctxs.empty()) return NTK_ERR_BAD_ARG;
    ntk_ctx *c = m->ctxs[0];
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double t = 0;
    for (auto &p : m->ev_used) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
        t += ms;
        c->ev_free.push_back(p.first); c->ev_free.push_back(p.second);
    }
    if (calls) *calls = m->ev_used.size();
    m->ev_used.clear();
    *total_ms = t;
    return NTK_OK;
}

void ntk_comm_destroy(ntk_comm *m)
{
    if (!m) return;
    const RcclApi &R = load_rccl();
    if (R.ok) for (ncclComm_t h : m->comms) if (h) (void)R.CommDestroy(h);
    // timing events of all-reduces nobody asked ntk_comm_allreduce_time_ms about: destroyed here (the ctx they were drawn from may be gone)
    for (auto &p : m->ev_used) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete m;
}

}  // extern "C"

