// ntk_pgzip.cpp — parallel inflate of an ordinary gzip stream: see ntk_pgzip.hpp for the scheme.
// Deflate as RFC 1951 writes it, gzip framing as RFC 1952; the accept / reject rules for Huffman code sets follow zlib's inflate
// (over-subscribed sets are errors; an incomplete set only as a single code of length 1; a block without distance codes is fine as
// long as it holds no match), so that every stream zlib inflates this inflates to the same bytes, and corrupt streams are errors.
#include "ntk_pgzip.hpp"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>   // crc32_z / crc32_combine only

#include <sys/mman.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace ntk {
namespace {

constexpr int kLitBits = 11, kDistBits = 9;        // primary table index bits
constexpr uint32_t kWin = 32768;                   // deflate's window
constexpr uint32_t kInvalid = 0x30;                // kind = sub-table link, 0 index bits: no such code
constexpr uint64_t kNone = ~0ull;
constexpr uint64_t kPending = ~0ull - 1;           // a chunk whose block-boundary search has not finished yet
constexpr uint64_t kSpecCap = (uint64_t)96 << 20;  // output symbols a chunk may produce while it is not at the head of the chain (ordinary text: <= 4 MiB x ~6)
enum { kLit = 0, kLen = 1, kEob = 2, kSub = 3 };
// table entry: bits 3:0 code bits to consume (link: index bits of the sub-table), 5:4 kind, 9:6 extra bits, 31:10 value (literal,
// base length, base distance, or the sub-table's offset)
inline uint32_t entry(uint32_t bits, uint32_t kind, uint32_t extra, uint32_t value) { return bits | (kind << 4) | (extra << 6) | (value << 10); }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
    uint32_t lit[(1 << kLitBits) + 288 * 16];
    uint32_t dist[(1 << kDistBits) + 32 * 64];
};

inline uint32_t bit_reverse(uint32_t c, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (c & 1); c >>= 1; }
    return r;
}

// Canonical Huffman decode table for lens[0, n).  is_dist: distance alphabet (30 symbols with a meaning), else literal / length.
// false = not a code set zlib accepts (over-subscribed, or incomplete other than one code of length 1 / no code at all for distances).
bool build_table(const uint8_t *lens, int n, int P, uint32_t *table, bool is_dist)
{
    int count[16] = {0};
    for (int s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    int left = 1, total = 0, maxlen = 0;
    for (int l = 1; l <= 15; l++) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
        total += count[l];
        if (count[l]) maxlen = l;
    }
    if (left > 0 && !(is_dist && total == 0) && !(total == 1 && maxlen == 1)) return false;
    uint32_t next[16];
    next[1] = 0;
    for (int l = 1; l < 15; l++) next[l + 1] = (next[l] + (uint32_t)count[l]) << 1;
    const uint32_t psize = 1u << P, pmask = psize - 1;
    for (uint32_t i = 0; i < psize; i++) table[i] = kInvalid;
    uint8_t sub_bits[1 << kLitBits];
    bool any_long = maxlen > P;
    if (any_long) memset(sub_bits, 0, psize);
    uint32_t code_of[288];
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = bit_reverse(next[l]++, l);
        code_of[s] = rev;
        if (l > P) { const uint32_t pre = rev & pmask; if (l - P > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - P); }
    }
    uint32_t sub_next = psize;
    if (any_long)
        for (uint32_t pre = 0; pre < psize; pre++)
            if (sub_bits[pre]) {
                table[pre] = entry(sub_bits[pre], kSub, 0, sub_next);
                for (uint32_t i = 0; i < (1u << sub_bits[pre]); i++) table[sub_next + i] = kInvalid;
                sub_next += 1u << sub_bits[pre];
            }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t e;
        if (is_dist) e = s < 30 ? entry(0, kLen, kDistExtra[s], kDistBase[s]) : kInvalid;
        else if (s < 256) e = entry(0, kLit, 0, (uint32_t)s);
        else if (s == 256) e = entry(0, kEob, 0, 0);
        else e = s < 286 ? entry(0, kLen, kLenExtra[s - 257], kLenBase[s - 257]) : kInvalid;
        const uint32_t rev = code_of[s];
        if (l <= P) {
            if (e != kInvalid) e |= (uint32_t)l;
            for (uint32_t i = rev; i < psize; i += 1u << l) table[i] = e;
        } else {
            const uint32_t pre = rev & pmask, sb = sub_bits[pre], off = table[pre] >> 10;
            if (e != kInvalid) e |= (uint32_t)(l - P);
            for (uint32_t i = rev >> P; i < (1u << sb); i += 1u << (l - P)) table[off + i] = e;
        }
    }
    return true;
}

const Tables &fixed_tables()
{
    static const Tables *t = [] {
        Tables *x = new Tables();
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        build_table(l, 288, kLitBits, x->lit, false);
        uint8_t d[32];
        for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(d, 32, kDistBits, x->dist, true);
        return x;
    }();
    return *t;
}

// ---- bit reader: LSB-first, 64-bit buffer, byte offsets (reads past the end yield zero bits and are counted) -------------------
struct Bits {
    const uint8_t *in; uint64_t n;   // the whole input
    uint64_t ip = 0;                 // next byte to load (may run past n: phantom zero bytes)
    uint64_t bb = 0; int bc = 0;     // bit buffer, valid bits
    void seek_bit(uint64_t bit) { ip = bit >> 3; bb = 0; bc = 0; refill(); bb >>= (bit & 7); bc -= (int)(bit & 7); }
    inline void refill()
    {
        if (ip + 8 <= n) {
            uint64_t w;
            memcpy(&w, in + ip, 8);
            bb |= w << bc;
            ip += (uint64_t)((63 - bc) >> 3);
            bc |= 56;
        } else {
            while (bc <= 56) { bb |= (uint64_t)(ip < n ? in[ip] : 0) << bc; ip++; bc += 8; }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(bb & ((1ull << k) - 1)); }
    inline void drop(int k) { bb >>= k; bc -= k; }
    inline uint32_t take(int k) { const uint32_t v = peek(k); drop(k); return v; }
    uint64_t bit_pos() const { return ip * 8 - (uint64_t)bc; }
    bool overrun() const { return bit_pos() > n * 8; }
    void align_byte() { drop(bc & 7); }
    void to_bytes(uint64_t *byte_off) { *byte_off = ip - (uint64_t)(bc >> 3); bb = 0; bc = 0; }   // (after align_byte)
};

// ---- memory: page-aligned anonymous mappings, advised to use huge pages (a first touch per 2 MiB instead of per 4 KiB - first-touch
// faults cost more than the decoding on virtualised hosts), grown with mremap, recycled through a pool while a call runs ----------
struct Block { uint8_t *p = nullptr; size_t cap = 0; };
constexpr size_t kHuge = (size_t)2 << 20;
bool block_alloc(Block &b, size_t bytes, bool reserve_only = false)
{
    if (bytes == 0 || bytes > ((size_t)1 << 46)) return false;   // (a limit of 2^64 - 1 must not wrap to a zero-byte mapping that "succeeds")
    bytes = (bytes + kHuge - 1) & ~(kHuge - 1);
    void *m = mmap(nullptr, bytes + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | (reserve_only ? MAP_NORESERVE : 0), -1, 0);
    if (m == MAP_FAILED) return false;
    uint8_t *a = (uint8_t *)(((uintptr_t)m + kHuge - 1) & ~(uintptr_t)(kHuge - 1));   // 2 MiB aligned: trim the slack on both sides
    if (a > (uint8_t *)m) munmap(m, (size_t)(a - (uint8_t *)m));
    const size_t tail = (size_t)((uint8_t *)m + bytes + kHuge - (a + bytes));
    if (tail) munmap(a + bytes, tail);
    (void)madvise(a, bytes, MADV_HUGEPAGE);
    b.p = a; b.cap = bytes;
    return true;
}
bool block_grow(Block &b, size_t bytes)
{
    bytes = (bytes + kHuge - 1) & ~(kHuge - 1);
    if (bytes <= b.cap) return true;
    if (!b.p) return block_alloc(b, bytes);
    void *m = mremap(b.p, b.cap, bytes, MREMAP_MAYMOVE);
    if (m == MAP_FAILED) return false;
    b.p = (uint8_t *)m; b.cap = bytes;
    (void)madvise(b.p, bytes, MADV_HUGEPAGE);
    return true;
}
void block_free(Block &b) { if (b.p) munmap(b.p, b.cap); b.p = nullptr; b.cap = 0; }

// Unmapping what a run touched is not free - the symbol buffers of the chunks (~1 GB resident at 16 threads) and the page tables of the text's
// range took 60 - 70 ms of a 370 ms configs[4] call when they were unmapped before the call returned (profiles/r06d).  The results are complete
// by then; the mappings go on a detached thread.
void release_async(std::vector<Block> &&blocks)
{
    if (blocks.empty()) return;
    try {
        // MADV_DONTNEED first: giving the pages back is the expensive part and only needs the address space's lock shared; the unmap that
        // follows holds it exclusively (every page fault of a run that has started meanwhile waits for it) but finds nothing left to free
        std::thread([b = std::move(blocks)]() mutable {
            for (Block &x : b) { if (x.p) (void)madvise(x.p, x.cap, MADV_DONTNEED); block_free(x); }
        }).detach();
    } catch (...) {
        for (Block &x : blocks) block_free(x);
    }
}

struct Pool {
    std::mutex mu;
    std::vector<Block> free_;
    Block take(size_t bytes)   // the largest free block (grown if needed), else a new one; p == nullptr: out of memory
    {
        Block b;
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++) if (best == free_.size() || free_[i].cap > free_[best].cap) best = i;
            if (best < free_.size()) { b = free_[best]; free_[best] = free_.back(); free_.pop_back(); }
        }
        if (b.p) { if (!block_grow(b, bytes)) { block_free(b); } return b; }
        block_alloc(b, bytes);
        return b;
    }
    void give(Block &b) { if (!b.p) return; std::lock_guard<std::mutex> g(mu); free_.push_back(b); b = Block(); }
    ~Pool() { release_async(std::move(free_)); }
};

// growable output of symbols of type T with a 32 KiB prefix in front (the window: markers, or the bytes carried over)
template <class T>
struct Out {
    Block blk; Pool *pool = nullptr;
    T *base = nullptr; size_t pos = kWin, cap = 0;
    bool reserve(size_t more)
    {
        if (pos + more <= cap) return true;
        size_t want = cap ? cap * 2 : ((size_t)4 << 20);
        while (pos + more > want) want *= 2;
        if (!blk.p) { blk = pool->take(want * sizeof(T)); if (!blk.p) return false; }
        else if (!block_grow(blk, want * sizeof(T))) return false;
        base = (T *)blk.p; cap = blk.cap / sizeof(T);
        return true;
    }
    size_t produced() const { return pos - kWin; }
    void drop() { if (blk.p) pool->give(blk); base = nullptr; cap = 0; }
};

enum { kOk = 0, kCorrupt = 1, kTooLarge = 2, kNoMem = 3, kCancelled = 4 };

template <class T>
inline void copy_match(T *dst, uint32_t dist, uint32_t len)
{
    const T *src = dst - dist;
    constexpr uint32_t kStep = 8 / sizeof(T);   // symbols per 8-byte word
    T *const end = dst + len;
    if (dist < kStep) {
        // a run of a short pattern (dist = 1: a homopolymer / a quality plateau): the first two words symbol by symbol, then word copies at
        // the largest multiple of the period that is at least a word back
        const uint32_t head = len < 2 * kStep ? len : 2 * kStep;
        for (uint32_t i = 0; i < head; i++) dst[i] = src[i];
        if (len <= head) return;
        dst += head;
        src = dst - dist * ((2 * kStep) / dist);
    }
    do { memcpy(dst, src, 8); dst += kStep; src += kStep; } while (dst < end);   // (may write up to 7 bytes past: slack is reserved)
}

// One Huffman-coded block body (after its header) up to and including the end-of-block code.  floor = the first position of o.base
// a match may read from (byte mode: history that does not exist is an error; marker mode: 0 - the prefix holds the markers).
// The reader's and the output's state live in locals while the loop runs (a byte store may alias anything: left in the structs they
// would be re-loaded after every literal).
template <class T>
int inflate_codes(Bits &bits, const Tables &tb, Out<T> &o, size_t floor, uint64_t max_out)
{
    constexpr uint32_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;
    const uint8_t *const in = bits.in;
    const uint64_t n = bits.n;
    uint64_t ip = bits.ip, bb = bits.bb;
    int bc = bits.bc;
    T *base = o.base;
    size_t pos = o.pos, cap = o.cap;
    const uint32_t *const lit = tb.lit, *const dtab = tb.dist;
    int rc;
#define PGZ_REFILL()                                                                              \
    do {                                                                                          \
        if (ip + 8 <= n) { uint64_t w_; memcpy(&w_, in + ip, 8); bb |= w_ << bc; ip += (uint64_t)((63 - bc) >> 3); bc |= 56; } \
        else while (bc <= 56) { bb |= (uint64_t)(ip < n ? in[ip] : 0) << bc; ip++; bc += 8; }     \
    } while (0)
#define PGZ_DROP(k) do { const int k_ = (int)(k); bb >>= k_; bc -= k_; } while (0)
    for (;;) {
        if (pos + 320 > cap) {
            o.pos = pos;
            if (o.produced() > max_out) { rc = kTooLarge; break; }
            if (!o.reserve(1 << 16)) { rc = kNoMem; break; }
            base = o.base; cap = o.cap;
        }
        PGZ_REFILL();
        // past the end of the input the reader hands out zero bits: a truncated stream whose all-zero code is a literal would otherwise emit
        // literals until max_out (checked on every trip: the literal path below never reaches the match path's exit)
        if (__builtin_expect(ip > n + 16, 0)) { rc = kCorrupt; break; }
        uint32_t e = lit[bb & lmask];
        if ((e & 0x30) == 0x30) {
            const uint32_t sb = e & 15;
            if (!sb) { rc = kCorrupt; break; }
            e = lit[(e >> 10) + ((bb >> kLitBits) & ((1u << sb) - 1))];
            if ((e & 0x30) == 0x30) { rc = kCorrupt; break; }
            PGZ_DROP(kLitBits);
        }
        PGZ_DROP(e & 15);
        const uint32_t kind = (e >> 4) & 3;
        if (kind == kLit) {
            // up to two more literals out of the bits already in the buffer (<= 15 consumed, >= 41 left; a literal of the primary table
            // takes <= 11): the common case in the sequence lines
            base[pos++] = (T)(e >> 10);
            uint32_t e2 = lit[bb & lmask];
            if ((e2 & 0x30) == 0) {
                PGZ_DROP(e2 & 15);
                base[pos++] = (T)(e2 >> 10);
                e2 = lit[bb & lmask];
                if ((e2 & 0x30) == 0) { PGZ_DROP(e2 & 15); base[pos++] = (T)(e2 >> 10); }
            }
            continue;
        }
        if (kind == kEob) { rc = kOk; break; }
        uint32_t xb = (e >> 6) & 15;
        const uint32_t len = (e >> 10) + (uint32_t)(bb & ((1ull << xb) - 1));
        PGZ_DROP(xb);
        uint32_t d = dtab[bb & dmask];
        if ((d & 0x30) == 0x30) {
            const uint32_t sb = d & 15;
            if (!sb) { rc = kCorrupt; break; }
            d = dtab[(d >> 10) + ((bb >> kDistBits) & ((1u << sb) - 1))];
            if ((d & 0x30) == 0x30) { rc = kCorrupt; break; }
            PGZ_DROP(kDistBits);
        }
        PGZ_DROP(d & 15);
        xb = (d >> 6) & 15;
        const uint32_t dist = (d >> 10) + (uint32_t)(bb & ((1ull << xb) - 1));
        PGZ_DROP(xb);
        if ((size_t)dist > pos - floor) { rc = kCorrupt; break; }
        copy_match(base + pos, dist, len);
        pos += len;
    }
#undef PGZ_DROP
#undef PGZ_REFILL
    bits.ip = ip; bits.bb = bb; bits.bc = bc;
    o.pos = pos;
    if (rc == kOk && bits.overrun()) rc = kCorrupt;
    return rc;
}

// A dynamic block's code description -> tables.  The reader stands behind the 3 header bits.
int read_dynamic(Bits &b, Tables &tb)
{
    b.refill();
    const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return kCorrupt;
    uint8_t cl[19] = {0};
    for (uint32_t i = 0; i < hclen; i++) { if (b.bc < 3) b.refill(); cl[kClOrder[i]] = (uint8_t)b.take(3); }
    uint32_t pre[1 << 7];
    {   // the code-length code: must be complete (zlib: an incomplete set is never accepted here)
        int left = 1;
        for (int l = 1; l <= 7; l++) { int c = 0; for (int s = 0; s < 19; s++) c += cl[s] == l; left = (left << 1) - c; if (left < 0) return kCorrupt; }
        if (left != 0) return kCorrupt;
        if (!build_table(cl, 19, 7, pre, false)) return kCorrupt;   // (entries: symbols 0..18 come out as "literals")
    }
    uint8_t lens[288 + 32];
    uint32_t i = 0;
    while (i < hlit + hdist) {
        b.refill();
        const uint32_t e = pre[b.bb & 127];
        if ((e & 0x30) == 0x30) return kCorrupt;
        b.drop((int)(e & 15));
        const uint32_t s = e >> 10;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        uint32_t rep, val = 0;
        if (s == 16) { if (i == 0) return kCorrupt; val = lens[i - 1]; rep = 3 + b.take(2); }
        else if (s == 17) rep = 3 + b.take(3);
        else rep = 11 + b.take(7);
        if (i + rep > hlit + hdist) return kCorrupt;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return kCorrupt;   // no end-of-block code
    if (!build_table(lens, (int)hlit, kLitBits, tb.lit, false)) return kCorrupt;
    if (!build_table(lens + hlit, (int)hdist, kDistBits, tb.dist, true)) return kCorrupt;
    return b.overrun() ? kCorrupt : kOk;
}

template <class T>
int inflate_stored(Bits &b, Out<T> &o, uint64_t max_out)
{
    b.align_byte();
    b.refill();
    const uint32_t len = b.take(16), nlen = b.take(16);
    if ((len ^ nlen) != 0xFFFFu) return kCorrupt;
    uint64_t off;
    b.to_bytes(&off);
    if (off + len > b.n) return kCorrupt;
    if (o.produced() + len > max_out) return kTooLarge;
    if (!o.reserve((size_t)len + 320)) return kNoMem;
    for (uint32_t i = 0; i < len; i++) o.base[o.pos + i] = (T)b.in[off + i];
    o.pos += len;
    b.ip = off + len;
    return kOk;
}

// gzip member header at byte offset *off (RFC 1952): on success *off = first byte of the deflate data
bool skip_gzip_header(const uint8_t *in, uint64_t n, uint64_t *off)
{
    uint64_t p = *off;
    if (n - p < 10 || in[p] != 0x1F || in[p + 1] != 0x8B || in[p + 2] != 8) return false;
    const uint8_t flg = in[p + 3];
    if (flg & 0xE0) return false;   // reserved bits
    p += 10;
    if (flg & 4) { if (n - p < 2) return false; const uint64_t xlen = in[p] | ((uint64_t)in[p + 1] << 8); p += 2; if (n - p < xlen) return false; p += xlen; }
    for (int f = 8; f <= 16; f <<= 1)
        if (flg & f) { const void *z = memchr(in + p, 0, (size_t)(n - p)); if (!z) return false; p = (uint64_t)((const uint8_t *)z - in) + 1; }
    if (flg & 2) { if (n - p < 2) return false; p += 2; }
    *off = p;
    return true;
}

struct MemberEnd { uint64_t out_off; uint32_t crc, isize; };   // out_off: chunk-relative output offset where the member ends

struct Chunk {
    uint64_t byte_begin = 0;            // the compressed range this chunk was cut for starts here
    uint64_t start_bit = kNone;         // block boundary the chunk starts at (chunk 0: the first member's deflate data)
    bool trusted_start = false;         // chunk 0: the start is the stream's, byte mode from the first symbol
    // results
    int status = kOk;
    bool done = false, dropped = false;
    uint32_t searching = 0;             // a worker is looking for the chunk's block boundary (read by the decoders of earlier chunks)
    bool deferred = false, in_flight = false;   // deferred: it outgrew the cap of a chunk that is not at the chain's head and waits to be decoded there
    bool at_end = false;                // the chunk decoded to the end of the file
    uint32_t end_chunk = 0;             // else: index of the chunk whose start_bit it stopped at
    Out<uint16_t> o16;                  // marker-mode symbols (o16.base[0, kWin) = the marker prefix)
    Out<uint8_t> o8;                    // byte-mode output (o8.base[0, kWin) = its window prefix)
    size_t nsym = 0, nbyt = 0;
    std::vector<MemberEnd> members;
    std::vector<uint64_t> member_starts;         // chunk-relative output offsets where a NEW member's output begins
    uint32_t min_marker = kWin;                  // smallest window index a surviving marker names
    double busy_s = 0;
    void release() { o16.drop(); o8.drop(); }
};

// position of the last symbol >= 256 in s[from, to), or `none`
size_t last_marker(const uint16_t *s, size_t from, size_t to, size_t none)
{
    size_t i = to;
    while (i > from + 32) {   // 32 symbols at a time from the back
        uint16_t any = 0;
        for (size_t j = i - 32; j < i; j++) any |= s[j];
        if (any >= 256) break;
        i -= 32;
    }
    while (i > from) { if (s[i - 1] >= 256) return i - 1; i--; }
    return none;
}

// Decodes from c.start_bit until the block boundary where a later chunk starts (or the end of the file).
// speculative_blocks > 0: validation run of the boundary search - decode that many blocks (or to the final block) and report only
// the status; nothing is kept.
void decode_chunk(const uint8_t *in, uint64_t n, std::vector<Chunk> &chunks, uint32_t own, uint64_t max_out, int speculative_blocks, Pool *pool)
{
    const auto t0 = std::chrono::steady_clock::now();
    Chunk &c = chunks[own];
    Bits b{in, n};
    b.seek_bit(c.start_bit);
    Out<uint16_t> &o16 = c.o16;
    Out<uint8_t> &o8 = c.o8;
    o16.pool = pool; o8.pool = pool;
    o16.pos = kWin; o8.pos = kWin;
    c.members.clear(); c.member_starts.clear(); c.at_end = false; c.end_chunk = 0; c.min_marker = kWin; c.nsym = c.nbyt = 0;   // (a deferred chunk is decoded twice)
    bool byte_mode = c.trusted_start;
    size_t floor8 = kWin;                 // byte mode: first position a match may read (kWin = no history)
    size_t scanned = kWin, lastm = 0;     // marker mode: symbols scanned for markers so far, last marker seen (0 = none: positions start at kWin)
    int status = kOk;
    Tables *tb = new (std::nothrow) Tables();
    uint32_t next_chunk = own + 1;
    int blocks = 0;
    auto out_off = [&]() -> uint64_t { return (uint64_t)(o16.base ? o16.produced() : 0) + (uint64_t)(o8.base ? o8.produced() : 0); };
    if (!tb) status = kNoMem;
    if (status == kOk && byte_mode && !o8.reserve(1 << 16)) status = kNoMem;
    if (status == kOk && !byte_mode) {
        if (!o16.reserve(1 << 16)) status = kNoMem;
        else for (uint32_t i = 0; i < kWin; i++) o16.base[i] = (uint16_t)(kWin + i);
    }
    while (status == kOk) {
        // block boundary: is this where a later chunk starts?
        if (!speculative_blocks && blocks > 0) {
            // (the later chunks' starts are found while this chunk decodes: a start that is not known yet cannot stop it - if this decoder passes
            // the place before the search reports it, it simply runs on to the next known start and that chunk's work is dropped)
            const uint64_t here = b.bit_pos();
            uint32_t stop = 0;
            for (uint32_t j = next_chunk; j < chunks.size(); j++) {
                uint64_t sb = __atomic_load_n(&chunks[j].start_bit, __ATOMIC_ACQUIRE);
                if (sb == kPending) {
                    if (chunks[j].byte_begin * 8 > here) break;
                    // the boundary lies in the range chunk j is being searched in right now: the search's answer (milliseconds away, and it
                    // always ends) decides whether this decoder stops here - running on would decode chunk j's text a second time
                    if (!__atomic_load_n(&chunks[j].searching, __ATOMIC_ACQUIRE)) continue;   // nobody has taken chunk j yet: it cannot stop us
                    while ((sb = __atomic_load_n(&chunks[j].start_bit, __ATOMIC_ACQUIRE)) == kPending) std::this_thread::yield();
                }
                if (sb == kNone || sb < here) { if (j == next_chunk) next_chunk++; continue; }
                if (sb == here) stop = j;
                break;
            }
            if (stop) { c.end_chunk = stop; break; }
        }
        if (speculative_blocks && blocks >= speculative_blocks) break;
        b.refill();
        const uint32_t bfinal = b.take(1), btype = b.take(2);
        blocks++;
        if (btype == 3) { status = kCorrupt; break; }
        if (btype == 0) status = byte_mode ? inflate_stored(b, o8, max_out) : inflate_stored(b, o16, max_out);
        else {
            const Tables *use = &fixed_tables();
            if (btype == 2) { status = read_dynamic(b, *tb); use = tb; }
            if (status == kOk) status = byte_mode ? inflate_codes(b, *use, o8, floor8, max_out) : inflate_codes(b, *use, o16, 0, max_out);
        }
        if (status != kOk) break;
        if (b.overrun()) { status = kCorrupt; break; }
        if (out_off() > max_out) { status = kTooLarge; break; }
        if (bfinal) {
            if (speculative_blocks) break;
            // member trailer, then the next member or the end of the file
            b.align_byte();
            uint64_t off;
            b.to_bytes(&off);
            if (off > n || n - off < 8) { status = kCorrupt; break; }
            MemberEnd me;
            me.out_off = out_off();
            me.crc = (uint32_t)in[off] | ((uint32_t)in[off + 1] << 8) | ((uint32_t)in[off + 2] << 16) | ((uint32_t)in[off + 3] << 24);
            me.isize = (uint32_t)in[off + 4] | ((uint32_t)in[off + 5] << 8) | ((uint32_t)in[off + 6] << 16) | ((uint32_t)in[off + 7] << 24);
            c.members.push_back(me);
            off += 8;
            bool pad = true;   // trailing zero padding after the last member is tolerated, as zlib-based readers do
            for (uint64_t i = off; i < n && pad; i++) pad = in[i] == 0;
            if (off >= n || pad) { c.at_end = true; break; }
            if (!skip_gzip_header(in, n, &off)) { status = kCorrupt; break; }
            c.member_starts.push_back(me.out_off);
            // a new member starts with an empty window: plain bytes from here on
            if (!byte_mode) { byte_mode = true; if (!o8.reserve(1 << 16)) { status = kNoMem; break; } }
            floor8 = o8.pos;
            b.seek_bit(off * 8);
            blocks = 0;   // (the member's first block boundary is the header's end, not a place another chunk could have found)
            continue;
        }
        if (!byte_mode && !speculative_blocks && o16.produced() >= kWin) {
            // no marker left in the last 32 KiB?  then the rest of the chunk is a plain byte decoder's work
            const size_t lm = last_marker(o16.base, scanned, o16.pos, 0);
            if (lm) lastm = lm;
            scanned = o16.pos;
            if (lastm + kWin <= o16.pos) {
                if (!o8.reserve(1 << 16)) { status = kNoMem; break; }
                for (uint32_t i = 0; i < kWin; i++) o8.base[i] = (uint8_t)o16.base[o16.pos - kWin + i];
                byte_mode = true;
                floor8 = 0;
            }
        }
    }
    if (status == kOk && !speculative_blocks && !c.at_end && c.end_chunk == 0) status = kCorrupt;   // (left the loop without an end: cannot happen)
    c.status = status;
    if (!speculative_blocks && status == kOk) {
        c.nsym = o16.base ? o16.produced() : 0;
        c.nbyt = o8.base ? o8.produced() : 0;
        if (c.nsym) {   // smallest window index still named by a marker (for the member-start check)
            uint32_t mn = 2 * kWin;
            const uint16_t *s = o16.base + kWin;
            for (size_t i = 0; i < c.nsym; i++) { const uint32_t v = s[i]; mn = (v >= 256 && v < mn) ? v : mn; }
            c.min_marker = mn >= 2 * kWin ? kWin : mn - kWin;
        }
    } else {
        c.release();
    }
    delete tb;
    c.busy_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// First deflate block boundary at a bit offset in [from_byte * 8, to_byte * 8): a non-final dynamic block whose header describes
// complete codes, whose body decodes, and which is followed by two more blocks that decode.  kNone: none found.
uint64_t find_block(const uint8_t *in, uint64_t n, uint64_t from_byte, uint64_t to_byte, Pool *pool)
{
    if (to_byte + 16 > n) to_byte = n > 16 ? n - 16 : 0;
    std::vector<Chunk> probe(1);
    Tables *tb = new (std::nothrow) Tables();
    if (!tb) return kNone;
    uint64_t found = kNone;
    for (uint64_t byte = from_byte; byte < to_byte && found == kNone; byte++) {
        uint64_t w0, w1;
        memcpy(&w0, in + byte, 8);
        memcpy(&w1, in + byte + 8, 8);
        for (int sh = 0; sh < 8; sh++) {
            const uint64_t v = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
            if ((v & 7) != 4) continue;                                    // BFINAL = 0, BTYPE = 2
            if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;    // HLIT, HDIST
            const uint32_t hclen = (uint32_t)((v >> 13) & 15) + 4;
            // the code-length code's own lengths (19 x 3 bits from bit 17 on): a complete set?
            uint64_t x = (v >> 17) | (w1 >> sh << 47);
            uint32_t kraft = 0;
            for (uint32_t i = 0; i < hclen; i++) { const uint32_t l = (uint32_t)(x & 7); x >>= 3; if (l) kraft += 128u >> l; }
            if (kraft != 128) continue;
            const uint64_t bit = byte * 8 + (uint64_t)sh;
            {   // the whole header before anything is decoded: code lengths that parse, complete literal / length and distance codes
                Bits b{in, n};
                b.seek_bit(bit + 3);
                if (read_dynamic(b, *tb) != kOk) continue;
            }
            probe[0] = Chunk();
            probe[0].start_bit = bit;
            decode_chunk(in, n, probe, 0, (uint64_t)64 << 20, 3, pool);
            const bool ok = probe[0].status == kOk;
            probe[0].release();
            if (ok) { found = bit; break; }
        }
    }
    delete tb;
    return found;
}

struct Crc {   // (one per process: crc_lib() below - the handle is never closed, so it is taken once)
    uint32_t (*fast)(uint32_t, const void *, size_t) = nullptr;
    Crc()
    {
        if (void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL)) fast = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
    }
    uint32_t run(const uint8_t *p, size_t n) const
    {
        if (fast) return fast(0, p, n);
        uint32_t c = 0;
        while (n) { const size_t k = n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n; c = (uint32_t)crc32_z(c, p, k); p += k; n -= k; }
        return c;
    }
};

const Crc &crc_lib() { static const Crc c; return c; }

template <class F>
void run_parallel(uint32_t threads, F &&f)
{
    std::vector<std::thread> th;
    try { for (uint32_t t = 1; t < threads; t++) th.emplace_back(f); } catch (...) {}   // fewer threads than asked for: the work is pulled, nothing is lost
    f();
    for (auto &t : th) t.join();
}

// a chunk's output on its way into the final buffer
struct ResolveTask {
    uint32_t chunk; uint64_t out_off;
    uint8_t *win;   // the 32 KiB before the chunk (owned: freed by the task)
};
struct CrcPiece { uint64_t off, len; uint32_t crc; };

}  // namespace

uint8_t *pgz_alloc(uint64_t n)
{
    const size_t bytes = (size_t)(((n ? n : 1) + 4095) & ~(uint64_t)4095);
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return nullptr;
    (void)madvise(m, bytes, MADV_HUGEPAGE);
    return (uint8_t *)m;
}

void pgz_free(uint8_t *p, uint64_t n)
{
    if (p) munmap(p, (size_t)(((n ? n : 1) + 4095) & ~(uint64_t)4095));
}

namespace {

std::mutex g_text_mu;
Block g_text;                 // the text range of the last streamed run, kept for the next one (pgz_stream_release)

// The decoder behind pgz_inflate (stream == nullptr: the whole output in one buffer, handed over at the end) and pgz_inflate_stream (the
// output becomes readable as the chain advances; see ntk_pgzip.hpp).
int inflate_impl(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, uint8_t **out, uint64_t *out_n, PgzStats *stats, PgzStream *stream)
{
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    if (out) { *out = nullptr; *out_n = 0; }
    PgzStats st;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    st.threads = n_threads;
    if (limit > ((uint64_t)1 << 46)) limit = (uint64_t)1 << 46;   // ("no limit" spelled as 2^64 - 1 must not wrap the mapping's size)
    uint64_t off = 0;
    if (!skip_gzip_header(in, n, &off)) return kCorrupt;
    Pool pool;
    // chunks of the compressed bytes: about sixteen per thread, 512 KiB .. 4 MiB each (thirty-two per thread, <= 2 MiB: 25 % less resident
    // memory - a chunk in flight holds its text as 16-bit symbols - and 4 % slower, profiles/r06c/config5_phases3.txt)
    uint64_t chunk_bytes = n / ((uint64_t)n_threads * 16) + 1;
    if (chunk_bytes < (512u << 10)) chunk_bytes = 512u << 10;
    if (chunk_bytes > (4u << 20)) chunk_bytes = 4u << 20;
    const uint64_t n_chunks64 = n_threads == 1 ? 1 : (n - off + chunk_bytes - 1) / chunk_bytes;
    if (n_chunks64 > 0x7FFFFFFFull) return kTooLarge;
    const uint32_t n_chunks = (uint32_t)(n_chunks64 < 1 ? 1 : n_chunks64);
    std::vector<Chunk> chunks;
    try { chunks.resize(n_chunks); } catch (...) { return kNoMem; }
    chunks[0].byte_begin = off; chunks[0].start_bit = off * 8; chunks[0].trusted_start = true;
    for (uint32_t i = 1; i < n_chunks; i++) { chunks[i].byte_begin = off + (uint64_t)i * chunk_bytes; chunks[i].start_bit = kPending; }
    // The boundary search (every later chunk looks for a block boundary in the first MiB of its own range; a stream without dynamic blocks -
    // stored data, Z_FIXED - has none to find: its chunks fall to the one before them) is part of the pipeline below: a worker that takes
    // chunk i for decoding searches its start first.  (Round 5 ran the search as a phase of its own before any decoding; with the progressive
    // form that would delay the first ready byte by the search of the WHOLE file.)
    // the output: address space for the whole limit, touched as it is written (no growing, no copying); smaller if the system refuses
    Block fin;
    uint64_t fin_limit = limit;
    if (stream) {   // the range a previous streamed run left behind, if it is large enough (every byte is written before it is read: what
                    // the earlier run left in it - the pages at its pieces' edges - is simply overwritten)
        std::lock_guard<std::mutex> g(g_text_mu);
        if (g_text.p && g_text.cap >= limit) { fin = g_text; g_text = Block(); }
    }
    if (!fin.p) {
        uint64_t want = limit < ((uint64_t)1 << 20) ? ((uint64_t)1 << 20) : limit;
        while (!block_alloc(fin, (size_t)want, true)) {
            if (want <= ((uint64_t)256 << 20)) return kNoMem;
            want /= 2;
        }
        if (fin.cap < fin_limit) fin_limit = fin.cap;
    }
    // One pipeline.  Workers take chunks in index order (at most max_ahead chunks beyond the chain's head are in flight): boundary search, then
    // decode; whenever the chunk at the chain's head is done, its window is fixed (sequentially: <= 32 KiB of work), its place in the output is
    // known and a resolve task for it is queued; resolve tasks go first.  Progressive form: the head is not placed while more than
    // stream->window bytes are placed and not consumed; [0, ready) = the resolved prefix.
    const Crc &crc = crc_lib();
    std::mutex own_mu;
    std::condition_variable own_cv;
    std::mutex &mu = stream ? stream->mu : own_mu;
    std::condition_variable &cv = stream ? stream->cv : own_cv;
    std::deque<ResolveTask> rq;
    std::vector<CrcPiece> pieces;
    std::vector<std::pair<uint64_t, MemberEnd>> ends;   // absolute output offsets of the member ends
    std::deque<std::pair<uint64_t, bool>> placed;       // (end offset, resolved) of the placed chunks beyond `ready`, in stream order
    uint64_t ready = 0;
    uint32_t next_decode = 0, chain_cur = 0, resolving = 0;
    const uint32_t max_ahead = n_threads + n_threads / 2 + 2;   // (2 T + 2 until round 6: the symbol buffers of the chunks in flight are the run's largest resident item)
    uint64_t total = 0, avail = 0;
    std::vector<uint8_t> tail(kWin, 0);   // the kWin bytes before `total` (valid: the last `avail`)
    bool chain_done = false;
    uint64_t dropped_in = 0;   // compressed bytes whose pages have been given back (file-mapped input only)
    int rc = kOk;
    if (stream) { std::lock_guard<std::mutex> g(mu); stream->base = fin.p; stream->map = fin.p; stream->map_bytes = fin.cap; }
    auto advance_chain = [&]() {   // under mu
        while (rc == kOk && !chain_done && chain_cur < n_chunks && chunks[chain_cur].done) {
            Chunk &c = chunks[chain_cur];
            if (c.status != kOk) { rc = c.status; break; }
            if (stream) {
                const uint64_t backlog = total - stream->consumed;
                if (backlog > stream->peak_backlog) stream->peak_backlog = backlog;
                if (backlog > stream->window) break;   // the consumer's next cv.notify_all() brings a worker back here
            }
            const uint64_t tot = c.nsym + c.nbyt;
            if (c.nsym && c.min_marker < kWin - avail) { rc = kCorrupt; break; }   // a match reaches back before the member's first byte
            if (total + tot > fin_limit) { rc = kTooLarge; break; }
            ResolveTask t{chain_cur, total, (uint8_t *)malloc(kWin)};
            if (!t.win) { rc = kNoMem; break; }
            memcpy(t.win, tail.data(), kWin);
            // the next window: the chunk's own last 32 KiB (what lies in its symbols is resolved here)
            if (tot >= kWin && c.nbyt >= kWin) memcpy(tail.data(), c.o8.base + kWin + c.nbyt - kWin, kWin);
            else {
                std::vector<uint8_t> nxt(kWin);
                for (uint32_t i = 0; i < kWin; i++) {   // stream position of nxt[i] relative to the chunk's start: tot - kWin + i
                    const int64_t p = (int64_t)tot - (int64_t)kWin + (int64_t)i;
                    if (p < 0) nxt[i] = t.win[kWin + p];
                    else if ((uint64_t)p < c.nsym) { const uint16_t s = c.o16.base[kWin + p]; nxt[i] = s < 256 ? (uint8_t)s : t.win[s - kWin]; }
                    else nxt[i] = c.o8.base[kWin + ((uint64_t)p - c.nsym)];
                }
                tail.swap(nxt);
            }
            for (const MemberEnd &me : c.members) ends.push_back({total + me.out_off, me});
            uint64_t a = avail + tot;
            if (!c.member_starts.empty()) a = tot - c.member_starts.back();
            avail = a < kWin ? a : kWin;
            total += tot;
            rq.push_back(t);
            placed.push_back({total, false});
            st.decode_busy_s += c.busy_s; st.marker_symbols += c.nsym;
            if (c.at_end) { chain_done = true; break; }
            // chunks between this one and the one it stopped at were no boundaries (or had none): dropped
            for (uint32_t i = chain_cur + 1; i < c.end_chunk; i++) {
                if (chunks[i].start_bit != kNone && chunks[i].start_bit != kPending) st.chunks_dropped++;
                chunks[i].dropped = true;
                if (chunks[i].done) chunks[i].release();
            }
            chain_cur = c.end_chunk;
            if (stream && stream->input_is_file_mapping && chain_cur < n_chunks && (chain_cur & 15) == 0) {
                // everything before the head's first byte has been decoded for good (the 4 KiB page holding that byte may still be read)
                const uintptr_t a = ((uintptr_t)in + dropped_in + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)in + chunks[chain_cur].byte_begin) & ~(uintptr_t)4095;
                if (e > a) { (void)madvise((void *)a, e - a, MADV_DONTNEED); dropped_in = (uint64_t)(e - (uintptr_t)in); }
            }
        }
        if (rc == kOk && !chain_done && chain_cur >= n_chunks) rc = kCorrupt;
    };
    std::atomic<uint64_t> resolve_ns{0};
    auto resolve = [&](const ResolveTask &t) -> int {   // kOk / kNoMem
        const auto r0 = clk::now();
        struct Timer { std::atomic<uint64_t> &acc; clk::time_point t0; ~Timer() { acc += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count(); } } timer{resolve_ns, r0};
        Chunk &c = chunks[t.chunk];
        uint8_t *dst = fin.p + t.out_off;
        std::vector<CrcPiece> mine;
        try {
            if (c.nsym) {
                // 0..255 -> the byte, 32768 + w -> window byte w.  Sixteen symbols at a time: a group without a marker is packed to bytes
                // (SSE2, part of the x86-64 baseline), one with markers goes symbol by symbol through a table.  (FASTQ text keeps markers alive
                // for the whole chunk - every header and every quality run copies from the record before it, back to the unknown window - so
                // both paths matter; the pass costs 0.8 core-seconds per 3.15 GB of text beside 3.4 of decoding, profiles/r06c.)
                std::vector<uint8_t> lut(2 * kWin);
                for (uint32_t i = 0; i < 256; i++) lut[i] = (uint8_t)i;
                memcpy(lut.data() + kWin, t.win, kWin);
                const uint16_t *s = c.o16.base + kWin;
                const uint8_t *L = lut.data();
                size_t i = 0;
#if defined(__SSE2__)
                const __m128i hi = _mm_set1_epi16((short)0xFF00);
                for (; i + 16 <= c.nsym; i += 16) {
                    const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 8));
                    const __m128i any = _mm_and_si128(_mm_or_si128(a, b), hi);
                    if (_mm_movemask_epi8(_mm_cmpeq_epi8(any, _mm_setzero_si128())) == 0xFFFF) {
                        _mm_storeu_si128((__m128i *)(dst + i), _mm_packus_epi16(a, b));
                    } else {
                        for (int j = 0; j < 16; j++) dst[i + j] = L[s[i + j]];
                    }
                }
#endif
                for (; i < c.nsym; i++) dst[i] = L[s[i]];
            }
            if (c.nbyt) memcpy(dst + c.nsym, c.o8.base + kWin, c.nbyt);
            // CRC-32 of what was just written (still in the cache), cut at the member ends inside the chunk
            const uint64_t tot = c.nsym + c.nbyt;
            uint64_t at = 0;
            for (size_t m = 0; m <= c.members.size(); m++) {
                const uint64_t to = m < c.members.size() ? c.members[m].out_off : tot;
                if (to > at) mine.push_back({t.out_off + at, to - at, crc.run(dst + at, (size_t)(to - at))});
                at = to;
            }
        } catch (...) { free(t.win); c.release(); return kNoMem; }
        free(t.win);
        const uint64_t end_off = t.out_off + c.nsym + c.nbyt;
        c.release();
        std::lock_guard<std::mutex> g(mu);
        try { for (const CrcPiece &p : mine) pieces.push_back(p); } catch (...) { return kNoMem; }
        for (auto &pl : placed) if (pl.first == end_off && !pl.second) { pl.second = true; break; }
        while (!placed.empty() && placed.front().second) { ready = placed.front().first; placed.pop_front(); }
        if (stream) stream->ready = ready;
        return kOk;
    };
    const auto tC = clk::now();
    std::atomic<uint64_t> search_ns{0};
    auto search_chunk = [&](uint32_t i) -> uint64_t {
        const auto s0 = clk::now();
        uint64_t end = i + 1 < n_chunks ? chunks[i + 1].byte_begin : n;
        if (end > chunks[i].byte_begin + (1u << 20)) end = chunks[i].byte_begin + (1u << 20);
        const uint64_t sb = find_block(in, n, chunks[i].byte_begin, end, &pool);
        search_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - s0).count();
        return sb;
    };
    // the starts of the first chunks are found before anything is decoded (a few milliseconds, in parallel): chunk 0's decoder starts at once
    // and would otherwise be past its successors' starts before their searches report; every later chunk is searched by the worker that takes
    // it, far ahead of the decoder that will stop at it
    {
        const uint32_t first = n_chunks < max_ahead + 1 ? n_chunks : max_ahead + 1;
        std::atomic<uint32_t> nx{1};
        if (first > 1)
            run_parallel(n_threads < first - 1 ? n_threads : first - 1, [&] {
                for (uint32_t i; (i = nx.fetch_add(1)) < first;) __atomic_store_n(&chunks[i].start_bit, search_chunk(i), __ATOMIC_RELEASE);
            });
    }
    run_parallel(n_threads, [&] {
        std::unique_lock<std::mutex> lk(mu);
        try {
            for (;;) {
                if (stream && stream->cancel && rc == kOk) rc = kCancelled;
                if (rc != kOk) break;
                if (!rq.empty()) {
                    const ResolveTask t = rq.front();
                    rq.pop_front();
                    resolving++;
                    lk.unlock();
                    const int r = resolve(t);
                    lk.lock();
                    resolving--;
                    if (r != kOk && rc == kOk) rc = r;
                    cv.notify_all();
                    continue;
                }
                if (chain_done) { if (resolving == 0) break; cv.wait(lk); continue; }
                if (stream && chain_cur < n_chunks && chunks[chain_cur].done) {   // a head that waited for the consumer
                    advance_chain();
                    if (rc != kOk || !rq.empty() || chain_done) { cv.notify_all(); continue; }
                }
                if (chain_cur < n_chunks && chunks[chain_cur].deferred && !chunks[chain_cur].in_flight) {
                    // the head outgrew its cap as a speculative chunk: now that it is the head it decodes against the whole limit (nothing
                    // else can be placed before it anyway)
                    Chunk &c = chunks[chain_cur];
                    const uint32_t i = chain_cur;
                    c.deferred = false; c.in_flight = true;
                    lk.unlock();
                    decode_chunk(in, n, chunks, i, fin_limit, 0, &pool);
                    lk.lock();
                    c.in_flight = false; c.done = true;
                    st.chunks_deferred++;
                    advance_chain();
                    cv.notify_all();
                    continue;
                }
                if (next_decode < n_chunks && next_decode < chain_cur + max_ahead) {
                    const uint32_t i = next_decode++;
                    Chunk &c = chunks[i];
                    if (c.dropped) { c.done = true; continue; }
                    c.in_flight = true;
                    const bool is_head = i == chain_cur;
                    const bool search = i > 0 && c.start_bit == kPending;
                    if (search) __atomic_store_n(&c.searching, 1u, __ATOMIC_RELEASE);
                    if (i > 0 && !search && c.start_bit == kNone) { c.done = true; c.dropped = true; c.in_flight = false; continue; }   // searched up front: no boundary
                    lk.unlock();
                    if (search) {
                        const uint64_t sb = search_chunk(i);
                        lk.lock();
                        __atomic_store_n(&c.start_bit, sb, __ATOMIC_RELEASE);   // (read by the decoders of earlier chunks at their block boundaries)
                        const bool skip = sb == kNone || c.dropped;
                        if (skip) { c.done = true; c.dropped = true; c.in_flight = false; continue; }
                        lk.unlock();
                    }
                    // A chunk that is not at the head of the chain decodes against a cap: high-ratio data (or a false block start decoding
                    // garbage) would otherwise let every one of the 2 T + 2 chunks in flight grow towards the global limit.  One that
                    // outgrows the cap gives its buffers back and is decoded again when it is the head.
                    const uint64_t cap = (is_head || fin_limit < kSpecCap) ? fin_limit : kSpecCap;
                    decode_chunk(in, n, chunks, i, cap, 0, &pool);
                    lk.lock();
                    c.in_flight = false;
                    if (c.status == kTooLarge && cap < fin_limit && !c.dropped) {
                        c.status = kOk; c.deferred = true;
                        cv.notify_all();
                        continue;
                    }
                    c.done = true;
                    if (c.dropped) c.release();
                    advance_chain();
                    cv.notify_all();
                    continue;
                }
                cv.wait(lk);
            }
        } catch (...) {   // bad_alloc in a container: an error of the run, not of the process
            if (!lk.owns_lock()) lk.lock();
            if (rc == kOk) rc = kNoMem;
        }
        cv.notify_all();
    });
    st.search_s = (double)search_ns.load() * 1e-9 / (double)n_threads;   // (summed over the workers; as wall seconds of n_threads)
    st.chunks = n_chunks;
    st.resolve_busy_s = (double)resolve_ns.load() * 1e-9;
    st.decode_s = secs(tC, clk::now());
    for (ResolveTask &t : rq) free(t.win);
    for (Chunk &c : chunks) c.release();
    // member CRCs and sizes, in stream order
    const auto tD = clk::now();
    if (rc == kOk) {
        std::sort(pieces.begin(), pieces.end(), [](const CrcPiece &a, const CrcPiece &b) { return a.off < b.off; });
        size_t pi = 0;
        uint64_t mstart = 0;
        for (const auto &e : ends) {
            uint32_t mc = 0; bool any = false;
            while (pi < pieces.size() && pieces[pi].off + pieces[pi].len <= e.first) {
                mc = any ? (uint32_t)crc32_combine(mc, pieces[pi].crc, (z_off_t)pieces[pi].len) : pieces[pi].crc;
                any = true; pi++;
            }
            if (mc != e.second.crc || (uint32_t)(e.first - mstart) != e.second.isize) { rc = kCorrupt; break; }
            mstart = e.first; st.members++;
        }
        if (rc == kOk && (pi != pieces.size() || mstart != total)) rc = kCorrupt;   // output after the last member's trailer: cannot be
    }
    st.crc_s = secs(tD, clk::now());
    if (stats) *stats = st;
    if (stream) {   // the mapping stays (the consumer may still be reading): pgz_stream_release
        std::lock_guard<std::mutex> g(mu);
        stream->rc = rc; stream->finished = true;
        cv.notify_all();
        return rc;
    }
    if (rc != kOk) { block_free(fin); return rc; }
    // give back the address space that was not needed
    const size_t keep = (size_t)(((total ? total : 1) + 4095) & ~(uint64_t)4095);
    if (keep < fin.cap) munmap(fin.p + keep, fin.cap - keep);
    *out = fin.p; *out_n = total;
    return kOk;
}

}  // namespace

int pgz_inflate(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, uint8_t **out, uint64_t *out_n, PgzStats *stats)
{
    return inflate_impl(in, n, n_threads, limit, out, out_n, stats, nullptr);
}

int pgz_inflate_stream(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, PgzStream *s, PgzStats *stats)
{
    if (!s) return kCorrupt;
    const int rc = inflate_impl(in, n, n_threads, limit, nullptr, nullptr, stats, s);
    if (!s->finished) {   // an early return (bad header, no memory for the mapping): the consumer must hear of it too
        std::lock_guard<std::mutex> g(s->mu);
        s->rc = rc; s->finished = true;
        s->cv.notify_all();
    }
    return rc;
}

void pgz_stream_release(PgzStream *s)
{
    if (!s || !s->map) return;
    Block b;
    b.p = s->map; b.cap = (size_t)s->map_bytes;
    s->map = nullptr; s->base = nullptr; s->map_bytes = 0;
    // The range is address space (its pages went back behind the parsers); unmapping it means walking the page tables of everything the run
    // touched with the address space locked.  One range is kept for the next streamed run of the process instead.
    // (DONTNEED gives pages back, not page tables: a range that has been written far - 2 MB of tables per GB of text - is not worth keeping)
    if (s->ready <= ((uint64_t)8 << 30)) {
        std::lock_guard<std::mutex> g(g_text_mu);
        if (!g_text.p) { g_text = b; return; }
    }
    std::vector<Block> one(1, b);
    release_async(std::move(one));
}

}  // namespace ntk
