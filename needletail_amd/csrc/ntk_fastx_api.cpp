// ntk_fastx_api.cpp — C ABI of the CPU record reader and the whole-file pipeline, written on top of the public
// batch face only (include/needletail_amd.h).
#include "../../include/needletail_amd.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <chrono>
#include <thread>

#include <new>
#include <string>
#include <vector>

#include "ntk_fastx.hpp"
#include "ntk_pgzip.hpp"

struct ntk_reader {
    ntk::FastxReader r;
};

namespace {
void copy_str(const std::string &s, char *dst, uint64_t cap)
{
    if (!dst || !cap) return;
    const uint64_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(dst, s.data(), n);
    dst[n] = 0;
}
}  // namespace

extern "C" {

int ntk_reader_open_file(const char *path, ntk_reader **out)
{
    if (!path || !out) return NTK_ERR_BAD_ARG;
    ntk_reader *h = new (std::nothrow) ntk_reader();
    if (!h) return NTK_ERR_NOMEM;
    *out = h;
    return h->r.open_file(path) ? NTK_OK : NTK_ERR_PARSE;
}

int ntk_reader_open_memory(const uint8_t *data, uint64_t n, ntk_reader **out)
{
    if ((!data && n) || !out) return NTK_ERR_BAD_ARG;
    ntk_reader *h = new (std::nothrow) ntk_reader();
    if (!h) return NTK_ERR_NOMEM;
    *out = h;
    return h->r.open_memory(data, n) ? NTK_OK : NTK_ERR_PARSE;
}

int ntk_reader_next(ntk_reader *h, ntk_record *rec)
{
    if (!h || !rec) return NTK_ERR_BAD_ARG;
    ntk::FastxRecord fr;
    const int rc = h->r.next(&fr);
    if (rc < 0) return NTK_ERR_PARSE;
    if (rc == 0) return NTK_EOF;
    rec->id = fr.id; rec->id_len = fr.id_len; rec->seq = fr.seq; rec->seq_len = fr.seq_len;
    rec->qual = fr.qual; rec->qual_len = fr.qual_len; rec->format = (uint32_t)fr.format; rec->line_ending = (uint32_t)fr.line_ending;
    rec->line = fr.line; rec->num_bases = fr.num_bases; rec->byte = fr.byte;
    return NTK_OK;
}

int ntk_reader_error(ntk_reader *h, int *kind, uint64_t *line, char *msg, uint64_t msg_cap, char *id, uint64_t id_cap)
{
    if (!h) return NTK_ERR_BAD_ARG;
    if (kind) *kind = h->r.error_kind();
    if (line) *line = h->r.error_line();
    copy_str(h->r.error_msg(), msg, msg_cap);
    copy_str(h->r.error_id(), id, id_cap);
    return NTK_OK;
}

int ntk_reader_position(ntk_reader *h, uint64_t *line, uint64_t *byte, int *ending)
{
    if (!h) return NTK_ERR_BAD_ARG;
    if (line) *line = h->r.position_line();
    if (byte) *byte = h->r.position_byte();
    if (ending) *ending = h->r.line_ending();
    return NTK_OK;
}

void ntk_reader_close(ntk_reader *h) { delete h; }

namespace {
// FASTQ records carry their quality line into the batch when the parameters hold a quality cutoff (bits 15:8 of flags)
inline int append_record(ntk_batch *b, const ntk_record &rec, const ntk_params *p)
{
    const uint32_t cutoff = (p->flags >> 8) & 0xFFu;
    if (cutoff && rec.qual && rec.qual_len == rec.seq_len)
        return ntk_batch_append_quality(b, rec.seq, rec.qual, rec.seq_len, p->pre, cutoff);
    return ntk_batch_append(b, rec.seq, rec.seq_len, p->pre);
}

// A record longer than a whole batch (a chromosome-sized contig against a small batch_bytes): scanned through a one-off
// batch sized for it, so that the batch size is a tuning knob and never a limit on the input.  The accumulators are sums, so
// the order relative to the batches still in flight does not matter.
int scan_oversized_record(ntk_ctx *ctx, std::mutex *mu, const ntk_record &rec, const ntk_params *p)
{
    ntk_batch *big = nullptr;
    int rc = ntk_batch_acquire(ctx, rec.seq_len + 64, 2, &big);
    if (rc != NTK_OK) return rc;
    rc = append_record(big, rec, p);
    if (rc == NTK_OK) {
        if (mu) { std::lock_guard<std::mutex> g(*mu); rc = ntk_batch_submit(ctx, big, p); }
        else rc = ntk_batch_submit(ctx, big, p);
    }
    const int w = ntk_batch_wait(ctx, big);
    ntk_batch_release(ctx, big);
    return rc != NTK_OK ? rc : w;
}
}  // namespace

int ntk_scan_reader(ntk_ctx *ctx, ntk_reader *h, const ntk_params *p, uint64_t batch_bytes, uint32_t n_batches,
                    uint64_t *n_records, uint64_t *n_bases)
{
    if (!ctx || !h || !p || batch_bytes < 1024 || n_batches < 2 || n_batches > 64) return NTK_ERR_BAD_ARG;
    std::vector<ntk_batch *> batches(n_batches, nullptr);
    int rc = NTK_OK;
    ntk_params p_local = *p;   // NTK_FLAG_RESET means "this whole scan starts a new result": once, here, not per batch
    if (p_local.flags & NTK_FLAG_RESET) { if ((rc = ntk_accum_reset(ctx)) != NTK_OK) return rc; p_local.flags &= ~NTK_FLAG_RESET; }
    p = &p_local;
    const uint64_t max_records = batch_bytes / 32 + 16;  // offsets are bookkeeping only
    for (auto &b : batches)
        if ((rc = ntk_batch_acquire(ctx, batch_bytes, max_records, &b)) != NTK_OK) break;
    uint64_t nrec = 0, nbases = 0;
    uint32_t cur = 0;
    ntk_record rec;
    while (rc == NTK_OK) {
        const int s = ntk_reader_next(h, &rec);
        if (s == NTK_EOF) break;
        if (s != NTK_OK) { rc = s; break; }
        int a = append_record(batches[cur], rec, p);
        if (a == NTK_ERR_CAPACITY) {
            if ((rc = ntk_batch_submit(ctx, batches[cur], p)) != NTK_OK) break;  // async: H2D copy + scan
            cur = (cur + 1) % n_batches;
            if ((rc = ntk_batch_wait(ctx, batches[cur])) != NTK_OK) break;        // the oldest batch in flight
            a = append_record(batches[cur], rec, p);
            if (a == NTK_ERR_CAPACITY) a = scan_oversized_record(ctx, nullptr, rec, p);  // longer than an empty batch
        }
        if (a != NTK_OK) { rc = a; break; }
        nrec++; nbases += rec.num_bases;
    }
    if (rc == NTK_OK) rc = ntk_batch_submit(ctx, batches[cur], p);
    for (auto b : batches) {
        if (!b) continue;
        const int w = ntk_batch_wait(ctx, b);
        if (rc == NTK_OK && w != NTK_OK) rc = w;
        ntk_batch_release(ctx, b);
    }
    if (n_records) *n_records = nrec;
    if (n_bases) *n_bases = nbases;
    return rc;
}


/* ---- parallel producer: one parser thread per file range ------------------------------------------------ */
namespace {

// First record start at or after `from` in a plain FASTA/FASTQ byte range (chunked parsing).
// FASTA: the byte after "\n" that is '>'.  FASTQ: a line starting with '@' whose line+2 starts with '+' (a quality line
// may start with '@', but then its line+2 is a sequence line, which cannot start with '+').
uint64_t next_record_start(const uint8_t *d, uint64_t n, uint64_t from, int format)
{
    if (from == 0) return 0;
    uint64_t p = from;
    while (p < n) {
        const uint8_t *nl = (const uint8_t *)memchr(d + p, '\n', n - p);
        if (!nl) return n;
        p = (uint64_t)(nl - d) + 1;
        if (p >= n) return n;
        if (format == ntk::kFasta) {
            if (d[p] != '>') continue;
            // Not after a header line: a record without a sequence line (">empty\n>next") must stay in one piece with its
            // successor - a range that ENDS in a bare header is a truncated record to the reader (UnexpectedEnd), although
            // the whole file parses (reference src/parser/fasta.rs:291-367: the next '>' line ends the empty record).
            const uint8_t *prev_nl = p >= 2 ? (const uint8_t *)memrchr(d, '\n', p - 1) : nullptr;
            const uint64_t prev_line = prev_nl ? (uint64_t)(prev_nl - d) + 1 : 0;
            if (d[prev_line] == '>') continue;
            return p;
        }
        if (d[p] != '@') continue;
        const uint8_t *l1 = (const uint8_t *)memchr(d + p, '\n', n - p);
        if (!l1) return n;
        const uint8_t *l2 = (const uint8_t *)memchr(l1 + 1, '\n', n - (uint64_t)(l1 + 1 - d));
        if (!l2 || (uint64_t)(l2 + 1 - d) >= n) return n;
        if (l2[1] == '+') return p;
    }
    return n;
}

struct Shared {
    ntk_ctx *ctx; const ntk_params *p; uint64_t batch_bytes;
    std::mutex mu;                       // the ctx is not thread-safe: submit / wait / acquire are serialised
    std::atomic<int> rc{NTK_OK};
    std::atomic<uint64_t> nrec{0}, nbases{0};
    // the input is cut into several pieces per thread, handed out on demand: host threads do not run at one speed (SMT siblings,
    // the far socket), and with one static range per thread the slowest one set the time (profiles/r02e/pipeline.txt)
    const uint8_t *data = nullptr;
    const uint64_t *cut = nullptr;
    uint32_t n_pieces = 0;
    std::atomic<uint32_t> next_piece{0};
    // the input is the library's own scratch (a gzip file inflated into an anonymous mapping): a worker hands the pages of a piece back to
    // the kernel as soon as it has parsed it - unmapping 3 GB in one go at the end took a third of the whole call (profiles/r05g)
    bool release_input = false;
};

void range_worker(Shared *sh)
{
    ntk_reader *rd = nullptr;
    int rc = NTK_OK;
    ntk_batch *b[2] = {nullptr, nullptr};
    const uint64_t max_records = sh->batch_bytes / 32 + 16;
    // acquire / release take the ctx's own pool lock; they are not serialised with submit
    for (int i = 0; i < 2 && rc == NTK_OK; i++) rc = ntk_batch_acquire(sh->ctx, sh->batch_bytes, max_records, &b[i]);
    uint64_t nrec = 0, nbases = 0;
    int cur = 0;
    uint32_t piece = 0;
    ntk_record rec;
    static const bool stats = getenv("NTK_PIPE_STATS") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_begin = now();
    double t_lock = 0, t_submit = 0, t_wait = 0;
    int n_sub = 0;
    // Ramp-up: the first fills are cut short (1/8, 1/4, 1/2 of a batch) so that the first copies start after a fraction of a
    // batch's parse time instead of a whole one; from then on full batches.
    uint64_t filled = 0, soft = sh->batch_bytes / 8 < (256 << 10) ? (256 << 10) : sh->batch_bytes / 8;
    auto flush = [&]() -> int {   // submit the batch in the making, continue in the other one (once its last use has completed)
        const auto t0 = now();
        int r;
        {
            std::lock_guard<std::mutex> g(sh->mu);
            const auto t1 = now();
            r = ntk_batch_submit(sh->ctx, b[cur], sh->p);
            const auto t2 = now();
            t_lock += secs(t0, t1); t_submit += secs(t1, t2); n_sub++;
        }
        if (r != NTK_OK) return r;
        cur ^= 1;
        const auto t3 = now();
        r = ntk_batch_wait(sh->ctx, b[cur]);   // event wait only: no ctx state touched
        t_wait += secs(t3, now());
        filled = 0;
        if (soft < sh->batch_bytes) soft *= 2;
        return r;
    };
    while (rc == NTK_OK && sh->rc.load() == NTK_OK) {
        if (!rd) {   // next piece of the input (a batch in the making carries over: pieces are not flushed)
            const uint32_t i = sh->next_piece.fetch_add(1);
            if (i >= sh->n_pieces) break;
            if (sh->cut[i + 1] <= sh->cut[i]) continue;
            piece = i;
            if ((rc = ntk_reader_open_memory(sh->data + sh->cut[i], sh->cut[i + 1] - sh->cut[i], &rd)) != NTK_OK) break;
        }
        const int s = ntk_reader_next(rd, &rec);
        if (s == NTK_EOF) {
            ntk_reader_close(rd); rd = nullptr;
            if (sh->release_input) {   // every record of the piece has been copied into a pinned batch
                const uintptr_t a = ((uintptr_t)(sh->data + sh->cut[piece]) + 4095) & ~(uintptr_t)4095, e = (uintptr_t)(sh->data + sh->cut[piece + 1]) & ~(uintptr_t)4095;
                if (e > a) (void)madvise((void *)a, e - a, MADV_DONTNEED);
            }
            continue;
        }
        if (s != NTK_OK) { rc = s; break; }
        int a = append_record(b[cur], rec, sh->p);
        if (a == NTK_ERR_CAPACITY) {
            if ((rc = flush()) != NTK_OK) break;
            a = append_record(b[cur], rec, sh->p);
            if (a == NTK_ERR_CAPACITY) a = scan_oversized_record(sh->ctx, &sh->mu, rec, sh->p);
        }
        if (a != NTK_OK) { rc = a; break; }
        nrec++; nbases += rec.num_bases;
        filled += rec.seq_len + 1;
        if (filled >= soft && soft < sh->batch_bytes && (rc = flush()) != NTK_OK) break;
    }
    {
        std::lock_guard<std::mutex> g(sh->mu);
        if (rc == NTK_OK && b[cur]) rc = ntk_batch_submit(sh->ctx, b[cur], sh->p);
    }
    for (int i = 0; i < 2; i++) {
        if (!b[i]) continue;
        const int w = ntk_batch_wait(sh->ctx, b[i]);
        if (rc == NTK_OK && w != NTK_OK) rc = w;
        ntk_batch_release(sh->ctx, b[i]);
    }
    if (rd) ntk_reader_close(rd);
    if (stats) fprintf(stderr, "worker: total %.1f ms, lock %.1f, submit %.1f (%d), wait %.1f, parse+fill %.1f\n", secs(t_begin, now()) * 1e3, t_lock * 1e3,
                       t_submit * 1e3, n_sub, t_wait * 1e3, (secs(t_begin, now()) - t_lock - t_submit - t_wait) * 1e3);
    if (rc != NTK_OK) { int ok = NTK_OK; sh->rc.compare_exchange_strong(ok, rc); }
    sh->nrec += nrec; sh->nbases += nbases;
}

}  // namespace

int ntk_fastx_split_points(const uint8_t *data, uint64_t n, uint32_t n_pieces, uint64_t *cuts)
{
    if ((!data && n) || !cuts || n_pieces < 1) return NTK_ERR_BAD_ARG;
    if (n < 2) return NTK_ERR_PARSE;
    const int format = data[0] == '>' ? ntk::kFasta : (data[0] == '@' ? ntk::kFastq : -1);
    if (format < 0) return NTK_ERR_PARSE;
    cuts[0] = 0; cuts[n_pieces] = n;
    for (uint32_t i = 1; i < n_pieces; i++) {
        const uint64_t c = next_record_start(data, n, n / n_pieces * i, format);
        cuts[i] = c < cuts[i - 1] ? cuts[i - 1] : c;
    }
    return NTK_OK;
}

static int scan_buffer_parallel_impl(ntk_ctx *ctx, const uint8_t *data, uint64_t n, const ntk_params *p, uint64_t batch_bytes,
                                     uint32_t n_threads, uint64_t *n_records, uint64_t *n_bases, bool release_input);

int ntk_scan_buffer_parallel(ntk_ctx *ctx, const uint8_t *data, uint64_t n, const ntk_params *p, uint64_t batch_bytes,
                             uint32_t n_threads, uint64_t *n_records, uint64_t *n_bases)
{
    return scan_buffer_parallel_impl(ctx, data, n, p, batch_bytes, n_threads, n_records, n_bases, false);
}

// release_input: `data` is an anonymous private mapping owned by the caller inside this library (the inflated text of a gzip file)
static int scan_buffer_parallel_impl(ntk_ctx *ctx, const uint8_t *data, uint64_t n, const ntk_params *p, uint64_t batch_bytes,
                                     uint32_t n_threads, uint64_t *n_records, uint64_t *n_bases, bool release_input)
{
    if (!ctx || (!data && n) || !p || batch_bytes < 1024 || n_threads < 1 || n_threads > 1024) return NTK_ERR_BAD_ARG;
    if (n_records) *n_records = 0;
    if (n_bases) *n_bases = 0;
    if (n < 2) return NTK_ERR_PARSE;                       // EmptyFile (reference src/parser/mod.rs:88-91)
    // compressed streams (gzip, bzip2, xz, zstd) are sequential: use ntk_scan_reader
    if ((data[0] == 0x1F && data[1] == 0x8B) || (data[0] == 0x42 && data[1] == 0x5A) || (data[0] == 0xFD && data[1] == 0x37) ||
        (data[0] == 0x28 && data[1] == 0xB5)) return NTK_ERR_UNSUPPORTED;
    ntk_params p_local = *p;   // NTK_FLAG_RESET: once for the whole scan, not per batch
    if (p_local.flags & NTK_FLAG_RESET) { const int r = ntk_accum_reset(ctx); if (r != NTK_OK) return r; p_local.flags &= ~NTK_FLAG_RESET; }
    p = &p_local;
    // a thread is only worth its two pinned batches if it has several batches of input to parse
    const uint64_t worth = n / (4 * batch_bytes) + 1;
    if (n_threads > worth) n_threads = (uint32_t)worth;
    if (n_threads > 64) n_threads = 64;   // more parser threads than that only queue up on the submit lock (profiles/r02e/pipeline.txt)
    // pieces of >= 1 MiB, about eight per thread
    uint64_t want_pieces = (uint64_t)n_threads * 8;
    if (want_pieces > n / (1 << 20) + 1) want_pieces = n / (1 << 20) + 1;
    if (want_pieces < n_threads) want_pieces = n_threads;
    const uint32_t n_pieces = (uint32_t)want_pieces;
    std::vector<uint64_t> cut(n_pieces + 1, n);
    if (ntk_fastx_split_points(data, n, n_pieces, cut.data()) != NTK_OK) return NTK_ERR_PARSE;
    Shared sh;
    sh.ctx = ctx; sh.p = p; sh.batch_bytes = batch_bytes;
    sh.data = data; sh.cut = cut.data(); sh.n_pieces = n_pieces; sh.release_input = release_input;
    // one worker per non-empty piece at most (on small or odd inputs many cuts collapse onto the end of the buffer, and a
    // worker without a piece would still acquire its two pinned batches)
    uint32_t live_pieces = 0;
    for (uint32_t i = 0; i < n_pieces; i++) live_pieces += cut[i + 1] > cut[i];
    if (n_threads > live_pieces) n_threads = live_pieces ? live_pieces : 1;
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n_threads; i++) th.emplace_back(range_worker, &sh);
    for (auto &t : th) t.join();
    if (n_records) *n_records = sh.nrec.load();
    if (n_bases) *n_bases = sh.nbases.load();
    return sh.rc.load();
}

namespace {
// Whole-file gzip for the parallel producer: the file is inflated into memory by all threads (inflate_whole below) and the plain
// text is then parsed in parallel.  libdeflate (a whole-buffer decoder; its runtime .so ships with the image, its headers do not; the
// three entry points below are its stable v1 ABI) inflates the members of BLOCK gzip files; ordinary gzip streams go through this
// library's own speculative parallel inflater (ntk_pgzip.cpp).  Outputs beyond the in-memory limit are NTK_ERR_UNSUPPORTED here: the
// streaming ntk_scan_reader (zlib, one thread - what the reference does) is the way for those.
struct Deflate {
    void *(*alloc)() = nullptr;
    int (*gzip_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    bool ok = false;
    Deflate()
    {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        gzip_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        free_ = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        ok = alloc && gzip_ex && free_;
    }
};

// BGZF (block gzip: bgzip / htslib): every member is <= 64 KiB and carries its own compressed size in a 'BC' extra
// subfield and its plain size in the trailer, so the members can be located without inflating anything and inflated
// independently.  Fills `blocks`; false when the stream is not pure BGZF.
struct BgzfBlock { uint64_t in_off, in_len, out_off, out_len; };
bool bgzf_index(const uint8_t *in, uint64_t n, std::vector<BgzfBlock> *blocks, uint64_t *total_out)
{
    uint64_t ip = 0, op = 0;
    while (ip < n) {
        if (n - ip < 28 || in[ip] != 0x1F || in[ip + 1] != 0x8B || in[ip + 2] != 8 || !(in[ip + 3] & 4)) return false;
        const uint32_t xlen = in[ip + 10] | (in[ip + 11] << 8);
        if (n - ip < 12 + (uint64_t)xlen + 8) return false;
        uint64_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = in + ip + 12 + x;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (uint64_t)(sf[4] | (sf[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + (uint64_t)xlen + 8 || bsize > n - ip) return false;
        const uint8_t *tr = in + ip + bsize - 4;
        const uint64_t isize = (uint64_t)tr[0] | ((uint64_t)tr[1] << 8) | ((uint64_t)tr[2] << 16) | ((uint64_t)tr[3] << 24);
        blocks->push_back(BgzfBlock{ip, bsize, op, isize});
        ip += bsize; op += isize;
    }
    *total_out = op;
    return !blocks->empty();
}

// The whole gzip file inflated into one buffer (every member, CRC-32 and ISIZE checked; reference: MultiGzDecoder,
// src/parser/mod.rs:95-108).  NTK_OK with *out (an anonymous mapping: ntk::pgz_free) / *out_n; NTK_ERR_UNSUPPORTED (larger than the
// limit), NTK_ERR_PARSE (corrupt or truncated data), NTK_ERR_NOMEM.  Routes:
//   1  block gzip (BGZF): the members carry their sizes, are located without inflating anything and inflated in parallel
//      (libdeflate when it can be loaded, else route 2)
//   2  an ordinary gzip stream, n_threads > 1: speculative parallel inflate (ntk_pgzip.cpp: chunks enter the stream at block
//      boundaries with the 32 KiB before them as unknowns, resolved afterwards)
//   3  the same decoder on one thread
int inflate_whole(const uint8_t *in, uint64_t n, uint32_t n_threads, uint8_t **out, uint64_t *out_n, ntk_gunzip_info *info)
{
    static const Deflate lib;
    uint64_t limit = (uint64_t)16 << 30;
    if (const char *e = getenv("NTK_GZ_INMEM_LIMIT_BYTES")) limit = strtoull(e, nullptr, 10);
    const uint32_t nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (info) { memset(info, 0, sizeof(*info)); info->threads = nt; }
    if (lib.ok) {   // BGZF: members inflate in parallel straight to their final offsets
        std::vector<BgzfBlock> blocks;
        uint64_t total = 0;
        if (bgzf_index(in, n, &blocks, &total)) {
            if (total > limit) return NTK_ERR_UNSUPPORTED;
            uint8_t *buf = ntk::pgz_alloc(total);
            if (!buf) return NTK_ERR_NOMEM;
            std::atomic<int> bad{0};
            std::atomic<size_t> next{0};
            auto work = [&]() {
                void *d = lib.alloc();
                if (!d) { bad = 1; return; }
                for (size_t i; !bad && (i = next.fetch_add(16)) < blocks.size();)   // 16 blocks (<= 1 MiB) per grab
                    for (size_t j = i; j < i + 16 && j < blocks.size(); j++) {
                        const BgzfBlock &b = blocks[j];
                        size_t used = 0, made = 0;
                        if (lib.gzip_ex(d, in + b.in_off, b.in_len, buf + b.out_off, b.out_len, &used, &made) != 0 ||
                            made != b.out_len) { bad = 1; break; }
                    }
                lib.free_(d);
            };
            std::vector<std::thread> th;
            try { for (uint32_t t = 1; t < nt; t++) th.emplace_back(work); } catch (...) {}
            work();
            for (auto &t : th) t.join();
            if (bad) { ntk::pgz_free(buf, total); return NTK_ERR_PARSE; }
            *out = buf; *out_n = total;
            if (info) { info->route = 1; info->chunks = (uint32_t)blocks.size(); info->members = (uint32_t)blocks.size(); }
            return NTK_OK;
        }
    }
    ntk::PgzStats st;
    const int r = ntk::pgz_inflate(in, n, nt, limit, out, out_n, &st);
    if (info) {
        info->route = nt > 1 ? 2 : 3; info->chunks = st.chunks; info->chunks_dropped = st.chunks_dropped; info->members = st.members;
        info->search_s = st.search_s; info->decode_s = st.decode_s; info->decode_busy_s = st.decode_busy_s; info->crc_s = st.crc_s;
        info->marker_symbols = st.marker_symbols;
    }
    return r == 0 ? NTK_OK : (r == 2 ? NTK_ERR_UNSUPPORTED : (r == 3 ? NTK_ERR_NOMEM : NTK_ERR_PARSE));
}
}  // namespace

int ntk_scan_file_parallel(ntk_ctx *ctx, const char *path, const ntk_params *p, uint64_t batch_bytes, uint32_t n_threads,
                           uint64_t *n_records, uint64_t *n_bases)
{
    if (!ctx || !path || !p) return NTK_ERR_BAD_ARG;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return NTK_ERR_PARSE;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 2) { close(fd); return NTK_ERR_PARSE; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return NTK_ERR_NOMEM;
    const uint8_t *data = (const uint8_t *)m;
    int rc;
    if (data[0] == 0x1F && data[1] == 0x8B) {
        uint8_t *plain = nullptr; uint64_t plain_n = 0;
        rc = inflate_whole(data, (uint64_t)st.st_size, n_threads, &plain, &plain_n, nullptr);
        if (rc == NTK_OK) {
            rc = scan_buffer_parallel_impl(ctx, plain, plain_n, p, batch_bytes, n_threads, n_records, n_bases, true);
            ntk::pgz_free(plain, plain_n);
        }
    } else {
        rc = ntk_scan_buffer_parallel(ctx, data, (uint64_t)st.st_size, p, batch_bytes, n_threads, n_records, n_bases);
    }
    munmap(m, (size_t)st.st_size);
    return rc;
}

int ntk_gunzip(const uint8_t *gz, uint64_t n, uint32_t n_threads, uint8_t **out, uint64_t *out_n, ntk_gunzip_info *info)
{
    if (!gz || !out || !out_n) return NTK_ERR_BAD_ARG;
    *out = nullptr; *out_n = 0;
    if (n < 18 || gz[0] != 0x1F || gz[1] != 0x8B) return NTK_ERR_PARSE;
    return inflate_whole(gz, n, n_threads, out, out_n, info);
}

void ntk_gunzip_free(uint8_t *out, uint64_t out_n) { ntk::pgz_free(out, out_n); }

}  // extern "C"
