// ntk_fastx_api.cpp — C ABI of the CPU record reader and the whole-file pipeline, written on top of the public
// batch face only (include/needletail_amd.h).
#include "../../include/needletail_amd.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
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

// Where the range workers get their pieces of plain text from: a buffer cut in advance (ntk_scan_buffer_parallel), or the growing output of
// a gzip inflater that is still running (ntk_scan_file_parallel on a .gz file: StreamPieces below).
struct PieceSource {
    virtual ~PieceSource() {}
    virtual bool next(const uint8_t **p, uint64_t *len) = 0;   // false: no more pieces (or an error: see rc())
    virtual void parsed(const uint8_t *, uint64_t) {}          // every record of the piece has been copied into a pinned batch
    virtual int rc() { return NTK_OK; }
};

void release_pages(const uint8_t *p, uint64_t len)
{
    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = (uintptr_t)(p + len) & ~(uintptr_t)4095;
    if (e > a) (void)madvise((void *)a, e - a, MADV_DONTNEED);
}

// the input is cut into several pieces per thread, handed out on demand: host threads do not run at one speed (SMT siblings,
// the far socket), and with one static range per thread the slowest one set the time (profiles/r02e/pipeline.txt)
struct StaticPieces : PieceSource {
    const uint8_t *data = nullptr;
    const uint64_t *cut = nullptr;
    uint32_t n_pieces = 0;
    std::atomic<uint32_t> next_piece{0};
    bool next(const uint8_t **p, uint64_t *len) override
    {
        for (;;) {
            const uint32_t i = next_piece.fetch_add(1);
            if (i >= n_pieces) return false;
            if (cut[i + 1] <= cut[i]) continue;
            *p = data + cut[i]; *len = cut[i + 1] - cut[i];
            return true;
        }
    }
};

uint64_t next_record_start(const uint8_t *d, uint64_t n, uint64_t from, int format);

// Pieces of the text a gzip inflater is still producing (ntk::PgzStream: [0, ready) is final and grows).  A piece is cut at a record start
// about `target` bytes after the previous one as soon as that much text (+ the few lines the cut looks ahead) is ready; what has been handed
// out counts as consumed - the inflater stays within its window of that point - and a worker gives the pages of a piece back to the kernel
// when it has parsed it, so that the text of a file of any size passes through a bounded amount of memory.
struct StreamPieces : PieceSource {
    ntk::PgzStream *s = nullptr;
    uint64_t cursor = 0, target = (uint64_t)4 << 20;
    int format = -2;   // -2: not looked at yet
    int err = NTK_OK;
    bool next(const uint8_t **p, uint64_t *len) override
    {
        std::unique_lock<std::mutex> lk(s->mu);
        for (;;) {
            if (err != NTK_OK || s->cancel) return false;
            if (s->finished && s->rc != 0) return false;   // (the caller maps the inflater's status)
            const uint64_t ready = s->ready, have = ready - cursor;
            if (format == -2 && (ready >= 2 || s->finished)) {
                if (ready < 2) { err = NTK_ERR_PARSE; s->cv.notify_all(); return false; }   // EmptyFile (reference src/parser/mod.rs:88-91)
                format = s->base[0] == '>' ? ntk::kFasta : (s->base[0] == '@' ? ntk::kFastq : -1);
                if (format < 0) { err = NTK_ERR_PARSE; s->cv.notify_all(); return false; }
            }
            if (s->finished && have == 0) return false;
            if (format >= 0 && (s->finished || have >= target + (64 << 10))) {
                uint64_t cut = ready;
                if (have > target + (64 << 10) || !s->finished) {
                    cut = next_record_start(s->base, ready, cursor + target, format);
                    if (cut >= ready && !s->finished) cut = 0;   // no record start in what is ready (a very long record): wait for more
                }
                if (cut > cursor) {
                    *p = s->base + cursor; *len = cut - cursor;
                    cursor = cut;
                    s->consumed = cursor;
                    s->cv.notify_all();   // the inflater may place more
                    return true;
                }
                if (have > s->window / 2 && s->window < ((uint64_t)1 << 40)) { s->window *= 2; s->cv.notify_all(); }   // one record longer than the window: let it grow
            }
            s->cv.wait(lk);
        }
    }
    void parsed(const uint8_t *p, uint64_t len) override { release_pages(p, len); }
    int rc() override { return err; }
};

struct Shared {
    ntk_ctx *ctx; const ntk_params *p; uint64_t batch_bytes;
    std::mutex mu;                       // the ctx is not thread-safe: submit / wait / acquire are serialised
    std::atomic<int> rc{NTK_OK};
    std::atomic<uint64_t> nrec{0}, nbases{0};
    PieceSource *src = nullptr;
    bool stats = false;                  // NTK_OPT_PIPE_STATS: per-thread phase times on stderr
    std::atomic<int64_t> first_submit_ns{-1};   // steady-clock time of the first batch submitted
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
    const uint8_t *piece = nullptr; uint64_t piece_len = 0;
    ntk_record rec;
    const bool stats = sh->stats;
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
            int64_t none = -1;
            sh->first_submit_ns.compare_exchange_strong(none, (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t2.time_since_epoch()).count());
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
            if (!sh->src->next(&piece, &piece_len)) { rc = sh->src->rc(); break; }
            if ((rc = ntk_reader_open_memory(piece, piece_len, &rd)) != NTK_OK) break;
        }
        const int s = ntk_reader_next(rd, &rec);
        if (s == NTK_EOF) {
            ntk_reader_close(rd); rd = nullptr;
            sh->src->parsed(piece, piece_len);
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

namespace {
uint64_t ctx_option(ntk_ctx *ctx, int option, uint64_t fallback)
{
    uint64_t v = 0;
    return ntk_ctx_get_option(ctx, option, &v) == NTK_OK && v ? v : fallback;
}

// The range workers over a piece source: n_threads parser threads, each with its two pinned batches.
int run_workers(ntk_ctx *ctx, const ntk_params *p, uint64_t batch_bytes, uint32_t n_threads, PieceSource *src, uint64_t *n_records, uint64_t *n_bases,
                int64_t *first_submit_ns = nullptr)
{
    Shared sh;
    sh.ctx = ctx; sh.p = p; sh.batch_bytes = batch_bytes; sh.src = src;
    sh.stats = ctx_option(ctx, NTK_OPT_PIPE_STATS, 0) != 0;
    std::vector<std::thread> th;
    try { for (uint32_t i = 1; i < n_threads; i++) th.emplace_back(range_worker, &sh); } catch (...) {}   // fewer threads than asked for: the pieces are pulled
    range_worker(&sh);
    for (auto &t : th) t.join();
    if (n_records) *n_records = sh.nrec.load();
    if (n_bases) *n_bases = sh.nbases.load();
    if (first_submit_ns) *first_submit_ns = sh.first_submit_ns.load();
    return sh.rc.load();
}

struct BufferPieces : StaticPieces {
    bool release = false;   // the buffer is the library's own scratch (a gzip file inflated into an anonymous mapping): a worker hands the pages of
                            // a piece back as soon as it has parsed it - unmapping 3 GB in one go at the end took a third of the call (profiles/r05g)
    void parsed(const uint8_t *p, uint64_t len) override { if (release) release_pages(p, len); }
};
}  // namespace

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
    BufferPieces src;
    src.data = data; src.cut = cut.data(); src.n_pieces = n_pieces; src.release = release_input;
    // one worker per non-empty piece at most (on small or odd inputs many cuts collapse onto the end of the buffer, and a
    // worker without a piece would still acquire its two pinned batches)
    uint32_t live_pieces = 0;
    for (uint32_t i = 0; i < n_pieces; i++) live_pieces += cut[i + 1] > cut[i];
    if (n_threads > live_pieces) n_threads = live_pieces ? live_pieces : 1;
    return run_workers(ctx, p, batch_bytes, n_threads, &src, n_records, n_bases);
}

namespace {
// gzip for the parallel producer.  libdeflate (a whole-buffer decoder; its runtime .so ships with the image, its headers do not; the
// three entry points below are its stable v1 ABI) inflates the members of BLOCK gzip files; ordinary gzip streams go through this
// library's own speculative parallel inflater (ntk_pgzip.cpp).  ntk_gunzip hands the whole text over in one buffer (up to the in-memory
// limit); ntk_scan_file_parallel consumes it WHILE it is inflated (scan_gzip_streamed below): bounded memory, no limit on the file's size.
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
const Deflate *deflate_lib() { static const Deflate lib; return &lib; }

// BGZF (block gzip: bgzip / htslib): every member is <= 64 KiB and carries its own compressed size in a 'BC' extra
// subfield and its plain size in the trailer, so the members can be located without inflating anything and inflated
// independently.  Fills `blocks`; false when the stream is not pure BGZF.
struct BgzfBlock { uint64_t in_off, in_len, out_off, out_len; };
bool bgzf_index(const uint8_t *in, uint64_t n, std::vector<BgzfBlock> *blocks, uint64_t *total_out)
{
    uint64_t ip = 0, op = 0;
    try {
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
    } catch (...) { return false; }
    *total_out = op;
    return !blocks->empty();
}

constexpr size_t kBgzfGroup = 16;   // blocks (<= 1 MiB of text) per grab

// Inflates the members of a BGZF file into buf (their final offsets are known from the index).  stream != nullptr: progressive - groups
// are taken in file order, none is started further than the stream's window beyond what the consumer has taken, and [0, ready) is the
// prefix of finished groups.  Returns 0, or 1 for corrupt data.
int bgzf_inflate(const uint8_t *in, const std::vector<BgzfBlock> &blocks, uint64_t total, uint8_t *buf, uint32_t nt, ntk::PgzStream *stream)
{
    const Deflate &lib = *deflate_lib();
    std::atomic<int> bad{0};
    std::atomic<size_t> next{0};
    const size_t n_groups = (blocks.size() + kBgzfGroup - 1) / kBgzfGroup;
    std::vector<uint8_t> done;
    size_t ready_group = 0;
    if (stream) { try { done.assign(n_groups, 0); } catch (...) { return 3; } }
    auto work = [&]() {
        void *d = lib.alloc();
        if (!d) { bad = 3; if (stream) { std::lock_guard<std::mutex> g(stream->mu); stream->cv.notify_all(); } return; }
        for (size_t g; !bad && (g = next.fetch_add(1)) < n_groups;) {
            const size_t i = g * kBgzfGroup;
            if (stream) {   // stay within the window of the consumer
                std::unique_lock<std::mutex> lk(stream->mu);
                while (!bad && !stream->cancel && blocks[i].out_off > stream->consumed + stream->window) stream->cv.wait(lk);
                if (stream->cancel) { bad = 4; break; }
                const uint64_t backlog = blocks[i].out_off > stream->consumed ? blocks[i].out_off - stream->consumed : 0;
                if (backlog > stream->peak_backlog) stream->peak_backlog = backlog;
            }
            for (size_t j = i; j < i + kBgzfGroup && j < blocks.size(); j++) {
                const BgzfBlock &b = blocks[j];
                size_t used = 0, made = 0;
                if (lib.gzip_ex(d, in + b.in_off, b.in_len, buf + b.out_off, b.out_len, &used, &made) != 0 || made != b.out_len) { bad = 1; break; }
            }
            if (stream) {
                std::lock_guard<std::mutex> lk(stream->mu);
                done[g] = 1;
                while (ready_group < n_groups && done[ready_group]) ready_group++;
                stream->ready = ready_group < n_groups ? blocks[ready_group * kBgzfGroup].out_off : total;
                stream->cv.notify_all();
            }
        }
        lib.free_(d);
        if (bad && stream) { std::lock_guard<std::mutex> g(stream->mu); stream->cv.notify_all(); }
    };
    std::vector<std::thread> th;
    try { for (uint32_t t = 1; t < nt; t++) th.emplace_back(work); } catch (...) {}
    work();
    for (auto &t : th) t.join();
    return bad.load();
}

int map_pgz_status(int r) { return r == 0 ? NTK_OK : (r == 2 ? NTK_ERR_UNSUPPORTED : (r == 3 ? NTK_ERR_NOMEM : NTK_ERR_PARSE)); }

void fill_info(ntk_gunzip_info *info, uint32_t route, uint32_t nt, const ntk::PgzStats &st)
{
    if (!info) return;
    info->route = route; info->threads = nt; info->chunks = st.chunks; info->chunks_dropped = st.chunks_dropped; info->members = st.members;
    info->search_s = st.search_s; info->decode_s = st.decode_s; info->decode_busy_s = st.decode_busy_s; info->crc_s = st.crc_s;
    info->marker_symbols = st.marker_symbols; info->chunks_deferred = st.chunks_deferred; info->resolve_busy_s = st.resolve_busy_s;
}

// The whole gzip file inflated into one buffer (every member, CRC-32 and ISIZE checked; reference: MultiGzDecoder,
// src/parser/mod.rs:95-108).  NTK_OK with *out (an anonymous mapping: ntk::pgz_free) / *out_n; NTK_ERR_UNSUPPORTED (larger than the
// limit), NTK_ERR_PARSE (corrupt or truncated data), NTK_ERR_NOMEM.  Routes:
//   1  block gzip (BGZF): the members carry their sizes, are located without inflating anything and inflated in parallel
//      (libdeflate when it can be loaded, else route 2)
//   2  an ordinary gzip stream, n_threads > 1: speculative parallel inflate (ntk_pgzip.cpp: chunks enter the stream at block
//      boundaries with the 32 KiB before them as unknowns, resolved afterwards)
//   3  the same decoder on one thread
int inflate_whole(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, uint8_t **out, uint64_t *out_n, ntk_gunzip_info *info)
{
    const uint32_t nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (info) { memset(info, 0, sizeof(*info)); info->threads = nt; }
    if (deflate_lib()->ok) {   // BGZF: members inflate in parallel straight to their final offsets
        std::vector<BgzfBlock> blocks;
        uint64_t total = 0;
        if (bgzf_index(in, n, &blocks, &total)) {
            if (total > limit) return NTK_ERR_UNSUPPORTED;
            uint8_t *buf = ntk::pgz_alloc(total);
            if (!buf) return NTK_ERR_NOMEM;
            const int bad = bgzf_inflate(in, blocks, total, buf, nt, nullptr);
            if (bad) { ntk::pgz_free(buf, total); return bad == 3 ? NTK_ERR_NOMEM : NTK_ERR_PARSE; }
            *out = buf; *out_n = total;
            if (info) { info->route = 1; info->chunks = (uint32_t)blocks.size(); info->members = (uint32_t)blocks.size(); }
            return NTK_OK;
        }
    }
    ntk::PgzStats st;
    const int r = ntk::pgz_inflate(in, n, nt, limit, out, out_n, &st);
    fill_info(info, nt > 1 ? 2 : 3, nt, st);
    return map_pgz_status(r);
}

thread_local ntk_gunzip_info t_last_info;   // ntk_scan_file_info

// A gzip file through the parallel producer WITHOUT holding its text: the inflater (route 1 or 2 above, on n_threads threads) runs beside
// the parser threads, which take pieces of the text as it becomes final, copy the records into pinned batches (H2D copies and scans overlap
// on the GPU's side as for any batch) and give the pages back.  BASELINE.json configs[4]: "CPU decompress overlapped with GPU k-mer via pinned
// async copies"; the reference's arrangement is one zlib thread feeding the parser (src/parser/mod.rs:95-108).  Same errors as the reader:
// every member is read and checked; truncated / corrupt data is NTK_ERR_PARSE (reported when the decoder gets there: batches scanned before
// that have been added to the accumulators, as a caller of the reference's reader has processed the records before the Io error).
int scan_gzip_streamed(ntk_ctx *ctx, const uint8_t *gz, uint64_t n, const ntk_params *p, uint64_t batch_bytes, uint32_t n_threads,
                       uint64_t *n_records, uint64_t *n_bases)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    if (n_records) *n_records = 0;
    if (n_bases) *n_bases = 0;
    const uint32_t nt = n_threads > 64 ? 64 : n_threads;
    ntk_params p_local = *p;   // NTK_FLAG_RESET: once for the whole scan, not per batch
    if (p_local.flags & NTK_FLAG_RESET) { const int r = ntk_accum_reset(ctx); if (r != NTK_OK) return r; p_local.flags &= ~NTK_FLAG_RESET; }
    ntk::PgzStream S;
    S.window = ctx_option(ctx, NTK_OPT_GZ_STREAM_WINDOW_BYTES, (uint64_t)512 << 20);
    S.input_is_file_mapping = true;   // (ntk_scan_file_parallel's own private mapping of the file)
    if (S.window < ((uint64_t)8 << 20)) S.window = (uint64_t)8 << 20;
    ntk_gunzip_info info;
    memset(&info, 0, sizeof(info));
    info.threads = nt; info.streamed = 1;
    // the parsers are a fraction of the inflater's work (10 M reads: 0.75 core-seconds of parsing and packing against 4.8 of inflating,
    // profiles/r05g); they sleep while no text is ready, so they are threads on top of the inflater's, not taken from them
    uint32_t n_parse = nt / 4;
    if (n_parse < 2) n_parse = 2;
    if (n_parse > 8) n_parse = 8;
    info.parse_threads = n_parse;
    std::vector<BgzfBlock> blocks;
    uint64_t total = 0;
    const bool bgzf = deflate_lib()->ok && bgzf_index(gz, n, &blocks, &total);
    int rc_inflate = 0;
    ntk::PgzStats st;
    std::thread producer;
    uint8_t *bgzf_buf = nullptr;
    if (bgzf) {
        bgzf_buf = ntk::pgz_alloc(total);
        if (!bgzf_buf) return NTK_ERR_NOMEM;
        S.base = bgzf_buf;
        info.route = 1; info.chunks = (uint32_t)blocks.size(); info.members = (uint32_t)blocks.size();
    }
    try {
        producer = std::thread([&] {
            if (bgzf) {
                rc_inflate = bgzf_inflate(gz, blocks, total, bgzf_buf, nt, &S);
                std::lock_guard<std::mutex> g(S.mu);
                S.rc = rc_inflate == 4 ? 0 : rc_inflate; S.finished = true;
                S.cv.notify_all();
            } else {
                // the address range reserved for the text: 4 TiB unless the option says otherwise (it is touched as it is written and given
                // back behind the parsers; the inflater halves the reservation until the system grants it)
                const uint64_t limit = ctx_option(ctx, NTK_OPT_GZ_INMEM_LIMIT_BYTES, (uint64_t)1 << 42);
                rc_inflate = ntk::pgz_inflate_stream(gz, n, nt, limit, &S, &st);
            }
        });
    } catch (...) { if (bgzf_buf) ntk::pgz_free(bgzf_buf, total); return NTK_ERR_NOMEM; }
    StreamPieces src;
    src.s = &S;
    // pieces: small enough that the first batch leaves early and the parsers share the text evenly, large enough to amortise a reader each
    src.target = batch_bytes < ((uint64_t)2 << 20) ? ((uint64_t)2 << 20) : (batch_bytes > ((uint64_t)16 << 20) ? ((uint64_t)16 << 20) : batch_bytes);
    if (src.target > S.window / 4) src.target = S.window / 4;
    int64_t first_ns = -1;
    int rc = run_workers(ctx, &p_local, batch_bytes, n_parse, &src, n_records, n_bases, &first_ns);
    if (rc != NTK_OK) {   // the parsers gave up: the inflater must not run on into memory nobody drains
        std::lock_guard<std::mutex> g(S.mu);
        S.cancel = true;
        S.cv.notify_all();
    }
    const auto t_workers = clk::now();
    producer.join();
    const auto t_joined = clk::now();
    if (ctx_option(ctx, NTK_OPT_PIPE_STATS, 0))
        fprintf(stderr, "streamed gzip: parsers done after %.1f ms, inflater joined after %.1f ms (decode+resolve pipeline %.1f ms, crc %.1f ms)\n",
                std::chrono::duration<double>(t_workers - t0).count() * 1e3, std::chrono::duration<double>(t_joined - t0).count() * 1e3, st.decode_s * 1e3, st.crc_s * 1e3);
    if (!bgzf) fill_info(&info, 2, nt, st);
    info.streamed = 1; info.parse_threads = n_parse;
    info.peak_backlog_bytes = S.peak_backlog;
    info.text_bytes = S.ready;
    info.total_s = std::chrono::duration<double>(clk::now() - t0).count();
    info.first_batch_s = first_ns < 0 ? 0.0 : (double)(first_ns - (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t0.time_since_epoch()).count()) * 1e-9;
    t_last_info = info;
    if (bgzf) ntk::pgz_free(bgzf_buf, total); else ntk::pgz_stream_release(&S);
    if (ctx_option(ctx, NTK_OPT_PIPE_STATS, 0))
        fprintf(stderr, "streamed gzip: text released after %.1f ms\n", std::chrono::duration<double>(clk::now() - t0).count() * 1e3);
    if (rc_inflate != 0 && rc_inflate != 4) return bgzf ? (rc_inflate == 3 ? NTK_ERR_NOMEM : NTK_ERR_PARSE) : map_pgz_status(rc_inflate);   // (4: cancelled above)
    return rc;
}
}  // namespace

int ntk_scan_file_parallel(ntk_ctx *ctx, const char *path, const ntk_params *p, uint64_t batch_bytes, uint32_t n_threads,
                           uint64_t *n_records, uint64_t *n_bases)
{
    if (!ctx || !path || !p || batch_bytes < 1024 || n_threads < 1 || n_threads > 1024) return NTK_ERR_BAD_ARG;
    memset(&t_last_info, 0, sizeof(t_last_info));
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
        if (n_threads >= 2 && st.st_size >= 18) {
            (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            rc = scan_gzip_streamed(ctx, data, (uint64_t)st.st_size, p, batch_bytes, n_threads, n_records, n_bases);
        } else {   // one thread: inflate, then parse (nothing to overlap with)
            uint8_t *plain = nullptr; uint64_t plain_n = 0;
            rc = inflate_whole(data, (uint64_t)st.st_size, n_threads, ctx_option(ctx, NTK_OPT_GZ_INMEM_LIMIT_BYTES, (uint64_t)16 << 30), &plain, &plain_n, &t_last_info);
            if (rc == NTK_OK) {
                rc = scan_buffer_parallel_impl(ctx, plain, plain_n, p, batch_bytes, n_threads, n_records, n_bases, true);
                ntk::pgz_free(plain, plain_n);
            }
        }
    } else {
        rc = ntk_scan_buffer_parallel(ctx, data, (uint64_t)st.st_size, p, batch_bytes, n_threads, n_records, n_bases);
    }
    // (unmapping half a gigabyte of touched file pages takes ~10 ms: off the caller's path, like the inflater's own mappings)
    const size_t map_bytes = (size_t)st.st_size;
    try { std::thread([m, map_bytes] { munmap(m, map_bytes); }).detach(); } catch (...) { munmap(m, map_bytes); }
    return rc;
}

int ntk_scan_file_info(ntk_gunzip_info *info)
{
    if (!info) return NTK_ERR_BAD_ARG;
    *info = t_last_info;
    return NTK_OK;
}

int ntk_gunzip(const uint8_t *gz, uint64_t n, uint32_t n_threads, uint8_t **out, uint64_t *out_n, ntk_gunzip_info *info)
{
    if (!gz || !out || !out_n) return NTK_ERR_BAD_ARG;
    *out = nullptr; *out_n = 0;
    if (n < 18 || gz[0] != 0x1F || gz[1] != 0x8B) return NTK_ERR_PARSE;
    return inflate_whole(gz, n, n_threads, (uint64_t)16 << 30, out, out_n, info);
}

void ntk_gunzip_free(uint8_t *out, uint64_t out_n) { ntk::pgz_free(out, out_n); }

}  // extern "C"
