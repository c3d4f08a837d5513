// ntk_fastx_api.cpp — C ABI of the CPU record reader and the whole-file pipeline, written on top of the public
// batch face only (include/needletail_amd.h).
#include "../../include/needletail_amd.h"

#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "ntk_fastx.hpp"

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
    rec->qual = fr.qual; rec->qual_len = fr.qual_len; rec->format = (uint32_t)fr.format; rec->reserved = 0;
    rec->line = fr.line; rec->num_bases = fr.num_bases;
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

void ntk_reader_close(ntk_reader *h) { delete h; }

int ntk_scan_reader(ntk_ctx *ctx, ntk_reader *h, const ntk_params *p, uint64_t batch_bytes, uint32_t n_batches,
                    uint64_t *n_records, uint64_t *n_bases)
{
    if (!ctx || !h || !p || batch_bytes < 1024 || n_batches < 2 || n_batches > 64) return NTK_ERR_BAD_ARG;
    std::vector<ntk_batch *> batches(n_batches, nullptr);
    int rc = NTK_OK;
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
        int a = ntk_batch_append(batches[cur], rec.seq, rec.seq_len, p->pre);
        if (a == NTK_ERR_CAPACITY) {
            if ((rc = ntk_batch_submit(ctx, batches[cur], p)) != NTK_OK) break;  // async: H2D copy + scan
            cur = (cur + 1) % n_batches;
            if ((rc = ntk_batch_wait(ctx, batches[cur])) != NTK_OK) break;        // the oldest batch in flight
            a = ntk_batch_append(batches[cur], rec.seq, rec.seq_len, p->pre);     // still too big -> CAPACITY
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

}  // extern "C"
