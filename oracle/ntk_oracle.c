/*
 * ntk_oracle.c — CPU restatement of needletail's per-sequence k-mer hot path (see ntk_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the product library.
 * Every function cites the reference lines (relative to /root/reference/) it follows.
 */
#include "ntk_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================== *
 *  src/sequence.rs
 * ======================================================================================== */

/* src/sequence.rs:19-62.  The reference matches on (byte, allow_iupac) in this priority order;
 * whitespace maps to the sentinel ' ' which is then not pushed (:48,53-55). */
size_t ntko_normalize(const uint8_t *seq, size_t n, int allow_iupac, uint8_t *out, int *changed)
{
    size_t w = 0;
    int any = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = seq[i], nc;
        int ch;
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'N': case '-': /* :25 */
            nc = c; ch = 0; break;
        case 'a': nc = 'A'; ch = 1; break;                          /* :26 */
        case 'c': nc = 'C'; ch = 1; break;                          /* :27 */
        case 'g': nc = 'G'; ch = 1; break;                          /* :28 */
        case 't': case 'u': case 'U': nc = 'T'; ch = 1; break;      /* :30 */
        case '.': case '~': nc = '-'; ch = 1; break;                /* :32 */
        case 'B': case 'D': case 'H': case 'V': case 'R':
        case 'Y': case 'S': case 'W': case 'K': case 'M':           /* :34-36 */
            if (allow_iupac) { nc = c; ch = 0; } else { nc = 'N'; ch = 1; }
            break;
        case 'b': case 'd': case 'h': case 'v': case 'r':
        case 'y': case 's': case 'w': case 'k': case 'm':           /* :37-46 */
            if (allow_iupac) { nc = (uint8_t)(c - 32); ch = 1; } else { nc = 'N'; ch = 1; }
            break;
        case ' ': case '\t': case '\r': case '\n':                  /* :48 */
            nc = ' '; ch = 1; break;
        default:                                                    /* :50 */
            nc = 'N'; ch = 1; break;
        }
        any = any || ch;                                            /* :52 */
        if (nc != ' ') out[w++] = nc;                               /* :53-55 */
    }
    if (changed) *changed = any;                                    /* :57-61 */
    return w;
}

/* src/sequence.rs:165-191: drop every '\r' and '\n'; Borrowed when there is none. */
size_t ntko_strip_returns(const uint8_t *seq, size_t n, uint8_t *out, int *borrowed)
{
    size_t w = 0;
    int found = 0;
    for (size_t i = 0; i < n; i++) {
        if (seq[i] == '\r' || seq[i] == '\n') { found = 1; continue; }
        out[w++] = seq[i];
    }
    if (borrowed) *borrowed = !found;
    return w;
}

/* src/sequence.rs:68-105. */
uint8_t ntko_complement(uint8_t n)
{
    switch (n) {
    case 'a': return 't'; case 'A': return 'T';
    case 'c': return 'g'; case 'C': return 'G';
    case 'g': return 'c'; case 'G': return 'C';
    case 't': return 'a'; case 'T': return 'A';
    case 'r': return 'y'; case 'y': return 'r';
    case 'k': return 'm'; case 'm': return 'k';
    case 'b': return 'v'; case 'v': return 'b';
    case 'd': return 'h'; case 'h': return 'd';
    case 's': return 's'; case 'w': return 'w';
    case 'R': return 'Y'; case 'Y': return 'R';
    case 'K': return 'M'; case 'M': return 'K';
    case 'B': return 'V'; case 'V': return 'B';
    case 'D': return 'H'; case 'H': return 'D';
    case 'S': return 'S'; case 'W': return 'W';
    default: return n; /* :103 anything else passes through */
    }
}

/* src/sequence.rs:202-208: iter().rev().map(complement). */
void ntko_reverse_complement(const uint8_t *seq, size_t n, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = ntko_complement(seq[n - 1 - i]);
}

/* src/sequence.rs:110-134. */
int ntko_canonical(const uint8_t *seq, size_t n, uint8_t *out)
{
    int enough = 0, original_was_canonical = 0;
    uint8_t *buf = (uint8_t *)malloc(n ? n : 1);
    size_t pushed = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t rn = ntko_complement(seq[n - 1 - i]); /* :117 */
        uint8_t c = seq[i];
        buf[pushed++] = rn;                           /* :118 */
        if (!enough && c < rn) { original_was_canonical = 1; break; } /* :119-121 */
        else if (!enough && rn < c) enough = 1;                       /* :122-124 */
    }
    int used_rc = (!original_was_canonical && enough);               /* :127-133 */
    memcpy(out, used_rc ? buf : seq, n);
    free(buf);
    return used_rc;
}

static int slice_lt(const uint8_t *a, const uint8_t *b, size_t n)
{
    return memcmp(a, b, n) < 0; /* equal-length slices: Rust's `<` on [u8] is bytewise lexicographic */
}

/* src/sequence.rs:139-152. */
void ntko_minimizer(const uint8_t *seq, size_t n, size_t length, uint8_t *out)
{
    uint8_t *rc = (uint8_t *)malloc(n ? n : 1);
    ntko_reverse_complement(seq, n, rc);               /* :140 */
    memcpy(out, seq, length);                          /* :141 */
    for (size_t i = 0; i + length <= n; i++) {         /* :143 windows zip windows */
        if (slice_lt(seq + i, out, length)) memcpy(out, seq + i, length); /* :144-146 */
        if (slice_lt(rc + i, out, length)) memcpy(out, rc + i, length);   /* :147-149 */
    }
    free(rc);
}

/* src/sequence.rs:285-296. */
size_t ntko_quality_mask(const uint8_t *seq, size_t n_seq, const uint8_t *qual, size_t n_qual,
                         uint8_t score, uint8_t *out)
{
    size_t n = n_seq < n_qual ? n_seq : n_qual;
    for (size_t i = 0; i < n; i++) out[i] = qual[i] < score ? (uint8_t)'N' : seq[i];
    return n;
}

/* ======================================================================================== *
 *  src/kmer.rs
 * ======================================================================================== */

int ntko_is_good_base(uint8_t c) /* src/kmer.rs:6-8 */
{
    return c == 'a' || c == 'c' || c == 'g' || c == 't' || c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

size_t ntko_kmers_count(size_t n, uint8_t k) /* src/kmer.rs:33-40 */
{
    return (size_t)k > n ? 0 : n - k + 1;
}

/* src/kmer.rs:84-108.  Note :100-101: kmer_len is zeroed BEFORE `start_pos += kmer_len + 1`, so a bad
 * base advances the start by exactly one and the window is rescanned from there. */
static int ck_update_position(ntko_canonical_kmers *it, int initial)
{
    if (it->start_pos + it->k > it->len) return 0;                 /* :86-88 */
    size_t kmer_len, stop_len;
    if (initial) { kmer_len = 0; stop_len = (size_t)(it->k - 1); } /* :90-94 */
    else { kmer_len = (size_t)(it->k - 1); stop_len = it->k; }
    while (kmer_len < stop_len) {                                  /* :96 */
        if (ntko_is_good_base(it->buffer[it->start_pos + kmer_len])) {
            kmer_len += 1;
        } else {
            kmer_len = 0;
            it->start_pos += kmer_len + 1;
            if (it->start_pos + it->k > it->len) return 0;
        }
    }
    return 1;
}

void ntko_ck_new(ntko_canonical_kmers *it, const uint8_t *buffer, size_t len,
                 const uint8_t *rc_buffer, size_t rc_len, uint8_t k) /* src/kmer.rs:73-82 */
{
    it->k = k; it->start_pos = 0; it->buffer = buffer; it->len = len;
    it->rc_buffer = rc_buffer; it->rc_len = rc_len;
    ck_update_position(it, 1);
}

int ntko_ck_next(ntko_canonical_kmers *it, size_t *pos, const uint8_t **slice, int *is_rc) /* :114-129 */
{
    if (!ck_update_position(it, 0)) return 0;
    size_t p = it->start_pos;
    it->start_pos += 1;
    const uint8_t *result = it->buffer + p;                            /* :121 */
    const uint8_t *rc_result = it->rc_buffer + (it->rc_len - p - it->k); /* :123 */
    if (slice_lt(result, rc_result, it->k)) { *pos = p; *slice = result; *is_rc = 0; } /* :124-125 */
    else { *pos = p; *slice = rc_result; *is_rc = 1; }                                 /* :126-127 */
    return 1;
}

/* ======================================================================================== *
 *  src/bitkmer.rs
 * ======================================================================================== */

int ntko_nuc2bit(uint8_t nuc) /* src/bitkmer.rs:5-18 */
{
    switch (nuc) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

static uint64_t kmask(uint8_t k) /* 2^(2k) - 1, src/bitkmer.rs:31 (k = 32 -> all ones) */
{
    return k >= 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
}

int ntko_extend_kmer(ntko_bitkmer *kmer, uint8_t new_char) /* src/bitkmer.rs:26-36 */
{
    int c = ntko_nuc2bit(new_char);
    if (c < 0) return 0;
    uint64_t nk = (kmer->seq << 2) + (uint64_t)c;
    kmer->seq = nk & kmask(kmer->k);
    return 1;
}

/* src/bitkmer.rs:39-70 (same shape as kmer.rs update_position; :61-63 zero the k-mer and advance by one). */
static int bit_update_position(size_t *start_pos, ntko_bitkmer *kmer, const uint8_t *buffer, size_t len,
                               int initial)
{
    if (*start_pos + kmer->k > len) return 0;
    size_t kmer_len, stop_len;
    if (initial) { kmer_len = 0; stop_len = (size_t)(kmer->k - 1); }
    else { kmer_len = (size_t)(kmer->k - 1); stop_len = kmer->k; }
    while (kmer_len < stop_len) {
        if (ntko_extend_kmer(kmer, buffer[*start_pos + kmer_len])) {
            kmer_len += 1;
        } else {
            kmer_len = 0;
            kmer->seq = 0;
            *start_pos += kmer_len + 1;
            if (*start_pos + kmer->k > len) return 0;
        }
    }
    return 1;
}

void ntko_bnk_new(ntko_bit_nucl_kmer *it, const uint8_t *slice, size_t len, uint8_t k, int canonical)
{ /* src/bitkmer.rs:80-91 */
    it->cur_kmer.seq = 0; it->cur_kmer.k = k; it->start_pos = 0;
    it->buffer = slice; it->len = len; it->canonical = canonical;
    bit_update_position(&it->start_pos, &it->cur_kmer, slice, len, 1);
}

int ntko_bnk_next(ntko_bit_nucl_kmer *it, size_t *pos, ntko_bitkmer *kmer, int *was_rc)
{ /* src/bitkmer.rs:97-108 */
    if (!bit_update_position(&it->start_pos, &it->cur_kmer, it->buffer, it->len, 0)) return 0;
    it->start_pos += 1;
    *pos = it->start_pos - 1;
    if (it->canonical) {
        *kmer = ntko_bit_canonical(it->cur_kmer, was_rc);
    } else {
        *kmer = it->cur_kmer; *was_rc = 0;
    }
    return 1;
}

ntko_bitkmer ntko_bit_reverse_complement(ntko_bitkmer kmer) /* src/bitkmer.rs:112-132 */
{
    uint64_t x = kmer.seq;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = ((x >> 32) & 0x00000000FFFFFFFFull) | ((x & 0x00000000FFFFFFFFull) << 32);
    x ^= 0xFFFFFFFFFFFFFFFFull;
    /* :130 `>>= 2 * (32 - k)`; k = 0 would shift by 64 (UB / panic in the reference) - callers reject k = 0 */
    unsigned sh = 2u * (32u - kmer.k);
    x = sh >= 64 ? 0 : x >> sh;
    ntko_bitkmer r = { x, kmer.k };
    return r;
}

ntko_bitkmer ntko_bit_canonical(ntko_bitkmer kmer, int *was_rc) /* src/bitkmer.rs:136-143 */
{
    ntko_bitkmer rc = ntko_bit_reverse_complement(kmer);
    if (kmer.seq > rc.seq) { if (was_rc) *was_rc = 1; return rc; }
    if (was_rc) *was_rc = 0;
    return kmer;
}

ntko_bitkmer ntko_bit_minimizer(ntko_bitkmer kmer, uint8_t m) /* src/bitkmer.rs:146-162 */
{
    uint64_t nk = kmer.seq, lowest = ~(uint64_t)0, bitmask = kmask(m);
    for (int i = 0; i <= (int)kmer.k - (int)m; i++) {
        uint64_t cur = bitmask & nk;
        if (cur < lowest) lowest = cur;
        ntko_bitkmer t = { bitmask & nk, kmer.k }; /* :155 rc taken at length k, not m */
        uint64_t cr = ntko_bit_reverse_complement(t).seq;
        if (cr < lowest) lowest = cr;
        nk >>= 2;
    }
    ntko_bitkmer r = { lowest, kmer.k };
    return r;
}

void ntko_bitmer_to_bytes(ntko_bitkmer kmer, uint8_t *out) /* src/bitkmer.rs:164-186 */
{
    static const char L[4] = { 'A', 'C', 'G', 'T' };
    uint64_t nk = kmer.seq;
    unsigned offset = (unsigned)(kmer.k - 1) * 2;
    uint64_t bitmask = (uint64_t)3 << offset;
    for (unsigned i = 0; i < kmer.k; i++) {
        out[i] = (uint8_t)L[(nk & bitmask) >> offset];
        nk <<= 2;
    }
}

ntko_bitkmer ntko_bytes_to_bitmer(const uint8_t *kmer, uint8_t k) /* src/bitkmer.rs:288-296 */
{
    ntko_bitkmer b = { 0, k };
    for (unsigned i = 0; i < k; i++) ntko_extend_kmer(&b, kmer[i]);
    return b;
}

/* ======================================================================================== *
 *  bulk drivers
 * ======================================================================================== */

size_t ntko_canonical_kmers_all(const uint8_t *buffer, size_t len, const uint8_t *rc, size_t rc_len,
                                uint8_t k, uint64_t *pos_out, uint8_t *is_rc_out, size_t cap)
{
    ntko_canonical_kmers it;
    ntko_ck_new(&it, buffer, len, rc, rc_len, k);
    size_t n = 0, pos; const uint8_t *sl; int f;
    while (ntko_ck_next(&it, &pos, &sl, &f)) {
        if (n < cap) { if (pos_out) pos_out[n] = pos; if (is_rc_out) is_rc_out[n] = (uint8_t)f; }
        n++;
    }
    return n;
}

size_t ntko_bit_kmers_all(const uint8_t *slice, size_t len, uint8_t k, int canonical,
                          uint64_t *pos_out, uint64_t *val_out, uint8_t *was_rc_out, size_t cap)
{
    ntko_bit_nucl_kmer it;
    ntko_bnk_new(&it, slice, len, k, canonical);
    size_t n = 0, pos; ntko_bitkmer km; int f;
    while (ntko_bnk_next(&it, &pos, &km, &f)) {
        if (n < cap) {
            if (pos_out) pos_out[n] = pos;
            if (val_out) val_out[n] = km.seq;
            if (was_rc_out) was_rc_out[n] = (uint8_t)f;
        }
        n++;
    }
    return n;
}

/* ======================================================================================== *
 *  reduced statistic
 * ======================================================================================== */

void ntko_stats_clear(ntko_stats *s) { memset(s, 0, sizeof(*s)); }

void ntko_stats_merge(ntko_stats *d, const ntko_stats *s)
{
    d->n_total += s->n_total; d->n_fwd += s->n_fwd; d->n_rc += s->n_rc;
    d->sum += s->sum; d->xr ^= s->xr;
    for (int i = 0; i < NTKO_HIST_BINS; i++) d->hist[i] += s->hist[i];
}

static inline void stats_emit(ntko_stats *s, uint64_t value, int flag, unsigned shift)
{
    s->n_total++;
    if (flag) s->n_rc++; else s->n_fwd++;
    s->sum += value;
    s->xr ^= value;
    s->hist[value >> shift]++;
}

static unsigned hist_shift(uint8_t k)
{
    unsigned p = k < NTKO_HIST_MAX_P ? k : NTKO_HIST_MAX_P;
    return 2u * (k - p);
}

/* Two buffers a thread may keep across records (the "arena" variant of the CPU baseline: NOT the reference's behaviour - the
 * reference allocates per record, sequence.rs:20,176,202-208 - it only shows what the allocator costs next to the literal chain). */
typedef struct { uint8_t *a, *b; size_t cap_a, cap_b; } ntko_scratch;
static uint8_t *scratch_get(uint8_t **p, size_t *cap, size_t n)
{
    if (*cap < n || !*p) { free(*p); *cap = n + n / 2 + 64; *p = (uint8_t *)malloc(*cap); }
    return *p;
}

static int reduce_record_impl(ntko_stats *s, const uint8_t *seq, size_t n, uint8_t k, int path, int pre, ntko_scratch *sc);
int ntko_reduce_record(ntko_stats *s, const uint8_t *seq, size_t n, uint8_t k, int path, int pre)
{
    return reduce_record_impl(s, seq, n, k, path, pre, NULL);
}

static int reduce_record_impl(ntko_stats *s, const uint8_t *seq, size_t n, uint8_t k, int path, int pre, ntko_scratch *sc)
{
    if (k < 1 || k > 32) return -1;
    unsigned shift = hist_shift(k);
    /* pre-step: the reference allocates a Vec for normalize (sequence.rs:20) / strip_returns (:176) */
    uint8_t *tmp = NULL;
    const uint8_t *cur = seq;
    size_t cn = n;
    if (pre == NTKO_PRE_STRIP_RETURNS) {
        tmp = sc ? scratch_get(&sc->a, &sc->cap_a, n ? n : 1) : (uint8_t *)malloc(n ? n : 1);
        int borrowed;
        cn = ntko_strip_returns(seq, n, tmp, &borrowed);
        cur = borrowed ? seq : tmp;
    } else if (pre == NTKO_PRE_NORMALIZE || pre == NTKO_PRE_NORMALIZE_IUPAC) {
        tmp = sc ? scratch_get(&sc->a, &sc->cap_a, n ? n : 1) : (uint8_t *)malloc(n ? n : 1);
        int changed;
        cn = ntko_normalize(seq, n, pre == NTKO_PRE_NORMALIZE_IUPAC, tmp, &changed);
        cur = changed ? tmp : seq; /* Cow::Borrowed when unchanged, sequence.rs:226-232 */
    } else if (pre != NTKO_PRE_NONE) {
        return -1;
    }

    if (path == NTKO_PATH_BYTES_CANONICAL) {
        uint8_t *rc = sc ? scratch_get(&sc->b, &sc->cap_b, cn ? cn : 1) : (uint8_t *)malloc(cn ? cn : 1); /* sequence.rs:202-208 returns a Vec */
        ntko_reverse_complement(cur, cn, rc);
        ntko_canonical_kmers it;
        ntko_ck_new(&it, cur, cn, rc, cn, k);
        size_t pos; const uint8_t *sl; int f;
        while (ntko_ck_next(&it, &pos, &sl, &f)) {
            uint64_t v = ntko_bytes_to_bitmer(sl, k).seq; /* 2-bit value of the yielded slice */
            stats_emit(s, v, f, shift);
        }
        if (!sc) free(rc);
    } else if (path == NTKO_PATH_BITS || path == NTKO_PATH_BITS_CANONICAL) {
        ntko_bit_nucl_kmer it;
        ntko_bnk_new(&it, cur, cn, k, path == NTKO_PATH_BITS_CANONICAL);
        size_t pos; ntko_bitkmer km; int f;
        while (ntko_bnk_next(&it, &pos, &km, &f)) stats_emit(s, km.seq, f, shift);
    } else {
        if (!sc) free(tmp);
        return -1;
    }
    if (!sc) free(tmp);
    return 0;
}

/* The reference benchmark's own per-record loop (benches/benchmark.rs:32-41 / :55-64): walk the iterator and
 * count items and `!was_rc` items - no value is folded, exactly the work the reference times. */
int ntko_count_record(uint64_t *n_total, uint64_t *n_fwd, const uint8_t *seq, size_t n, uint8_t k, int path, int pre)
{
    if (k < 1 || (path != NTKO_PATH_BYTES_CANONICAL && k > 32)) return -1;
    uint8_t *tmp = NULL;
    const uint8_t *cur = seq;
    size_t cn = n;
    if (pre == NTKO_PRE_STRIP_RETURNS) {
        tmp = (uint8_t *)malloc(n ? n : 1);
        int borrowed;
        cn = ntko_strip_returns(seq, n, tmp, &borrowed);
        cur = borrowed ? seq : tmp;
    } else if (pre == NTKO_PRE_NORMALIZE || pre == NTKO_PRE_NORMALIZE_IUPAC) {
        tmp = (uint8_t *)malloc(n ? n : 1);
        int changed;
        cn = ntko_normalize(seq, n, pre == NTKO_PRE_NORMALIZE_IUPAC, tmp, &changed);
        cur = changed ? tmp : seq;
    }
    uint64_t nt = 0, nf = 0;
    if (path == NTKO_PATH_BYTES_CANONICAL) {
        uint8_t *rc = (uint8_t *)malloc(cn ? cn : 1);
        ntko_reverse_complement(cur, cn, rc);
        ntko_canonical_kmers it;
        ntko_ck_new(&it, cur, cn, rc, cn, k);
        size_t pos; const uint8_t *sl; int f;
        while (ntko_ck_next(&it, &pos, &sl, &f)) { nt++; nf += !f; }
        free(rc);
    } else {
        ntko_bit_nucl_kmer it;
        ntko_bnk_new(&it, cur, cn, k, path == NTKO_PATH_BITS_CANONICAL);
        size_t pos; ntko_bitkmer km; int f;
        while (ntko_bnk_next(&it, &pos, &km, &f)) { nt++; nf += !f; }
    }
    free(tmp);
    *n_total += nt; *n_fwd += nf;
    return 0;
}

typedef struct {
    uint64_t nt, nf;
    const uint8_t *seq; const uint64_t *offsets; size_t r0, r1, gap; uint8_t k; int path, pre;
} cnt_job;

static void *cnt_run(void *p)
{
    cnt_job *j = (cnt_job *)p;
    j->nt = j->nf = 0;
    for (size_t r = j->r0; r < j->r1; r++) {
        size_t b = j->offsets[r], e = j->offsets[r + 1];
        size_t len = e - b >= j->gap ? e - b - j->gap : 0;
        ntko_count_record(&j->nt, &j->nf, j->seq + b, len, j->k, j->path, j->pre);
    }
    return NULL;
}

int ntko_count_batch_mt(uint64_t *n_total, uint64_t *n_fwd, const uint8_t *seq, const uint64_t *offsets,
                        size_t n_records, size_t gap, uint8_t k, int path, int pre, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    cnt_job *jobs = (cnt_job *)calloc((size_t)n_threads, sizeof(cnt_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].seq = seq; jobs[t].offsets = offsets; jobs[t].gap = gap;
        jobs[t].r0 = n_records * (size_t)t / (size_t)n_threads;
        jobs[t].r1 = n_records * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].k = k; jobs[t].path = path; jobs[t].pre = pre;
        if (n_threads == 1) cnt_run(&jobs[t]); else pthread_create(&th[t], NULL, cnt_run, &jobs[t]);
    }
    *n_total = 0; *n_fwd = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        *n_total += jobs[t].nt; *n_fwd += jobs[t].nf;
    }
    free(jobs); free(th);
    return 0;
}

int ntko_reduce_batch(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                      size_t gap, uint8_t k, int path, int pre)
{
    for (size_t r = 0; r < n_records; r++) {
        size_t b = offsets[r], e = offsets[r + 1];
        size_t len = e - b >= gap ? e - b - gap : 0;
        int rcode = ntko_reduce_record(s, seq + b, len, k, path, pre);
        if (rcode) return rcode;
    }
    return 0;
}

typedef struct {
    ntko_stats st;
    const uint8_t *seq; const uint64_t *offsets; size_t r0, r1, gap; uint8_t k; int path, pre, rcode, reuse;
} mt_job;

static void *mt_run(void *p)
{
    mt_job *j = (mt_job *)p;
    ntko_stats_clear(&j->st);
    if (!j->reuse) {
        j->rcode = ntko_reduce_batch(&j->st, j->seq, j->offsets + j->r0, j->r1 - j->r0, j->gap, j->k, j->path, j->pre);
        return NULL;
    }
    ntko_scratch sc = {NULL, NULL, 0, 0};   /* the thread's own two buffers, kept across its records */
    j->rcode = 0;
    for (size_t r = j->r0; r < j->r1 && !j->rcode; r++) {
        size_t b = j->offsets[r], e = j->offsets[r + 1];
        size_t len = e - b >= j->gap ? e - b - j->gap : 0;
        j->rcode = reduce_record_impl(&j->st, j->seq + b, len, j->k, j->path, j->pre, &sc);
    }
    free(sc.a); free(sc.b);
    return NULL;
}

int ntko_reduce_batch_mt2(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                          size_t gap, uint8_t k, int path, int pre, int n_threads, int reuse_buffers);
int ntko_reduce_batch_mt(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                         size_t gap, uint8_t k, int path, int pre, int n_threads)
{
    return ntko_reduce_batch_mt2(s, seq, offsets, n_records, gap, k, path, pre, n_threads, 0);
}

/* reuse_buffers: every thread keeps its normalize / reverse-complement buffers across records instead of allocating them per
 * record as the reference does (cpu_baseline's "arena" variant: informational, not the reference's behaviour). */
int ntko_reduce_batch_mt2(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                          size_t gap, uint8_t k, int path, int pre, int n_threads, int reuse_buffers)
{
    if (n_threads < 1) n_threads = 1;
    mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof(mt_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    int rcode = 0;
    for (int t = 0; t < n_threads; t++) {
        jobs[t].seq = seq; jobs[t].offsets = offsets; jobs[t].gap = gap;
        jobs[t].r0 = n_records * (size_t)t / (size_t)n_threads;
        jobs[t].r1 = n_records * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].k = k; jobs[t].path = path; jobs[t].pre = pre; jobs[t].reuse = reuse_buffers;
        pthread_create(&th[t], NULL, mt_run, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].rcode) rcode = jobs[t].rcode;
        ntko_stats_merge(s, &jobs[t].st);
    }
    free(jobs); free(th);
    return rcode;
}

/* Second, independent formulation: a window ending at byte i is emitted iff the run of base bytes
 * ending at i is >= k; forward and reverse-complement values roll in registers. */
int ntko_reduce_fused(ntko_stats *s, const uint8_t *buf, size_t n, uint8_t k, int canonical,
                      int tie_rc, int accept_u)
{
    if (k < 1 || k > 32) return -1;
    unsigned shift = hist_shift(k);
    uint64_t mask = kmask(k), fwd = 0, rc = 0;
    unsigned rcshift = 2u * (k - 1u);
    size_t run = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = buf[i];
        int code = ntko_nuc2bit(c);
        if (code < 0 && accept_u && (c == 'U' || c == 'u')) code = 3;
        if (code < 0) { run = 0; fwd = 0; rc = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)code) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - code) << rcshift);
        if (++run < k) continue;
        if (!canonical) { stats_emit(s, fwd, 0, shift); continue; }
        int flag = tie_rc ? !(fwd < rc) : (fwd > rc);
        stats_emit(s, flag ? rc : fwd, flag, shift);
    }
    return 0;
}

/* ======================================================================================== *
 *  synthetic inputs
 * ======================================================================================== */

uint64_t ntko_splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void ntko_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                      uint32_t n_per_1024, uint8_t *out)
{
    static const char L[4] = { 'A', 'C', 'G', 'T' };
    uint64_t wpr = (read_len + 31) / 32, npr = (read_len + 5) / 6;
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t r = first_read + i;
        uint8_t *o = out + i * ((uint64_t)read_len + 1);
        uint64_t w = 0, m = 0;
        for (uint32_t j = 0; j < read_len; j++) {
            if (j % 32 == 0) w = ntko_splitmix64_at(seed, r * wpr + j / 32);
            uint8_t b = (uint8_t)L[(w >> (2 * (j % 32))) & 3];
            if (n_per_1024) {
                if (j % 6 == 0) m = ntko_splitmix64_at(seed + 1, r * npr + j / 6);
                if (((m >> (10 * (j % 6))) & 1023) < n_per_1024) b = 'N';
            }
            o[j] = b;
        }
        o[read_len] = '\n';
    }
}

/* ======================================================================================== *
 *  windowed minimizers (BASELINE.json configs[4]: "minimizers (w=11,k=21)")
 * ======================================================================================== */

/* The reference has no windowed-minimizer iterator; the nearest defined semantics is
 * sequence::minimizer(window, k) (src/sequence.rs:139-152) applied to every window of w+k-1 good bases of a
 * normalised sequence.  For each such window this calls the restated minimizer literally and folds the 2-bit
 * value of the returned k bytes into the statistic; the flag is the strand flag CanonicalKmers (tie_rc = 1) or
 * BitNuclKmer (tie_rc = 0) reports for the leftmost k-mer of the window whose canonical form is the minimizer. */
int ntko_minimizers_reduce(ntko_stats *s, const uint8_t *buf, size_t n, uint8_t k, uint32_t w, int accept_u, int tie_rc)
{
    if (k < 1 || k > 32 || w < 1) return -1;
    const size_t span = (size_t)w + k - 1;
    unsigned shift = hist_shift(k);
    uint8_t *win = (uint8_t *)malloc(span), *mn = (uint8_t *)malloc(k);
    size_t run = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = buf[i];
        int good = ntko_is_good_base(c) || (accept_u && (c == 'U' || c == 'u'));
        run = good ? run + 1 : 0;
        if (run < span) continue;
        /* normalise the window bytes as sequence::normalize would (upper-case, U -> T) */
        for (size_t t = 0; t < span; t++) {
            uint8_t b = buf[i + 1 - span + t];
            b = (uint8_t)(b & 0xDF);
            win[t] = b == 'U' ? (uint8_t)'T' : b;
        }
        ntko_minimizer(win, span, k, mn);
        uint64_t v = ntko_bytes_to_bitmer(mn, k).seq;
        /* strand of the leftmost window position whose canonical k-mer equals the minimizer */
        int flag = 0;
        for (size_t t = 0; t + k <= span; t++) {
            uint64_t f = ntko_bytes_to_bitmer(win + t, k).seq;
            ntko_bitkmer fk = { f, k };
            uint64_t r = ntko_bit_reverse_complement(fk).seq;
            if (f == v || r == v) { flag = tie_rc ? !(f < r) : (f > r); break; }  /* that k-mer's own iterator flag */
        }
        stats_emit(s, v, flag, shift);
    }
    free(win); free(mn);
    return 0;
}
