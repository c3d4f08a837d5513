/*
 * ntk_oracle.h — CPU restatement of needletail's per-sequence k-mer hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped library
 * (needletail_amd/csrc -> libneedletail_amd.so) never links, loads or calls anything here.
 *
 * The reference crate (onecodex/needletail v0.7.3, Rust) cannot be compiled in this image
 * (no rustc/cargo, un-vendored deps), so this file restates its algorithm in plain C,
 * function by function, following the cited lines.  The restatement is pinned by the
 * reference's own known-answer vectors (tests/test_oracle_golden.py): the unit-test literals in
 * src/sequence.rs:311-375, src/kmer.rs:132-227, src/bitkmer.rs:188-297, test_python.py:36-41,
 * 101-149, and the whole-file counts asserted in benches/benchmark.rs:43-44,66-67
 * (718 007 / 350 983 at k=31 on tests/data/28S.fasta, both code paths).
 *
 * All "reference file:line" citations are relative to /root/reference/.
 */
#ifndef NTK_ORACLE_H
#define NTK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/sequence.rs ------------------------------------------------------------------ */

/* sequence::normalize (src/sequence.rs:19-62).  `out` must hold n bytes.  Returns the output
 * length; *changed is the reference's `changed` flag (None <=> *changed == 0). */
size_t ntko_normalize(const uint8_t *seq, size_t n, int allow_iupac, uint8_t *out, int *changed);

/* Sequence::strip_returns (src/sequence.rs:165-191).  Returns output length; *borrowed = 1 when the
 * reference would return Cow::Borrowed (no CR/LF present). */
size_t ntko_strip_returns(const uint8_t *seq, size_t n, uint8_t *out, int *borrowed);

/* sequence::complement (src/sequence.rs:68-105). */
uint8_t ntko_complement(uint8_t n);

/* Sequence::reverse_complement (src/sequence.rs:202-208). `out` holds n bytes. */
void ntko_reverse_complement(const uint8_t *seq, size_t n, uint8_t *out);

/* sequence::canonical (src/sequence.rs:110-134): writes the canonical form (n bytes) to out,
 * returns 1 when the reverse complement was chosen. */
int ntko_canonical(const uint8_t *seq, size_t n, uint8_t *out);

/* sequence::minimizer (src/sequence.rs:139-152): writes `length` bytes to out. n >= length. */
void ntko_minimizer(const uint8_t *seq, size_t n, size_t length, uint8_t *out);

/* QualitySequence::quality_mask (src/sequence.rs:285-296): out[i] = qual[i] < score ? 'N' : seq[i],
 * over min(n_seq, n_qual) items (zip). Returns the output length. */
size_t ntko_quality_mask(const uint8_t *seq, size_t n_seq, const uint8_t *qual, size_t n_qual,
                         uint8_t score, uint8_t *out);

/* ---- src/kmer.rs ---------------------------------------------------------------------- */

int ntko_is_good_base(uint8_t chr); /* src/kmer.rs:6-8 */

/* Kmers (src/kmer.rs:13-41): number of windows; window i is seq[i..i+k]. */
size_t ntko_kmers_count(size_t n, uint8_t k);

/* CanonicalKmers (src/kmer.rs:48-130), literal iterator. */
typedef struct {
    uint8_t k;
    size_t start_pos;
    const uint8_t *buffer;
    size_t len;
    const uint8_t *rc_buffer;
    size_t rc_len;
} ntko_canonical_kmers;

void ntko_ck_new(ntko_canonical_kmers *it, const uint8_t *buffer, size_t len,
                 const uint8_t *rc_buffer, size_t rc_len, uint8_t k);
/* Returns 1 and fills (pos, slice, is_rc) or 0 at the end.  slice points into buffer or
 * rc_buffer exactly as the reference's borrowed slice does. */
int ntko_ck_next(ntko_canonical_kmers *it, size_t *pos, const uint8_t **slice, int *is_rc);

/* ---- src/bitkmer.rs ------------------------------------------------------------------- */

typedef struct {
    uint64_t seq; /* BitKmerSeq */
    uint8_t k;
} ntko_bitkmer; /* BitKmer = (u64, u8), src/bitkmer.rs:2-3 */

/* NUC2BIT_LOOKUP (src/bitkmer.rs:5-18): returns 0..3 or -1. */
int ntko_nuc2bit(uint8_t nuc);
/* extend_kmer (src/bitkmer.rs:26-36): 1 on success. */
int ntko_extend_kmer(ntko_bitkmer *kmer, uint8_t new_char);

/* BitNuclKmer (src/bitkmer.rs:72-109), literal iterator. */
typedef struct {
    size_t start_pos;
    ntko_bitkmer cur_kmer;
    const uint8_t *buffer;
    size_t len;
    int canonical;
} ntko_bit_nucl_kmer;

void ntko_bnk_new(ntko_bit_nucl_kmer *it, const uint8_t *slice, size_t len, uint8_t k, int canonical);
int ntko_bnk_next(ntko_bit_nucl_kmer *it, size_t *pos, ntko_bitkmer *kmer, int *was_rc);

ntko_bitkmer ntko_bit_reverse_complement(ntko_bitkmer kmer);   /* src/bitkmer.rs:112-132 */
ntko_bitkmer ntko_bit_canonical(ntko_bitkmer kmer, int *was_rc); /* src/bitkmer.rs:136-143 */
ntko_bitkmer ntko_bit_minimizer(ntko_bitkmer kmer, uint8_t minmer_size); /* src/bitkmer.rs:146-162 */
void ntko_bitmer_to_bytes(ntko_bitkmer kmer, uint8_t *out);    /* src/bitkmer.rs:164-186, k bytes */
ntko_bitkmer ntko_bytes_to_bitmer(const uint8_t *kmer, uint8_t k); /* test helper, src/bitkmer.rs:288-296 */

/* ---- bulk drivers over the literal iterators (what user code does per record) --------- */

/* Walk CanonicalKmers to the end; writes up to cap items. Returns the total item count. */
size_t ntko_canonical_kmers_all(const uint8_t *buffer, size_t len, const uint8_t *rc, size_t rc_len,
                                uint8_t k, uint64_t *pos_out, uint8_t *is_rc_out, size_t cap);
/* Walk BitNuclKmer to the end. */
size_t ntko_bit_kmers_all(const uint8_t *slice, size_t len, uint8_t k, int canonical,
                          uint64_t *pos_out, uint64_t *val_out, uint8_t *was_rc_out, size_t cap);

/* ---- the reduced statistic compared bit-exactly with the GPU ----------- */

#define NTKO_HIST_MAX_P 6
#define NTKO_HIST_BINS 4096 /* 4^6 */

enum { NTKO_PATH_BYTES_CANONICAL = 0, NTKO_PATH_BITS = 1, NTKO_PATH_BITS_CANONICAL = 2 };
enum { NTKO_PRE_NONE = 0, NTKO_PRE_STRIP_RETURNS = 1, NTKO_PRE_NORMALIZE = 2, NTKO_PRE_NORMALIZE_IUPAC = 3 };

typedef struct {
    uint64_t n_total;  /* emitted k-mers */
    uint64_t n_fwd;    /* items with flag == false (benches/benchmark.rs:37-39 "n_canonical") */
    uint64_t n_rc;     /* items with flag == true */
    uint64_t sum;      /* sum of emitted 2-bit values mod 2^64 */
    uint64_t xr;       /* xor of emitted 2-bit values */
    uint64_t hist[NTKO_HIST_BINS]; /* bin = value >> 2*(k-p), p = min(k, 6) */
} ntko_stats;

void ntko_stats_clear(ntko_stats *s);
void ntko_stats_merge(ntko_stats *dst, const ntko_stats *src);

/* One record through the reference's documented per-record chain, literally (with the same
 * heap allocations the reference makes):
 *   BYTES_CANONICAL: pre-step -> reverse_complement -> CanonicalKmers   (src/lib.rs:22-31,
 *                    benches/benchmark.rs:32-41)
 *   BITS[_CANONICAL]: pre-step -> BitNuclKmer                           (benches/benchmark.rs:55-64)
 * k must be 1..32 (values are folded into a u64).  Returns 0, or -1 on bad arguments. */
int ntko_reduce_record(ntko_stats *s, const uint8_t *seq, size_t n, uint8_t k, int path, int pre);

/* The reference benchmark's own loop (benches/benchmark.rs:32-41,55-64): iterate and count items and
 * `!was_rc` items only.  This is what bench.py's cpu_baseline times. */
int ntko_count_record(uint64_t *n_total, uint64_t *n_fwd, const uint8_t *seq, size_t n, uint8_t k, int path, int pre);
int ntko_count_batch_mt(uint64_t *n_total, uint64_t *n_fwd, const uint8_t *seq, const uint64_t *offsets,
                        size_t n_records, size_t gap, uint8_t k, int path, int pre, int n_threads);

/* Records concatenated in `seq`; record r = seq[offsets[r] .. offsets[r+1] - gap).  `gap` is the
 * number of separator bytes after every record (1 in the device batch layout, 0 for none). */
int ntko_reduce_batch(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                      size_t gap, uint8_t k, int path, int pre);
/* Same, records partitioned statically over n_threads pthreads. */
int ntko_reduce_batch_mt(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                         size_t gap, uint8_t k, int path, int pre, int n_threads);
/* reuse_buffers != 0: each thread keeps its two scratch buffers across records (NOT the reference's behaviour: informational) */
int ntko_reduce_batch_mt2(ntko_stats *s, const uint8_t *seq, const uint64_t *offsets, size_t n_records,
                          size_t gap, uint8_t k, int path, int pre, int n_threads, int reuse_buffers);

/* Independent second formulation ("run length of good bases >= k") over a
 * whole separator-delimited buffer in one pass: every byte that is not a base is a break.
 * accept_u = 1 gives the normalize pipeline's alphabet (U/u -> T), 0 the bit path's.
 * tie_rc = 1: fwd == rc reports flag true (byte path); 0: false (bit path).
 * canonical = 0: forward value, flag false.  Used to cross-check the literal iterators and to
 * check large GPU runs quickly. */
int ntko_reduce_fused(ntko_stats *s, const uint8_t *buf, size_t n, uint8_t k, int canonical,
                      int tie_rc, int accept_u);

/* Windowed minimizers: sequence::minimizer(window, k) (src/sequence.rs:139-152) for every window of w+k-1 good
 * bases; value folded like the k-mer statistic, flag = the iterator flag of the leftmost minimal k-mer. */
int ntko_minimizers_reduce(ntko_stats *s, const uint8_t *buf, size_t n, uint8_t k, uint32_t w, int accept_u, int tie_rc);

/* ---- deterministic synthetic inputs, SplitMix64, counter-based ---------- */

uint64_t ntko_splitmix64_at(uint64_t seed, uint64_t index);
/* Reads first_read .. first_read+n_reads-1 of the synthetic set `seed`: each read is read_len
 * bases followed by one '\n' separator; out holds n_reads*(read_len+1) bytes.
 * Base j of read r: 2 bits of SplitMix64(seed)[r*words_per_read + j/32]; it becomes 'N' when the
 * 10-bit field j%6 of SplitMix64(seed+1)[r*nwords_per_read + j/6] is < n_per_1024. */
void ntko_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                      uint32_t n_per_1024, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
