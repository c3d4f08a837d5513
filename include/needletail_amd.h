/*
 * needletail_amd.h — C ABI of the MI355X-native k-mer extraction engine.
 *
 * Drop-in boundary for ONE hot path of onecodex/needletail (v0.7.3): the per-sequence chain
 *     normalize -> reverse_complement -> canonical_kmers      (byte path)
 *     strip_returns -> bit_kmers / BitNuclKmer                (2-bit path)
 * behind the reference's `Sequence` trait (reference src/sequence.rs:156-253).  Plain pointers and
 * sizes only; no C++/torch types.  Every entry point returns an `int` status (NTK_OK == 0); nothing
 * throws or aborts across this boundary (the reference panics on misuse, src/kmer.rs:91,
 * src/bitkmer.rs:51 — here misuse is NTK_ERR_BAD_K / NTK_ERR_BAD_ARG).
 *
 * There is no CPU fallback anywhere behind this header: every function that computes runs HIP
 * kernels on a gfx950 device and fails with NTK_ERR_NO_DEVICE / NTK_ERR_HIP when it cannot.
 *
 * Two faces:
 *   1. batch face   — the fast path.  Whole record batches (concatenated sequence bytes, one break
 *                     byte between records) are scanned by one kernel launch; results are either
 *                     REDUCED on device (counters + prefix histogram + digests, ntk_result) or
 *                     MATERIALISED densely (one u64 per window + valid / is_rc bit planes).
 *   2. compat face  — per-sequence functions with the reference's shapes (eager, synchronous),
 *                     returning exactly what the reference's iterators yield.
 *
 * A Rust maintainer binds these with an `extern "C"` block (INTEGRATION.md shows it) and keeps
 * `impl Sequence` unchanged on top.
 */
#ifndef NEEDLETAIL_AMD_H
#define NEEDLETAIL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): ntk_result.n_undigested (k = 33..255 on the byte path's reduce face), ntk_gunzip_info grew (streamed runs), + ntk_ctx_get_option,
 * ntk_scan_file_info; the environment variables of earlier rounds are ntk_ctx_set_option options.
 * 3 (round 5): + ntk_ctx_set_option (the library no longer reads A/B switches from the environment), ntk_gunzip / ntk_gunzip_free,
 * ntk_bit_kmers_batch_planes.  2 (round 4): + ntk_canonical_kmers_batch_planes, ntk_ctx_trim, ntk_comm_allreduce_time_ms; round 3 had added
 * ntk_device_count, ntk_pinned_alloc / ntk_pinned_free under version 1.  ntk_abi_version() of the loaded library says what it exports. */
#define NTK_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------------------------- */
enum {
    NTK_OK = 0,
    NTK_ERR_BAD_K = 1,       /* k == 0, or k beyond what the entry point supports               */
    NTK_ERR_BAD_ARG = 2,     /* null pointer, misaligned device pointer, bad enum value ...     */
    NTK_ERR_HIP = 3,         /* a HIP runtime call failed; ntk_last_hip_error() has the code    */
    NTK_ERR_NO_DEVICE = 4,   /* no usable gfx950 device                                         */
    NTK_ERR_CAPACITY = 5,    /* caller-provided output / batch capacity too small               */
    NTK_ERR_UNSUPPORTED = 6, /* combination not available on the device path (never falls back) */
    NTK_ERR_NOMEM = 7,
    NTK_ERR_PARSE = 8,       /* FASTA/FASTQ parse error; ntk_reader_error() has kind / line / id / message */
    NTK_ERR_RCCL = 9,        /* librccl could not be loaded or a collective failed; ntk_last_rccl_error() has the ncclResult_t */
    NTK_EOF = 100            /* ntk_reader_next: no more records (not an error)                 */
};
const char *ntk_strerror(int status);
int ntk_last_hip_error(void);
int ntk_last_rccl_error(void);   /* ncclResult_t of the last failed RCCL call on this thread; -1 = librccl not loadable */
int ntk_abi_version(void);
/* Usable gfx950 devices (0 and NTK_OK when there is none: a count, not an error); ntk_ctx_create accepts 0 .. n-1.
 * Multi-GPU callers size ntk_comm_init_all / their rank-per-GPU launch with it. */
int ntk_device_count(int *n);

/* ---- parameters (mirror the arguments of the reference's trait methods) --------------------- */
enum { /* which iterator: reference src/sequence.rs:237-252 */
    NTK_PATH_BYTES_CANONICAL = 0, /* Sequence::canonical_kmers(k, &rc): tie (fwd == rc) reports is_rc = true  (src/kmer.rs:124-128) */
    NTK_PATH_BITS = 1,            /* Sequence::bit_kmers(k, false): forward value, flag false (src/bitkmer.rs:105-107)   */
    NTK_PATH_BITS_CANONICAL = 2   /* Sequence::bit_kmers(k, true): tie reports was_rc = false (src/bitkmer.rs:136-143)   */
};
enum { /* which pre-step the record went through: fixes the alphabet */
    NTK_PRE_NONE = 0,            /* raw slice: bases = acgtACGT, every other byte breaks the window         */
    NTK_PRE_STRIP_RETURNS = 1,   /* Sequence::strip_returns (src/sequence.rs:165-191): CR/LF deleted        */
    NTK_PRE_NORMALIZE = 2,       /* Sequence::normalize(false) (src/sequence.rs:19-62,226-232): U/u -> T,   */
    NTK_PRE_NORMALIZE_IUPAC = 3  /*   whitespace deleted; _IUPAC = normalize(true) (same k-mer stream)      */
};

typedef struct ntk_params {
    uint32_t k;    /* 1..32 on the batch face (values are packed into a u64)          */
    uint32_t path; /* NTK_PATH_*                                                       */
    uint32_t pre;  /* NTK_PRE_*                                                        */
    uint32_t flags;/* bits 7:0 = minimizer window w (k-mers per window; 0 = plain k-mers):  */
                   /* reduce entry points then fold windowed minimizers instead of k-mers; */
                   /* bits 15:8 = quality cutoff (0 = none): bases whose quality byte is below it  */
                   /* are masked as QualitySequence::quality_mask does (src/sequence.rs:285-296)   */
                   /* on the *_quality entry points and on batches that carry qualities;           */
                   /* bit 16 = NTK_FLAG_RESET (reduce entry points: zero the accumulators first, as      */
                   /* ntk_accum_reset would, but inside this call's own kernel launch - a pass that      */
                   /* starts a new result saves the separate memset); other bits reserved, must be 0     */
} ntk_params;
#define NTK_FLAGS(window_w, quality_cutoff) (((uint32_t)(window_w) & 0xFFu) | (((uint32_t)(quality_cutoff) & 0xFFu) << 8))
#define NTK_FLAG_RESET (1u << 16)

/* Reduced result of a scan.  `value` is the emitted k-mer in the reference's
 * 2-bit encoding (A0 C1 G2 T3, first base most significant; src/bitkmer.rs:5-36). */
#define NTK_HIST_MAX_PREFIX 6
#define NTK_HIST_BINS 4096
typedef struct ntk_result {
    uint64_t n_total; /* emitted k-mers                                                         */
    uint64_t n_fwd;   /* items whose flag is false ("n_canonical" in benches/benchmark.rs:37-39) */
    uint64_t n_rc;    /* items whose flag is true                                               */
    uint64_t sum;     /* sum of values mod 2^64                                                 */
    uint64_t xr;      /* xor of values                                                          */
    uint64_t hist[NTK_HIST_BINS]; /* bin = value >> 2*(k-p), p = min(k,6): leading p bases       */
    uint64_t n_undigested; /* of n_total: k-mers of a k > 32 scan (byte path, reduce face; CanonicalKmers takes k: u8,
                              reference src/kmer.rs:48-82) - counted and binned, but a 2-bit value of more than 64 bits
                              has no place in sum / xr, which cover the other n_total - n_undigested k-mers (ABI 4)  */
} ntk_result;
/* Device-side accumulator layout (u64 words), for callers that reduce across GPUs themselves: */
#define NTK_ACC_N_TOTAL 0
#define NTK_ACC_N_FWD 1
#define NTK_ACC_N_RC 2
#define NTK_ACC_SUM 3
#define NTK_ACC_XOR 4   /* NOT summable: combine with xor (or use the bit counters below) */
#define NTK_ACC_UNDIGESTED 5 /* ntk_result.n_undigested */
#define NTK_ACC_REDONE 6 /* diagnostic, not part of ntk_result: speculative launches since the reset (un-normalised byte-path input,
                            k > 32) whose result came from the byte-walking kernel queued behind them - lower case, or two k-mers
                            equal over 32 bases.  Summable; read through ntk_accum_device_ptr / a bound buffer                   */
#define NTK_ACC_HIST 8  /* 4096 words follow */
#define NTK_ACC_XOR_BITS (8 + NTK_HIST_BINS) /* 64 words: how many folded partial xors had bit i set; summable
                                                across GPUs with one ncclSum all-reduce, xor bit i = parity */
#define NTK_ACC_WORDS (8 + NTK_HIST_BINS + 64)

/* ---- context ---------------------------------------------------------------------------------
 * One ctx per (thread, device); a ctx is not thread-safe, independent ctxs are.  The ctx owns the
 * device accumulators, per-block partials, scratch and (unless borrowed) its HIP stream. */
typedef struct ntk_ctx ntk_ctx;
int ntk_ctx_create(int device, ntk_ctx **out);
/* Borrow the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream. */
int ntk_ctx_create_on_stream(int device, void *hip_stream, ntk_ctx **out);
void ntk_ctx_destroy(ntk_ctx *ctx);
int ntk_ctx_synchronize(ntk_ctx *ctx);
/* Launch geometry of the scan kernel: blocks (0 = auto: the resident grid) x threads per block (a multiple of 64 up to
 * 1024; 0 = auto: the largest of 768 / 640 / 512 that keeps two blocks of a reduce build resident per CU - 768 for every shipped
 * build; materialise mode always runs 256-thread blocks). */
int ntk_ctx_set_launch(ntk_ctx *ctx, int blocks, int threads_per_block);
/* Test / A-B support, per ctx (the library's dispatch reads no environment variable): value 0 restores an option's default.
 *   NTK_OPT_COMPAT_CHUNK_BYTES     bytes per chunk of the batched compat faces (default 16 MiB; the suites force hundreds of chunks
 *                                  out of small batches with it; values below 64 are taken as 64)
 *   NTK_OPT_MINIMIZER_CHUNK_BYTES  input bytes per pass of the two-pass minimizer route (default 256 MiB; a multiple of 4096)
 *   NTK_OPT_MINIMIZER_ROUTE        NTK_ROUTE_NO_* bits: routes of ntk_minimizers_reduce_device switched OFF - the register-fused
 *                                  builds, the generic fused kernel (both off = materialise + window-min), the v_min_f64 keys of the
 *                                  generic kernel (k <= 25 then runs the general keys) - so that each route can be checked against
 *                                  the others on the same input; NTK_ROUTE_NO_SPECULATION: un-normalised byte-path input
 *                                  (NTK_PATH_BYTES_CANONICAL, pre < NORMALIZE) goes straight to the raw-byte kernel instead of the
 *                                  packed-value scan that watches for lower case and is redone by that kernel only if it saw any;
 *                                  k = 33..255 likewise goes straight to the byte-walking kernel instead of the packed-stream one
 *                                  that decides the strand on 32 bases and is redone only for an inverted repeat or lower case
 *   NTK_OPT_COMPAT_PACK_THREADS    host threads that pack a chunk of the item-array compat faces (default 8)
 *   NTK_OPT_COPY_STREAMS           HIP streams that take the pinned batches' H2D copies in turn (1 or 2; default 2: the next batch's copy is
 *                                  queued while one runs - measured +14 % on the H2D-inclusive FASTQ pipeline with 4 MiB batches)
 *   NTK_OPT_BATCH_WAIT_POLL_US     ntk_batch_wait polls the batch's event every that many microseconds (default 50; NTK_POLL_BLOCK = block
 *                                  in hipEventSynchronize instead; values are clamped to 10 s)
 *   NTK_OPT_GZ_INMEM_LIMIT_BYTES   largest inflated size ntk_scan_file_parallel accepts for a gzip file: with n_threads = 1 the text is held in
 *                                  memory (default 16 GiB); with more threads it is the address range reserved for the text, which is consumed
 *                                  while it is inflated and never resident as a whole (default 4 TiB)
 *   NTK_OPT_GZ_STREAM_WINDOW_BYTES how far the inflater of ntk_scan_file_parallel may run ahead of the parsers (default 512 MiB, at least 8 MiB):
 *                                  the bound on the inflated text resident at any time
 *   NTK_OPT_PIPE_STATS             != 0: the parser threads of the parallel producer print their phase times on stderr
 * ntk_ctx_get_option reads an option back (0 = the default is in force). */
enum { NTK_OPT_COMPAT_CHUNK_BYTES = 1, NTK_OPT_MINIMIZER_CHUNK_BYTES = 2, NTK_OPT_MINIMIZER_ROUTE = 3, NTK_OPT_COMPAT_PACK_THREADS = 4,
       NTK_OPT_COPY_STREAMS = 5, NTK_OPT_BATCH_WAIT_POLL_US = 6, NTK_OPT_GZ_INMEM_LIMIT_BYTES = 7, NTK_OPT_GZ_STREAM_WINDOW_BYTES = 8,
       NTK_OPT_PIPE_STATS = 9 };
#define NTK_POLL_BLOCK 0xFFFFFFFFu
#define NTK_ROUTE_NO_REGFUSED 1u
#define NTK_ROUTE_NO_GENERIC 2u
#define NTK_ROUTE_NO_F64 4u
#define NTK_ROUTE_NO_SPECULATION 8u
int ntk_ctx_set_option(ntk_ctx *ctx, int option, uint64_t value);
int ntk_ctx_get_option(ntk_ctx *ctx, int option, uint64_t *value);
/* Record hipEvents around every scan-kernel launch; ntk_ctx_scan_time_ms returns the sum of the
 * scan kernels' durations since the last call and how many launches that covers (synchronises). */
int ntk_ctx_enable_timing(ntk_ctx *ctx, int on);
int ntk_ctx_scan_time_ms(ntk_ctx *ctx, double *total_ms, uint64_t *launches);

/* ---- multi-GPU: ONE RCCL sum all-reduce of the accumulators -------------------
 * Record batches shard across the GPUs of a node with no other exchange (the reference has no counterpart: it is a
 * single-threaded CPU library, Cargo.toml:28-54; a k-mer never spans two records, reference src/sequence.rs:237-252, so the
 * reduced result is a plain sum over shards).  Every ctx accumulates its own shard; ntk_allreduce_accumulators then runs
 *     ncclAllReduce(acc, acc, NTK_ACC_WORDS, ncclUint64, ncclSum)
 * on each ctx's stream (the xor digest travels as 64 one-bit counters, so the one sum carries everything) and rebuilds the
 * xor word.  librccl is loaded on first use (the copy already in the process, else librccl.so.1). */
typedef struct ntk_comm ntk_comm;
#define NTK_COMM_ID_BYTES 128
/* one process drives n GPUs (ncclCommInitAll): ctxs[i] are contexts on n distinct devices */
int ntk_comm_init_all(ntk_ctx *const *ctxs, int n, ntk_comm **out);
/* one process per GPU (ncclCommInitRank): rank 0 makes the id, ships its 128 bytes to the other ranks by any means */
int ntk_comm_unique_id(uint8_t id[NTK_COMM_ID_BYTES]);
int ntk_comm_init_rank(ntk_ctx *ctx, int n_ranks, int rank, const uint8_t id[NTK_COMM_ID_BYTES], ntk_comm **out);
int ntk_comm_size(const ntk_comm *comm);     /* ranks in the communicator */
/* All ranks (all local ctxs of an init_all communicator): accumulators <- sum over ranks, in place, asynchronous on the
 * ctx streams; ntk_accum_read / ntk_ctx_synchronize order after it.  Call it once, after the last batch of the run. */
int ntk_allreduce_accumulators(ntk_comm *comm);
/* Durations of the all-reduces issued while the (first local) ctx had ntk_ctx_enable_timing on, summed since the last call: what a
 * multi-GPU run pays for the collective on this rank, next to ntk_ctx_scan_time_ms (the scaling bench prints both per rank). */
int ntk_comm_allreduce_time_ms(ntk_comm *comm, double *total_ms, uint64_t *calls);
void ntk_comm_destroy(ntk_comm *comm);

/* ---- batch face, reduce mode ----------------------------------------------------------------
 * Device batch layout: the records' sequence bytes back to back, each followed by ONE break byte
 * (any non-base byte; the packer writes '\n'), no bytes of the pre-step's "deleted" class inside a
 * record (the packer, ntk_batch_append, removed them).  d_seq must be 16-byte aligned and readable up to
 * round_up(n_bytes, 16).
 * Replaces, per record: seq.normalize(..) / strip_returns(), seq.reverse_complement(),
 * seq.canonical_kmers(k,&rc) or seq.bit_kmers(k,canonical) and the user's counting loop
 * (reference src/lib.rs:22-31, benches/benchmark.rs:32-41,55-64).
 * NTK_PATH_BYTES_CANONICAL with pre = NONE / STRIP_RETURNS (bytes the caller did not normalise): the reference iterator compares RAW
 * bytes (src/kmer.rs:121-128; lower case sorts above upper case), so reduce mode runs a raw-byte kernel for it (slower than the
 * packed-value scan, exact on mixed case); dense values (ntk_materialize_device), quality masking and windowed minimizers on such
 * input are NTK_ERR_UNSUPPORTED - normalize first, as the reference's documented chain does. */
int ntk_accum_reset(ntk_ctx *ctx);
int ntk_reduce_device(ntk_ctx *ctx, const uint8_t *d_seq, uint64_t n_bytes, const ntk_params *p); /* async */
/* Quality masking fused into the scan: `(seq, qual).quality_mask(cutoff)` (reference
 * src/sequence.rs:285-296: a base whose RAW quality byte is < cutoff becomes N) followed by the chain above, in one pass.
 * d_qual has the layout and alignment of d_seq (one quality byte per sequence byte; the byte under a record's break byte
 * is ignored).  cutoff = bits 15:8 of p->flags; d_qual NULL or cutoff 0 = ntk_reduce_device. */
int ntk_reduce_device_quality(ntk_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_bytes,
                              const ntk_params *p);                                      /* async */
int ntk_accum_read(ntk_ctx *ctx, ntk_result *out);            /* synchronises the ctx stream        */
int ntk_accum_device_ptr(ntk_ctx *ctx, uint64_t **d_words);  /* NTK_ACC_WORDS u64 words on device  */
/* Accumulate into caller-owned device memory (NTK_ACC_WORDS u64, e.g. a torch tensor that is then
 * all-reduced over RCCL); NULL re-binds the ctx's own buffer.  The caller zeroes / resets it via
 * ntk_accum_reset as usual. */
int ntk_accum_bind_device(ntk_ctx *ctx, uint64_t *d_words);

/* ---- batch face, materialise mode -------------------------------------------------------------
 * Dense outputs indexed by the window's END byte e (window = d_seq[e-k+1 .. e]):
 *   d_values[e]                     emitted value (canonical or forward), undefined where invalid
 *   d_valid16[e/16] bit (15 - e%16) window is emitted
 *   d_rc16[e/16]    bit (15 - e%16) flag (is_rc / was_rc)
 * The reference's `pos` is e - (k-1) - record_start.  d_values holds round_up(n_bytes,16) u64, d_valid16 and
 * d_rc16 hold ceil(n_bytes/16) u16 each.  d_values may be NULL (flags only). */
int ntk_materialize_device(ntk_ctx *ctx, const uint8_t *d_seq, uint64_t n_bytes, const ntk_params *p,
                           uint64_t *d_values, uint16_t *d_valid16, uint16_t *d_rc16);   /* async */
int ntk_materialize_device_quality(ntk_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_bytes,
                                   const ntk_params *p, uint64_t *d_values, uint16_t *d_valid16, uint16_t *d_rc16);

/* ---- batch face, pinned host batches (CPU parser fills, H2D copy overlaps the kernels) ---------
 * A batch exposes PINNED host memory; the caller (FastxReader loop) appends records between
 * acquire and submit; submit enqueues hipMemcpyAsync + the scan on the ctx stream and returns at
 * once; the batch may be re-acquired after ntk_batch_wait.  Results accumulate in the ctx. */
typedef struct ntk_batch ntk_batch;
int ntk_batch_acquire(ntk_ctx *ctx, uint64_t max_bytes, uint64_t max_records, ntk_batch **out);
/* Copies one record's raw sequence (`SequenceRecord::sequence()`, reference src/parser/record.rs:181-185)
 * applying the DELETE part of `pre` (NONE: nothing; STRIP_RETURNS: CR/LF; NORMALIZE*: space, tab, CR, LF),
 * then one break byte.  NTK_ERR_CAPACITY when the batch is full (submit it and acquire another). */
int ntk_batch_append(ntk_batch *b, const uint8_t *seq, uint64_t n, uint32_t pre);
/* The same with the record's quality line (`SequenceRecord::qual()`, n bytes) for masking at `cutoff` (1..255): the batch
 * then carries a parallel pinned quality stream, copied and applied by ntk_batch_submit, whose p->flags must hold the
 * same cutoff (NTK_ERR_BAD_ARG otherwise; one cutoff per fill).  A byte of the deleted class is dropped with its quality
 * byte unless that quality is below the cutoff - the reference masks before it normalizes, so such a byte is an N by
 * then.  Records appended without qualities are never masked. */
int ntk_batch_append_quality(ntk_batch *b, const uint8_t *seq, const uint8_t *qual, uint64_t n, uint32_t pre,
                             uint32_t cutoff);
int ntk_batch_buffers(ntk_batch *b, uint8_t **seq, uint64_t **offsets, uint64_t *n_bytes, uint64_t *n_records);
int ntk_batch_submit(ntk_ctx *ctx, ntk_batch *b, const ntk_params *p);  /* async: H2D + reduce */
/* Waits by polling the batch's event every 50 microseconds (ntk_ctx_set_option NTK_OPT_BATCH_WAIT_POLL_US; NTK_POLL_BLOCK = block in
 * hipEventSynchronize). */
int ntk_batch_wait(ntk_ctx *ctx, ntk_batch *b);
void ntk_batch_release(ntk_ctx *ctx, ntk_batch *b);

/* ---- CPU producer: FastxReader-shaped record reader + the whole-file pipeline -------------------
 * The FASTX parser stays on the CPU (BASELINE.json north_star).  ntk_reader_* mirrors parse_fastx_file /
 * parse_fastx_reader and FastxReader::next (reference src/parser/mod.rs:85-163, src/parser/utils.rs:119-130):
 * plain or gzip input, FASTA or FASTQ by first byte, one BORROWED record at a time (pointers are valid until the
 * next call), the reference's error kinds.  ntk_scan_reader is the README program (reference src/lib.rs:15-35)
 * on the GPU: read records, pack them into pinned batches (deleting the pre-step's whitespace class), overlap the
 * H2D copies with the scan kernels, accumulate into the ctx. */
typedef struct ntk_reader ntk_reader;
typedef struct ntk_record {
    const uint8_t *id;   uint64_t id_len;    /* SequenceRecord::id()       reference src/parser/record.rs:68-73  */
    const uint8_t *seq;  uint64_t seq_len;   /* SequenceRecord::raw_seq()  reference src/parser/record.rs:78-83  */
    const uint8_t *qual; uint64_t qual_len;  /* SequenceRecord::qual(); NULL for FASTA                          */
    uint32_t format;                         /* 0 = FASTA, 1 = FASTQ                                            */
    uint32_t line_ending;                    /* SequenceRecord::line_ending(): 1 = Unix, 2 = Windows  record.rs:150-153 */
    uint64_t line;                           /* SequenceRecord::start_line_number()                            */
    uint64_t num_bases;                      /* SequenceRecord::num_bases()                                    */
    uint64_t byte;                           /* SequenceRecord::position().byte()  reference src/parser/record.rs:147-149 */
} ntk_record;
enum { /* ParseErrorKind, reference src/errors.rs:26-44 */
    NTK_PARSE_IO = 1, NTK_PARSE_UNKNOWN_FORMAT = 2, NTK_PARSE_INVALID_START = 3, NTK_PARSE_INVALID_SEPARATOR = 4,
    NTK_PARSE_UNEQUAL_LENGTHS = 5, NTK_PARSE_UNEXPECTED_END = 6, NTK_PARSE_EMPTY_FILE = 7
};
/* On NTK_ERR_PARSE *out is still a valid handle (query ntk_reader_error, then close it). */
int ntk_reader_open_file(const char *path, ntk_reader **out);   /* "-" reads standard input (parse_fastx_stdin) */
int ntk_reader_open_memory(const uint8_t *data, uint64_t n, ntk_reader **out); /* data must outlive the reader */
int ntk_reader_next(ntk_reader *r, ntk_record *rec);  /* NTK_OK, NTK_EOF or NTK_ERR_PARSE */
int ntk_reader_error(ntk_reader *r, int *kind, uint64_t *line, char *msg, uint64_t msg_cap, char *id, uint64_t id_cap);
/* FastxReader::position / line_ending (reference src/parser/utils.rs:125-130): line and byte offset of the record handed
 * out last; *ending = 0 before the first record (the reference's None), 1 = Unix, 2 = Windows.  Out-pointers may be NULL. */
int ntk_reader_position(ntk_reader *r, uint64_t *line, uint64_t *byte, int *ending);
void ntk_reader_close(ntk_reader *r);
/* Drains the reader through n_batches (>= 2) pinned batches of batch_bytes each; results accumulate in the ctx
 * (ntk_accum_reset / ntk_accum_read around it).  A record longer than a whole batch is scanned through a one-off batch
 * of its own size (batch_bytes tunes the overlap, it does not limit the input). */
int ntk_scan_reader(ntk_ctx *ctx, ntk_reader *r, const ntk_params *p, uint64_t batch_bytes, uint32_t n_batches,
                    uint64_t *n_records, uint64_t *n_bases);

/* Parallel producer for PLAIN (uncompressed) input: the byte range is cut at record starts into about eight pieces per
 * thread, handed out on demand to n_threads parser threads (at most 64 are used; 32 reach the PCIe rate), each filling
 * its own pinned batches (record order is not preserved; the reduced result does not depend on it).  A gzip stream is sequential: ntk_scan_buffer_parallel refuses it (NTK_ERR_UNSUPPORTED, use
 * ntk_scan_reader; the same for bzip2 / xz / zstd).  ntk_scan_file_parallel takes a gzip file of any size: with n_threads >= 2 the file is
 * inflated by n_threads threads (every member, CRC-32 and ISIZE checked; ordinary one-member streams in parallel too, see ntk_gunzip below)
 * WHILE max(2, n_threads / 4) parser threads take the text as it becomes final, fill pinned batches and hand the pages back: the H2D copies
 * and the scans run inside the inflate's span and the resident text is bounded by NTK_OPT_GZ_STREAM_WINDOW_BYTES (BASELINE.json configs[4];
 * ntk_scan_file_info says what the run did).  Truncated or corrupt data is NTK_ERR_PARSE when the decoder reaches it - batches scanned before
 * that have been accumulated, as a caller of the reference's reader has seen the records before the Io error.  With n_threads = 1 the file
 * is inflated into memory first (at most NTK_OPT_GZ_INMEM_LIMIT_BYTES, else NTK_ERR_UNSUPPORTED).
 * Parse errors return NTK_ERR_PARSE without position detail. */
/* The cut points the parallel producer uses: cuts[0] = 0 <= cuts[1] <= ... <= cuts[n_pieces] = n, every cut a record start. */
int ntk_fastx_split_points(const uint8_t *data, uint64_t n, uint32_t n_pieces, uint64_t *cuts);
int ntk_scan_buffer_parallel(ntk_ctx *ctx, const uint8_t *data, uint64_t n, const ntk_params *p, uint64_t batch_bytes,
                             uint32_t n_threads, uint64_t *n_records, uint64_t *n_bases);
int ntk_scan_file_parallel(ntk_ctx *ctx, const char *path, const ntk_params *p, uint64_t batch_bytes, uint32_t n_threads,
                           uint64_t *n_records, uint64_t *n_bases);

/* The gzip front-end of the parallel producer on its own: every member of a gzip file inflated into ONE buffer by
 * n_threads threads, CRC-32 and ISIZE of every member checked; truncated or corrupt data is NTK_ERR_PARSE (what the reference's reader
 * reports as an Io error: MultiGzDecoder, src/parser/mod.rs:95-108), output beyond the in-memory limit NTK_ERR_UNSUPPORTED.  An
 * ORDINARY gzip stream (one member, as `gzip` writes it) is inflated in parallel too: chunks of the compressed bytes enter the deflate
 * stream at block boundaries found by search, decode with the 32 KiB of history before them as unknowns and are resolved once the
 * chunk before them is known (route 2); block gzip (bgzip) inflates member by member (route 1); n_threads = 1 is a plain sequential
 * inflate (route 3).  *out is released with ntk_gunzip_free(*out, *out_n) (an anonymous mapping, not malloc'ed memory); it is what
 * ntk_scan_buffer_parallel takes.  info may be NULL. */
typedef struct ntk_gunzip_info {
    uint32_t route, threads, chunks, chunks_dropped;   /* chunks_dropped: chunk starts that turned out not to be block boundaries */
    uint32_t members, streamed;                        /* streamed: the text was consumed while it was inflated (ntk_scan_file_parallel) */
    double search_s, decode_s, decode_busy_s, crc_s;   /* boundary search (CPU seconds / threads); wall seconds of the decode + resolve pipeline; CPU seconds inside the chunk decoders; CRC combination */
    uint64_t marker_symbols;                           /* symbols that went through the 16-bit "unknown window" form */
    /* ABI 4 */
    uint32_t chunks_deferred, parse_threads;           /* chunks that outgrew the speculative cap and were decoded again at the chain's head; parser threads of a streamed run */
    uint64_t peak_backlog_bytes, text_bytes;           /* streamed runs: the most inflated text that was placed and not yet handed to a parser; the text's size */
    double first_batch_s, total_s;                     /* streamed runs: seconds from the call to the first batch submitted / to the return */
    double resolve_busy_s;                             /* CPU seconds inside the marker replacement + copy + CRC of the resolved chunks */
} ntk_gunzip_info;
int ntk_gunzip(const uint8_t *gz, uint64_t n, uint32_t n_threads, uint8_t **out, uint64_t *out_n, ntk_gunzip_info *info);
void ntk_gunzip_free(uint8_t *out, uint64_t out_n);
/* What the gzip front-end did in the calling thread's last ntk_scan_file_parallel (route 0: the file was not gzip). */
int ntk_scan_file_info(ntk_gunzip_info *info);

/* ---- compat face: the reference's per-sequence functions, eager ---------------------------------
 * Caller-allocated outputs; *_len out-params; outputs need capacity n unless stated. */
/* sequence::normalize (src/sequence.rs:19-62): *changed == 0 <=> the reference returns None. */
int ntk_normalize(ntk_ctx *ctx, const uint8_t *seq, uint64_t n, int allow_iupac,
                  uint8_t *out, uint64_t *out_len, int *changed);
/* Sequence::strip_returns (src/sequence.rs:165-191): *borrowed != 0 <=> Cow::Borrowed. */
int ntk_strip_returns(ntk_ctx *ctx, const uint8_t *seq, uint64_t n,
                      uint8_t *out, uint64_t *out_len, int *borrowed);
/* Sequence::reverse_complement (src/sequence.rs:202-208, complement :68-105). */
int ntk_reverse_complement(ntk_ctx *ctx, const uint8_t *seq, uint64_t n, uint8_t *out);
/* Sequence::canonical_kmers(k, rc) (src/kmer.rs:48-130): item i is (pos_out[i], is_rc_out[i]); the
 * slice is seq[pos..pos+k] or rc[n-pos-k..n-pos] as in src/kmer.rs:121-128.  Raw-byte comparison,
 * any k in 1..255.  Returns NTK_ERR_CAPACITY (with *count = needed) if cap is too small. */
int ntk_canonical_kmers(ntk_ctx *ctx, const uint8_t *seq, uint64_t n, uint32_t k,
                        uint64_t *pos_out, uint8_t *is_rc_out, uint64_t cap, uint64_t *count);
/* Sequence::bit_kmers(k, canonical) (src/bitkmer.rs:72-109): items (pos, (value,k), was_rc); k 1..32. */
int ntk_bit_kmers(ntk_ctx *ctx, const uint8_t *seq, uint64_t n, uint32_t k, int canonical,
                  uint64_t *pos_out, uint64_t *val_out, uint8_t *was_rc_out, uint64_t cap, uint64_t *count);

/* The same for a whole batch of records in ONE call (one upload, one scan, device-side compaction, one download) - what
 * an `impl Sequence` over a FastxReader loop should bind: a per-record call is launch-latency-bound for 150 bp reads.
 * seq + offsets[n_records + 1]: record i = seq[offsets[i] .. offsets[i + 1]).  counts[i] = items of record i; the item
 * arrays hold the records' items consecutively, each record's in the reference iterator's order (positions are relative
 * to the record).  cap >= sum over records of max(0, len - k + 1) always suffices; NTK_ERR_CAPACITY sets *total = needed.
 * ntk_canonical_kmers_batch: any k <= 255, raw-byte comparison as reference src/kmer.rs:121-128 (rc = the record's own
 * reverse complement).  ntk_bit_kmers_batch: k <= 32, reference src/bitkmer.rs:80-109. */
int ntk_canonical_kmers_batch(ntk_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k,
                              uint64_t *counts, uint64_t *pos_out, uint8_t *is_rc_out, uint64_t cap, uint64_t *total);
int ntk_bit_kmers_batch(ntk_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k, int canonical,
                        uint64_t *counts, uint64_t *pos_out, uint64_t *val_out, uint8_t *was_rc_out, uint64_t cap, uint64_t *total);
/* Both batched calls cut the batch into chunks of <= 16 MiB of packed bytes and keep up to three chunks in flight: while the copy engine
 * returns chunk i's items (9 or 17 bytes each - the bound of this face) the host packs chunk i + 1 into pinned staging and the
 * GPU scans it.  The caller's arrays may be pageable; page-locked ones (below) take the copies without a staging pass. */
int ntk_pinned_alloc(uint64_t bytes, void **out);   /* hipHostMalloc: for the arrays handed to the batched calls */
/* The byte-path items of a batch as TWO BITS per sequence byte instead of (pos, is_rc) per item: what Sequence::canonical_kmers yields
 * is (pos, buffer[pos..pos+k] or the rc slice, is_rc) (reference src/kmer.rs:114-129), and pos and slice follow from where the window
 * starts - so the device returns, per window START, "emitted" and "is_rc" as bit planes and the host iterator walks the bits
 * (include/needletail_amd.hpp CanonicalKmersPlanes, rust/src/amd.rs AmdCanonicalKmersPlanes, needletail_amd.canonical_kmers_planes).
 * 1/4 byte comes back per input byte where ntk_canonical_kmers_batch returns 9 bytes per item: the call is bound by the upload.
 * The records are uploaded as they lie in seq (no packing, no break bytes).  Plane position b = bit (15 - b % 16) of word b / 16;
 * the window starting at byte p of record r sits at b = rec_bit[r] + p, for p in 0 .. len(r) - k (positions past that, and the
 * padding up to the next 16-position boundary where a new chunk of the batch begins, read 0).  rec_bit has n_records + 1 entries
 * (the last = 16 * *n_words).  cap_words >= (offsets[n_records] - offsets[0]) / 16 + n_records + 1 always suffices;
 * NTK_ERR_CAPACITY sets *n_words = needed.  *total = items of the whole batch.  Any k <= 255, raw-byte comparison, ties -> is_rc. */
int ntk_canonical_kmers_batch_planes(ntk_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k,
                                     uint64_t *rec_bit, uint16_t *valid16, uint16_t *rc16, uint64_t cap_words, uint64_t *n_words,
                                     uint64_t *total);
/* Sequence::bit_kmers(k, canonical) (reference src/bitkmer.rs:97-108: items (pos, (value, k), was_rc)) for a batch in the same form: the two
 * planes per window START - "emitted" and "was_rc" (ties keep the forward k-mer, was_rc = 0, src/bitkmer.rs:136-143; canonical = 0: the
 * forward k-mer of every window, was_rc plane all 0) - and, when values != NULL, the items' packed values DENSE, one u64 per plane position
 * (values[rec_bit[r] + p] for the window starting at byte p of record r; 0 where nothing is emitted; 16 * cap_words u64): 8.25 bytes per
 * position come back instead of the 17 per item of ntk_bit_kmers_batch, with no compaction pass and no positions (the download of the
 * values is the call's bound; with values == NULL a quarter byte per base comes back and the host can pack the value of an emitted window from
 * its k bases, A0 C1 G2 T3, first base most significant).  Alphabet acgtACGT, k <= 32; geometry, capacity protocol and rec_bit exactly as
 * ntk_canonical_kmers_batch_planes.  Host iterators: BitKmersPlanes (needletail_amd.hpp), AmdBitKmersPlanes (rust/src/amd.rs),
 * needletail_amd.bit_kmers_planes. */
int ntk_bit_kmers_batch_planes(ntk_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t k, int canonical,
                               uint64_t *rec_bit, uint16_t *valid16, uint16_t *rc16, uint64_t *values, uint64_t cap_words, uint64_t *n_words,
                               uint64_t *total);
/* Releases what the batched calls and the compat face keep between calls (staging, device buffers); they are re-made on demand. */
int ntk_ctx_trim(ntk_ctx *ctx);
void ntk_pinned_free(void *p);

/* ---- minimizers and quality masking ------------------------------------------------ */
/* Windowed minimizers, reduce mode (BASELINE.json configs[4], "minimizers (w, k)"): for every window of w+k-1 good
 * bases the smallest canonical k-mer in it = sequence::minimizer(window, k) (reference src/sequence.rs:139-152).
 * Accumulates n_total (windows), n_fwd (minimizer drawn from the forward strand), sum/xor/histogram of the
 * minimizer values into the ctx accumulators.  p->path must be a canonical path.  Async on the ctx stream. */
int ntk_minimizers_reduce_device(ntk_ctx *ctx, const uint8_t *d_seq, uint64_t n_bytes, const ntk_params *p, uint32_t w);
/* sequence::minimizer (reference src/sequence.rs:139-152) for one sequence; out holds m bytes; n >= m >= 1. */
int ntk_minimizer(ntk_ctx *ctx, const uint8_t *seq, uint64_t n, uint32_t m, uint8_t *out);
/* sequence::minimizer (reference src/sequence.rs:139-152) for every record of a reader batch in one call: record r = the bytes
 * offsets[r] .. offsets[r+1] of seq (host arrays, as in ntk_canonical_kmers_batch); out + r * m receives its m bytes.  Optional
 * (may be NULL): pos_out[r] = the winning window's start in ITS strand's string (the sequence, or its reverse complement),
 * is_rc_out[r] = 1 when it was drawn from the reverse complement; among equal byte strings the window the reference's loop meets
 * first wins (forward window i, then reverse-complement window i, i ascending: src/sequence.rs:143-150).  A record shorter than m
 * (the reference panics there, :141) fails the call with NTK_ERR_BAD_ARG before anything is computed; *bad_record (may be NULL)
 * = its index, ~0 otherwise.  Synchronous. */
int ntk_minimizer_batch(ntk_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint64_t n_records, uint32_t m, uint8_t *out,
                        uint64_t *pos_out, uint8_t *is_rc_out, uint64_t *bad_record);
/* sequence::canonical (reference src/sequence.rs:110-134): the lexicographically lower of seq and its reverse complement
 * (raw-byte order, complement as ntk_reverse_complement); out holds n bytes; *was_rc = 1 when the reverse complement won. */
int ntk_canonical(ntk_ctx *ctx, const uint8_t *seq, uint64_t n, uint8_t *out, int *was_rc);
/* bitkmer::minimizer (reference src/bitkmer.rs:146-162), element-wise over n packed k-mers (host arrays). */
int ntk_bit_minimizers(ntk_ctx *ctx, const uint64_t *values, uint64_t n, uint32_t k, uint32_t m, uint64_t *out);
/* bitkmer::reverse_complement (reference src/bitkmer.rs:112-132) / bitkmer::canonical (:136-143), element-wise over n
 * packed k-mers (host arrays).  canonical = 0: out = reverse complement, was_rc_out ignored; canonical != 0: out =
 * canonical k-mer, was_rc_out[i] = 1 when the reverse complement was chosen (ties keep the forward k-mer). */
int ntk_bit_canonical(ntk_ctx *ctx, const uint64_t *values, uint64_t n, uint32_t k, int canonical, uint64_t *out,
                      uint8_t *was_rc_out);
/* QualitySequence::quality_mask (reference src/sequence.rs:285-296): out[i] = qual[i] < score ? 'N' : seq[i]. */
int ntk_quality_mask(ntk_ctx *ctx, const uint8_t *seq, const uint8_t *qual, uint64_t n, uint8_t score, uint8_t *out);

/* ---- device utilities (bench / parity properties; records stay resident in HBM) ----------------- */
/* Synthetic read set of SURVEY.md §8d generated straight into HBM: reads first_read..+n_reads, each
 * read_len bases + '\n'; d_out holds n_reads*(read_len+1) bytes (+ padding to 16). */
int ntk_synth_reads_device(ntk_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                           uint32_t read_len, uint32_t n_per_1024, uint8_t *d_out);
/* Reverse-complement every fixed-stride record in place order (record r = bytes [r*stride, r*stride+len)),
 * bytes outside records copied: the batch form of Sequence::reverse_complement. */
int ntk_reverse_complement_records_device(ntk_ctx *ctx, const uint8_t *d_in, uint8_t *d_out,
                                          uint64_t n_records, uint32_t record_len, uint32_t stride);

#ifdef __cplusplus
}
#endif
#endif /* NEEDLETAIL_AMD_H */
