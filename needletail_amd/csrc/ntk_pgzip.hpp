// ntk_pgzip.hpp — parallel inflate of an ORDINARY gzip stream (one member, or a few large ones), for the CPU producer of the hot path
// (BASELINE.json configs[4]: "gzip FASTQ stream ... CPU decompress overlapped with GPU k-mer").
//
// What the reference does there: `MultiGzDecoder` on the reader thread (reference src/parser/mod.rs:95-108) - every member of the file,
// one core, `UnexpectedEof` on a truncated stream, an Io error on corrupt data.  This keeps those semantics (every member, CRC-32 and
// ISIZE of every member checked, truncation and corruption are errors) and lifts the one-core wall: a deflate stream can be entered at
// any BLOCK boundary if the 32 KiB of history before it are treated as unknowns.
//   1. the compressed bytes are cut into chunks; every chunk but the first looks for a deflate block header after its first byte (a
//      dynamic-Huffman header whose code-length code, literal/length code and distance code are all complete prefix codes, whose block
//      decodes without error and is followed by another plausible header: ~10^-15 false positives per bit offset, and a false positive
//      only costs its chunk - below);
//   2. all chunks inflate in parallel.  A chunk that starts in the middle of the stream writes 16-bit symbols: 0..255 = a byte,
//      32768 + w = "the byte at position w of the unknown 32 KiB window"; a back-reference copies symbols, markers included.  Once the last
//      32 KiB it produced hold no marker it continues as a plain byte decoder.  A chunk stops at the block boundary where the next chunk
//      started; if it runs past that position the next chunk's start was no boundary at all and its work is dropped;
//   3. the chunks' windows follow in stream order (only a chunk's last 32 KiB need resolving for that), then all chunks replace their
//      markers and land in the output buffer in parallel; CRC-32 per member over the pieces (crc32_combine).
// (The scheme is the one of pugz / rapidgzip; the code is this repository's own.)
#pragma once
#include <stdint.h>

#include <condition_variable>
#include <mutex>

namespace ntk {

struct PgzStats {
    uint32_t threads = 0, chunks = 0, chunks_dropped = 0, members = 0;
    uint32_t chunks_deferred = 0;   // chunks that outgrew the speculative cap and were decoded again at the head of the chain
    double search_s = 0, decode_s = 0, crc_s = 0;   // wall seconds: boundary search, the decode + resolve pipeline, the CRC combination
    double decode_busy_s = 0;                                     // summed over threads: seconds inside the chunk decoder
    double resolve_busy_s = 0;                                    // ... inside the marker replacement + copy + CRC of the resolved chunks
    uint64_t marker_symbols = 0;                                  // symbols that went through the 16-bit form
};

// Inflates every member of the gzip file in[0, n) into one buffer (*out, *out_n; release it with pgz_free(*out, *out_n): an anonymous
// mapping, not malloc'ed memory).  limit = largest output accepted.
// Returns 0 = ok, 1 = corrupt / truncated input, 2 = output larger than limit, 3 = out of memory (4 = cancelled by the stream's consumer).
// n_threads = 1 runs the same decoder sequentially (no speculation).
void pgz_free(uint8_t *p, uint64_t n);
uint8_t *pgz_alloc(uint64_t n);   // a buffer pgz_free releases (n bytes, page-aligned, huge pages advised); nullptr: out of memory
int pgz_inflate(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, uint8_t **out, uint64_t *out_n, PgzStats *stats);

// The progressive form (round 6: BASELINE.json configs[4] asks for the decompression to be OVERLAPPED with the GPU work).  The text appears in
// ONE contiguous address range (reserved up front, touched as it is written); a consumer reads [0, ready) while the decoder is still running
// and says how far it has got (consumed): the decoder places no chunk while more than `window` bytes are placed and not yet consumed, so
// the resident memory of a run is the window plus the chunks in flight, whatever the file's size.  The consumer may hand the pages below
// `consumed` back (madvise(MADV_DONTNEED)): the decoder never reads its own output again.
struct PgzStream {
    std::mutex mu;                      // guards everything below; the decoder's own pipeline state lives under it as well
    std::condition_variable cv;         // notified on every change (ready, consumed, finished)
    const uint8_t *base = nullptr;      // set before the first byte is ready
    uint64_t ready = 0;                 // bytes [0, ready) are final
    uint64_t consumed = 0;              // written by the consumer (then cv.notify_all()): it will not read below this offset again
    uint64_t window = (uint64_t)512 << 20;
    bool input_is_file_mapping = false; // the compressed bytes are a private read-only FILE mapping the caller owns: the decoder drops its pages behind
                                        // the chain's head (madvise(MADV_DONTNEED): they come back from the page cache if ever touched again)
    bool cancel = false;                // set by the consumer (then cv.notify_all()): the decoder stops at once and returns 4
    bool finished = false;              // the decoder has returned: rc says how (member CRCs and sizes are checked when the last byte is ready)
    int rc = 0;
    uint64_t peak_backlog = 0;          // largest (placed - consumed) the run saw
    // set by pgz_inflate_stream for pgz_stream_release
    uint8_t *map = nullptr; uint64_t map_bytes = 0;
};
// Runs the decoder on the calling thread + n_threads - 1 helpers and returns when every byte is ready (or on the first error); s->finished /
// s->rc are set before it returns.  limit = largest output accepted (the address range reserved up front; halved until the system grants it).
int pgz_inflate_stream(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, PgzStream *s, PgzStats *stats);
void pgz_stream_release(PgzStream *s);   // unmaps the output range (after the consumer is done with it)

}  // namespace ntk
