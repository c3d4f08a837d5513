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

namespace ntk {

struct PgzStats {
    uint32_t threads = 0, chunks = 0, chunks_dropped = 0, members = 0;
    double search_s = 0, decode_s = 0, crc_s = 0;   // wall seconds: boundary search, the decode + resolve pipeline, the CRC combination
    double decode_busy_s = 0;                                     // summed over threads: seconds inside the chunk decoder
    uint64_t marker_symbols = 0;                                  // symbols that went through the 16-bit form
};

// Inflates every member of the gzip file in[0, n) into one buffer (*out, *out_n; release it with pgz_free(*out, *out_n): an anonymous
// mapping, not malloc'ed memory).  limit = largest output accepted.
// Returns 0 = ok, 1 = corrupt / truncated input, 2 = output larger than limit, 3 = out of memory.
// n_threads = 1 runs the same decoder sequentially (no speculation).
void pgz_free(uint8_t *p, uint64_t n);
uint8_t *pgz_alloc(uint64_t n);   // a buffer pgz_free releases (n bytes, page-aligned, huge pages advised); nullptr: out of memory
int pgz_inflate(const uint8_t *in, uint64_t n, uint32_t n_threads, uint64_t limit, uint8_t **out, uint64_t *out_n, PgzStats *stats);

}  // namespace ntk
