// ntk_fastx.hpp — the CPU producer of the hot path: a streaming FASTA/FASTQ record reader with needletail's
// FastxReader semantics (reference src/parser/{mod,fasta,fastq,utils}.rs), kept on the CPU as BASELINE.json's north_star
// says.  It hands out one borrowed record at a time (valid until the next call, like reference
// src/parser/utils.rs:123) so that the caller can copy the sequence into a pinned batch (ntk_batch_append).
//
// What is mirrored (and tested against the reference's own data files):
//   * parse_fastx_reader's sniffing: fewer than 2 bytes -> EmptyFile; gzip magic 1F 8B -> concatenated-member inflate
//     (MultiGzDecoder); 'BZ' -> bzip2, FD 37 -> xz, 28 B5 -> zstd (their run-time libraries are loaded on demand; an Io
//     error when one is not installed); first byte '>' -> FASTA, '@' -> FASTQ, else UnknownFormat
//     (reference src/parser/mod.rs:27-35,85-147)
//   * FASTA: records split at "\n>", raw sequence = everything between the header's '\n' and the record's last '\n'
//     (interior line breaks kept, one trailing '\r' trimmed); header without any newline at EOF -> UnexpectedEnd
//     (reference src/parser/fasta.rs:55-63,196-243,291-367)
//   * FASTQ: strict 4-line records, '@' / '+' checks, equal sequence/quality lengths, last record may lack its newline,
//     trailing blank lines allowed, otherwise UnexpectedEnd   (reference src/parser/fastq.rs:155-187,240-285,335-355)
//   * buffer policy: 64 KiB, doubling to 8 MiB, then +8 MiB steps (reference src/parser/utils.rs:8,24-30)
// Not mirrored here: header masking and the record writers (they live in the host mirrors).
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

struct z_stream_s;

namespace ntk {
// A streaming decoder of a compressed container other than gzip (ntk_fastx_codecs.cpp; nullptr from make_stream_decoder: the codec's
// run-time library is not installed, or the library was built without that translation unit)
struct StreamDecoder {
    virtual ~StreamDecoder() {}
    // consumes from [in, in+in_n), produces into [out, out+cap); returns false on a stream error.  *end is set when the
    // compressed stream is complete.
    virtual bool step(const uint8_t *in, size_t in_n, size_t *used, uint8_t *out, size_t cap, size_t *made, bool *end) = 0;
    // input that runs out while this is true is a TRUNCATED stream (an error, like the reference's decoders: UnexpectedEof),
    // not an end of file.  bzip2 / xz: true until the stream end marker has been seen (read_plain stops at the marker).
    virtual bool mid_stream() const { return true; }
    virtual const char *name() const = 0;
};
StreamDecoder *make_stream_decoder(uint8_t first_magic_byte);   // 0x42 bzip2, 0xFD xz, 0x28 zstd
}

namespace ntk {

enum FastxFormat { kFasta = 0, kFastq = 1 };
enum FastxErrorKind {  // reference src/errors.rs:26-44
    kErrNone = 0, kErrIo = 1, kErrUnknownFormat = 2, kErrInvalidStart = 3, kErrInvalidSeparator = 4,
    kErrUnequalLengths = 5, kErrUnexpectedEnd = 6, kErrEmptyFile = 7
};

struct FastxRecord {
    const uint8_t *id = nullptr; uint64_t id_len = 0;      // header line without '>'/'@', trailing '\r' trimmed
    const uint8_t *seq = nullptr; uint64_t seq_len = 0;    // raw_seq(): FASTA keeps interior line breaks
    const uint8_t *qual = nullptr; uint64_t qual_len = 0;  // FASTQ only (nullptr for FASTA)
    int format = kFasta;
    uint64_t line = 0;        // start_line_number()
    uint64_t num_bases = 0;   // reference src/parser/fasta.rs:102-107 / fastq.rs:52
    uint64_t byte = 0;        // position().byte(): offset of the record start in the (decompressed) stream
    int line_ending = 0;      // SequenceRecord::line_ending(): 1 = Unix, 2 = Windows
};

class FastxReader {
public:
    FastxReader() = default;
    ~FastxReader();
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;

    bool open_file(const char *path);                    // parse_fastx_file; "-" = parse_fastx_stdin
    bool open_memory(const uint8_t *data, uint64_t n);   // parse_fastx_reader over a byte slice (data must outlive the reader)
    // 1 = record, 0 = end of input, -1 = error (see error_*)
    int next(FastxRecord *rec);

    int error_kind() const { return err_kind_; }
    const std::string &error_msg() const { return err_msg_; }
    uint64_t error_line() const { return err_line_; }
    const std::string &error_id() const { return err_id_; }
    int format() const { return format_; }
    // FastxReader::position (reference src/parser/utils.rs:125-126): line / byte of the record handed out last
    uint64_t position_line() const { return line_; }
    uint64_t position_byte() const { return byte_; }
    // FastxReader::line_ending (reference src/parser/utils.rs:127-130): 0 before the first record, 1 = Unix, 2 = Windows
    int line_ending() const { return line_ending_; }

private:
    // raw source
    FILE *fp_ = nullptr; bool is_stdin_ = false;
    const uint8_t *mem_ = nullptr; uint64_t mem_n_ = 0, mem_pos_ = 0;
    size_t read_raw(uint8_t *dst, size_t cap);
    // gzip layer
    StreamDecoder *dec_ = nullptr;   // bzip2 / xz / zstd, through their run-time libraries
    bool gz_ = false; z_stream_s *zs_ = nullptr; std::vector<uint8_t> zin_; size_t zin_pos_ = 0, zin_len_ = 0; bool z_eof_ = false;
    size_t read_plain(uint8_t *dst, size_t cap);   // after optional inflate; 0 = EOF; (size_t)-1 = error
    // record buffer
    std::vector<uint8_t> buf_; size_t len_ = 0, start_ = 0; bool eof_ = false;
    const uint8_t *view_ = nullptr;   // plain in-memory input: the caller's bytes themselves (buf_ unused)
    const uint8_t *data() const { return view_ ? view_ : buf_.data(); }
    size_t fill();             // appends to buf_, returns bytes added (0 at EOF)
    void make_room_or_grow();
    int format_ = kFasta; bool started_ = false, finished_ = false;
    uint64_t line_ = 1;        // line number of the record about to be parsed
    size_t prev_len_ = 0; uint64_t prev_lines_ = 0;
    uint64_t byte_ = 0; int line_ending_ = 0;
    void note_line_ending(const uint8_t *all, size_t n);
    bool fail(int kind, const std::string &msg, uint64_t line, const std::string &id = std::string());
    int next_fasta(FastxRecord *rec);
    int next_fastq(FastxRecord *rec);
    bool sniff();
    int err_kind_ = kErrNone; std::string err_msg_, err_id_; uint64_t err_line_ = 0;
};

}  // namespace ntk
