// ntk_fastx.cpp — see ntk_fastx.hpp.  Host-only C++ (zlib for gzip); part of libneedletail_amd.so.
#include "ntk_fastx.hpp"

#include <dlfcn.h>
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

namespace ntk {

namespace {
constexpr size_t kBufSize = 64 * 1024;  // reference src/parser/utils.rs:8

size_t grow_to(size_t cur)  // reference src/parser/utils.rs:24-30
{
    return cur < ((size_t)1 << 23) ? cur * 2 : cur + ((size_t)1 << 23);
}

inline size_t trim_cr_len(const uint8_t *p, size_t n)  // reference src/parser/utils.rs:12-18
{
    return (n && p[n - 1] == '\r') ? n - 1 : n;
}

std::string escape_byte(uint8_t b)  // Rust's char::escape_default for the bytes that can show up here
{
    char tmp[16];
    if (b == '\n') return "\\n";
    if (b == '\r') return "\\r";
    if (b == '\t') return "\\t";
    if (b == '\\') return "\\\\";
    if (b == '\'') return "\\'";
    if (b == '"') return "\\\"";
    if (b >= 0x20 && b < 0x7f) { tmp[0] = (char)b; tmp[1] = 0; return tmp; }
    snprintf(tmp, sizeof(tmp), "\\u{%x}", b);
    return tmp;
}

std::string first_word(const uint8_t *p, size_t n)
{
    size_t e = 0;
    while (e < n && p[e] != ' ') e++;
    return std::string((const char *)p, e);
}
}  // namespace

// (bzip2 / xz / zstd input: ntk_fastx_codecs.cpp - an optional translation unit outside SURVEY.md section 8, see there)

FastxReader::~FastxReader()
{
    delete dec_;
    if (zs_) { inflateEnd(zs_); delete zs_; }
    if (fp_ && !is_stdin_) fclose(fp_);
}

bool FastxReader::fail(int kind, const std::string &msg, uint64_t line, const std::string &id)
{
    err_kind_ = kind; err_msg_ = msg; err_line_ = line; err_id_ = id; finished_ = true;
    return false;
}

size_t FastxReader::read_raw(uint8_t *dst, size_t cap)
{
    if (fp_) return fread(dst, 1, cap, fp_);
    const size_t n = mem_n_ - mem_pos_ < cap ? (size_t)(mem_n_ - mem_pos_) : cap;
    if (n) memcpy(dst, mem_ + mem_pos_, n);
    mem_pos_ += n;
    return n;
}

size_t FastxReader::read_plain(uint8_t *dst, size_t cap)
{
    if (dec_) {
        if (z_eof_ || cap == 0) return 0;
        size_t produced = 0;
        while (produced == 0) {   // until at least one byte is produced or the stream / the input ends
            if (zin_pos_ == zin_len_) {
                zin_len_ = read_raw(zin_.data(), zin_.size());
                zin_pos_ = 0;
            }
            const bool no_more_input = zin_len_ == zin_pos_;
            size_t used = 0, made = 0; bool end = false;
            if (!dec_->step(zin_.data() + zin_pos_, zin_len_ - zin_pos_, &used, dst, cap, &made, &end)) {
                fail(kErrIo, std::string(dec_->name()) + " stream error: corrupt data", line_);
                return (size_t)-1;
            }
            zin_pos_ += used; produced += made;
            if (end) { z_eof_ = true; break; }
            if (no_more_input && made == 0) {
                if (dec_->mid_stream()) {   // the reference's decoders return UnexpectedEof here (a partial download must not parse)
                    fail(kErrIo, std::string(dec_->name()) + " stream error: unexpected end of compressed stream", line_);
                    return (size_t)-1;
                }
                z_eof_ = true; break;
            }
        }
        return produced;
    }
    if (!gz_) return read_raw(dst, cap);
    if (z_eof_ || cap == 0) return 0;
    zs_->next_out = dst;
    zs_->avail_out = (uInt)(cap > 0x40000000u ? 0x40000000u : cap);
    const size_t want = zs_->avail_out;
    while (zs_->avail_out == want) {  // until at least one byte is produced or the input ends
        if (zin_pos_ == zin_len_) {
            zin_len_ = read_raw(zin_.data(), zin_.size());
            zin_pos_ = 0;
            if (zin_len_ == 0) {
                // the input ends inside a member (no trailer seen): flate2's MultiGzDecoder reports UnexpectedEof
                // (reference src/parser/mod.rs:95-97); a truncated download must not parse as a shorter file
                fail(kErrIo, "gzip stream error: unexpected end of compressed stream", line_);
                return (size_t)-1;
            }
        }
        zs_->next_in = zin_.data() + zin_pos_;
        zs_->avail_in = (uInt)(zin_len_ - zin_pos_);
        const int rc = inflate(zs_, Z_NO_FLUSH);
        zin_pos_ = zin_len_ - zs_->avail_in;
        if (rc == Z_STREAM_END) {
            // concatenated members (MultiGzDecoder, reference src/parser/mod.rs:96): restart on the next one
            if (zin_pos_ == zin_len_) {
                zin_len_ = read_raw(zin_.data(), zin_.size());
                zin_pos_ = 0;
            }
            if (zin_len_ == zin_pos_) { z_eof_ = true; break; }
            inflateReset(zs_);
        } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
            fail(kErrIo, std::string("gzip stream error: ") + (zs_->msg ? zs_->msg : "corrupt data"), line_);
            return (size_t)-1;
        }
    }
    return want - zs_->avail_out;
}

size_t FastxReader::fill()
{
    if (eof_) return 0;
    if (len_ == buf_.size()) make_room_or_grow();
    size_t added = 0;
    while (len_ < buf_.size()) {  // fill_buf keeps reading until the buffer is full or EOF (reference utils.rs:34-49)
        const size_t n = read_plain(buf_.data() + len_, buf_.size() - len_);
        if (n == (size_t)-1) { eof_ = true; return added; }
        if (n == 0) { eof_ = true; break; }
        len_ += n; added += n;
    }
    return added;
}

void FastxReader::make_room_or_grow()
{
    if (start_ > 0) {  // make_room: drop consumed records, move the incomplete one to the front
        memmove(buf_.data(), buf_.data() + start_, len_ - start_);
        len_ -= start_;
        start_ = 0;
    } else {           // grow: a single record does not fit
        buf_.resize(grow_to(buf_.size()));
    }
}

bool FastxReader::sniff()
{
    buf_.assign(kBufSize, 0);
    len_ = start_ = 0;
    uint8_t two[2];
    size_t got = read_raw(two, 2);
    if (got == 1) got += read_raw(two + 1, 1);
    if (got < 2) return fail(kErrEmptyFile, "Failed to read the first two bytes. Is the file empty?", 0);
    if (two[0] == 0x1F && two[1] == 0x8B) {  // GZ_MAGIC, reference src/parser/mod.rs:29,95-108
        gz_ = true;
        zin_.assign(kBufSize, 0);
        zin_[0] = two[0]; zin_[1] = two[1];
        zin_pos_ = 0; zin_len_ = 2;
        zs_ = new z_stream_s();
        memset(zs_, 0, sizeof(*zs_));
        if (inflateInit2(zs_, 15 + 16) != Z_OK) return fail(kErrIo, "zlib initialisation failed", 0);
    } else if ((two[0] == 0x42 && two[1] == 0x5A) || (two[0] == 0xFD && two[1] == 0x37) || (two[0] == 0x28 && two[1] == 0xB5)) {
        // BZ_MAGIC / XZ_MAGIC / ZST_MAGIC, reference src/parser/mod.rs:30-35,109-147
        dec_ = make_stream_decoder(two[0]);
        if (!dec_) return fail(kErrIo, "compressed input, but the codec's run-time library (libbz2 / liblzma / libzstd) cannot be loaded", 0);
        zin_.assign(kBufSize, 0);
        zin_[0] = two[0]; zin_[1] = two[1];
        zin_pos_ = 0; zin_len_ = 2;
    } else if (mem_ && !fp_) {
        // plain text in the caller's memory (which outlives the reader): records are parsed where they lie - no pass through the
        // reader's own buffer (copying every byte once and moving record tails to the buffer's front cost a third of the time
        // a parser thread of the parallel producer spends per record)
        view_ = mem_;
        len_ = (size_t)mem_n_;
        mem_pos_ = mem_n_;
        eof_ = true;
        buf_.clear(); buf_.shrink_to_fit();
    } else {
        buf_[0] = two[0]; buf_[1] = two[1];
        len_ = 2;
    }
    if (len_ == 0) {
        fill();
        if (err_kind_) return false;
        if (len_ == 0) return fail(kErrEmptyFile, "Failed to read the first two bytes. Is the file empty?", 0);
    }
    const uint8_t first = data()[0];
    if (first == '>') format_ = kFasta;
    else if (first == '@') format_ = kFastq;
    else return fail(kErrUnknownFormat, "Expected '@' or '>' at the start of the file but found '" + escape_byte(first) + "'.", 0);
    started_ = true;
    line_ = 1;
    return true;
}

bool FastxReader::open_file(const char *path)
{
    // "-" is standard input: parse_fastx_stdin (reference src/parser/mod.rs:154-159)
    if (path[0] == '-' && path[1] == 0) { fp_ = stdin; is_stdin_ = true; return sniff(); }
    fp_ = fopen(path, "rb");
    if (!fp_) return fail(kErrIo, std::string("cannot open ") + path, 0);
    return sniff();
}

bool FastxReader::open_memory(const uint8_t *data, uint64_t n)
{
    mem_ = data; mem_n_ = n; mem_pos_ = 0;
    return sniff();
}

int FastxReader::next(FastxRecord *rec)
{
    if (finished_ || !started_) return err_kind_ ? -1 : 0;
    // release the previous record
    start_ += prev_len_; line_ += prev_lines_; byte_ += prev_len_;
    prev_len_ = 0; prev_lines_ = 0;
    return format_ == kFasta ? next_fasta(rec) : next_fastq(rec);
}

// find_line_ending over the first record's bytes (reference src/parser/utils.rs:106-117, fasta.rs:358-360, fastq.rs:438-440)
void FastxReader::note_line_ending(const uint8_t *all, size_t n)
{
    if (line_ending_ || n == 0) return;
    const uint8_t *p = (const uint8_t *)memchr(all, '\n', n);
    if (p) line_ending_ = (p > all && p[-1] == '\r') ? 2 : 1;
}

// '\n' and '\r' counts of [p, p+n): 32-byte compares where the CPU has AVX2, a plain loop otherwise
__attribute__((target("avx2"))) static void count_nl_cr_avx2(const uint8_t *p, size_t n, uint64_t *nl, uint64_t *cr)
{
    uint64_t a = 0, b = 0;
    size_t i = 0;
    const __m256i vn = _mm256_set1_epi8('\n'), vr = _mm256_set1_epi8('\r');
    for (; i + 32 <= n; i += 32) {
        const __m256i x = _mm256_loadu_si256((const __m256i *)(p + i));
        a += (uint64_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vn)));
        b += (uint64_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vr)));
    }
    for (; i < n; i++) { a += p[i] == '\n'; b += p[i] == '\r'; }
    *nl = a; *cr = b;
}
static void count_nl_cr_plain(const uint8_t *p, size_t n, uint64_t *nl, uint64_t *cr)
{
    uint64_t a = 0, b = 0;
    size_t i = 0;
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
    const __m128i vn = _mm_set1_epi8('\n'), vr = _mm_set1_epi8('\r');
    for (; i + 16 <= n; i += 16) {
        const __m128i x = _mm_loadu_si128((const __m128i *)(p + i));
        a += (uint64_t)__builtin_popcount((uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vn)));
        b += (uint64_t)__builtin_popcount((uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vr)));
    }
#endif
    for (; i < n; i++) { a += p[i] == '\n'; b += p[i] == '\r'; }
    *nl = a; *cr = b;
}
static inline void count_nl_cr(const uint8_t *p, size_t n, uint64_t *nl, uint64_t *cr)
{
#if defined(__HIP_DEVICE_COMPILE__)   // hipcc also runs a device pass over this host-only file
    count_nl_cr_plain(p, n, nl, cr);
#else
    static void (*const fn)(const uint8_t *, size_t, uint64_t *, uint64_t *) = [] {
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx2") ? count_nl_cr_avx2 : count_nl_cr_plain;
    }();
    fn(p, n, nl, cr);
#endif
}

int FastxReader::next_fasta(FastxRecord *rec)
{
    if (start_ == len_) {
        if (!eof_) fill();
        if (err_kind_) return -1;
        if (start_ == len_) { finished_ = true; return 0; }
    }
    // A record runs from '>' to the last '\n' before the next line that starts with '>' (or to EOF): the header's line
    // feed is one memchr, the next record one memchr for '>' per candidate (a '>' counts only right after a line feed), and
    // the line count (reference fasta.rs:235 pushes every line end) one vectorised pass - not a memchr per 60..80-byte
    // line.  Offsets are relative to start_ so that they survive make_room() / grow() inside fill().
    size_t scan = 1, first_nl = (size_t)-1, last_nl = (size_t)-1, rec_len = 0;
    for (;;) {
        const uint8_t *base = data() + start_;
        const size_t avail = len_ - start_;
        bool complete = false;
        if (first_nl == (size_t)-1 && scan < avail) {
            const uint8_t *p = (const uint8_t *)memchr(base + scan, '\n', avail - scan);
            if (p) { first_nl = (size_t)(p - base); scan = first_nl + 1; } else scan = avail;
        }
        while (first_nl != (size_t)-1 && scan < avail) {
            const uint8_t *p = (const uint8_t *)memchr(base + scan, '>', avail - scan);
            if (!p) { scan = avail; break; }
            const size_t pos = (size_t)(p - base);
            if (base[pos - 1] == '\n') { rec_len = pos; last_nl = pos - 1; complete = true; break; }
            scan = pos + 1;
        }
        if (complete) break;
        if (!eof_) {
            fill();
            if (err_kind_) return -1;
            continue;
        }
        // EOF: the record ends with the input (fasta.rs:204-212).  A header line with nothing after it - with or without its
        // own line feed - is a truncated record (seq_pos stays empty -> fasta.rs:342-350).
        if (first_nl == (size_t)-1 || first_nl + 1 == avail) {
            fail(kErrUnexpectedEnd, "Unexpected end of input", line_);
            return -1;
        }
        last_nl = base[avail - 1] == '\n' ? avail - 1 : avail;  // no final line feed: fasta.rs:208 pushes the buffer end
        rec_len = avail;
        break;
    }
    const uint8_t *base = data() + start_;
    rec->format = kFasta;
    rec->line = line_;
    rec->id = base + 1;
    rec->id_len = trim_cr_len(base + 1, first_nl - 1);
    uint64_t n_lines = 1, nl = 0, cr = 0;  // the header's line end
    if (last_nl > first_nl) {  // fasta.rs:55-63 raw_seq
        rec->seq = base + first_nl + 1;
        rec->seq_len = trim_cr_len(rec->seq, last_nl - first_nl - 1);
        count_nl_cr(rec->seq, rec->seq_len, &nl, &cr);
        n_lines += nl + 1;     // interior line ends + the record's last one (real or the buffer end)
    } else {
        rec->seq = base + first_nl; rec->seq_len = 0;
    }
    rec->qual = nullptr; rec->qual_len = 0;
    rec->num_bases = rec->seq_len - nl - cr;  // fasta.rs:102-107
    prev_len_ = rec_len; prev_lines_ = n_lines;
    // the reference looks at the record without its final line end (fasta.rs:40-42 `all`): ">id\nACGT"
    note_line_ending(base, last_nl < rec_len ? last_nl : rec_len);
    rec->byte = byte_; rec->line_ending = line_ending_ ? line_ending_ : 1;  // record.rs:39,53 unwrap_or(Unix)
    return 1;
}

// The first `want` line feeds of [p, p+n): one pass with 32-byte compares where the CPU has AVX2 (a FASTQ record is four
// short lines: four memchr calls cost more in call overhead than in scanning), plain memchr otherwise.
__attribute__((target("avx2"))) static int find_newlines_avx2(const uint8_t *p, size_t n, size_t *out, int want)
{
    int found = 0;
    size_t i = 0;
    const __m256i nlv = _mm256_set1_epi8('\n');
    for (; i + 32 <= n && found < want; i += 32) {
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nlv));
        while (m && found < want) { out[found++] = i + (size_t)__builtin_ctz(m); m &= m - 1; }
    }
    for (; i < n && found < want; i++)
        if (p[i] == '\n') out[found++] = i;
    return found;
}
static int find_newlines_memchr(const uint8_t *p, size_t n, size_t *out, int want)
{
    int found = 0;
    size_t from = 0;
    while (found < want && from < n) {
        const uint8_t *q = (const uint8_t *)memchr(p + from, '\n', n - from);
        if (!q) break;
        out[found++] = (size_t)(q - p);
        from = out[found - 1] + 1;
    }
    return found;
}
static inline int find_newlines(const uint8_t *p, size_t n, size_t *out, int want)
{
#if defined(__HIP_DEVICE_COMPILE__)   // hipcc also runs a device pass over this host-only file
    return find_newlines_memchr(p, n, out, want);
#else
    static int (*const fn)(const uint8_t *, size_t, size_t *, int) = [] {
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx2") ? find_newlines_avx2 : find_newlines_memchr;
    }();
    return fn(p, n, out, want);
#endif
}

int FastxReader::next_fastq(FastxRecord *rec)
{
    size_t nl[4];
    int found = 0;
    for (;;) {
        const uint8_t *base = data() + start_;
        const size_t avail = len_ - start_;
        if (view_ && avail > 4096) {   // records parsed where they lie come straight from DRAM: ask for the lines a few records ahead now
            // (without it the zero-copy path was SLOWER than the copying one, 69 against 44 ns per 150-base record: the copy had been the
            //  prefetch; with it 37 ns, and the record's bytes are still in L1 when the packer copies them)
            __builtin_prefetch(base + 2048); __builtin_prefetch(base + 2112); __builtin_prefetch(base + 2176);
            __builtin_prefetch(base + 2240); __builtin_prefetch(base + 2304);
        }
        found = find_newlines(base, avail, nl, 4);
        if (found == 4) break;
        if (!eof_) {
            fill();
            if (err_kind_) return -1;
            continue;
        }
        // EOF with an incomplete record: reference check_end, src/parser/fastq.rs:335-355
        if (found == 3) { nl[3] = avail; break; }  // no line ending at the end of the last record
        bool blank = true;
        for (size_t i = 0, ls = 0; i <= avail && blank; i++) {
            if (i == avail || base[i] == '\n') { blank = trim_cr_len(base + ls, i - ls) == 0; ls = i + 1; }
        }
        finished_ = true;
        if (blank) return 0;
        std::string id;
        if (found > 0 && nl[0] > 1) id = first_word(base + 1, trim_cr_len(base + 1, nl[0] - 1));
        fail(kErrUnexpectedEnd, "Unexpected end of input", line_ + (uint64_t)found, id);
        return -1;
    }
    const uint8_t *base = data() + start_;
    const size_t seq0 = nl[0] + 1, sep0 = nl[1] + 1, qual0 = nl[2] + 1, end = nl[3];
    // validate, reference src/parser/fastq.rs:240-285
    if (base[0] != '@') {
        fail(kErrInvalidStart, "Expected '@' but found '" + escape_byte(base[0]), line_);
        return -1;
    }
    // the record id is only needed for error messages: built on the error paths, not per record
    auto rec_id = [&]() { return seq0 > 1 ? first_word(base + 1, trim_cr_len(base + 1, nl[0] - 1)) : std::string(); };
    if (base[sep0] != '+') {
        fail(kErrInvalidSeparator, "Expected '+' separator but found '" + escape_byte(base[sep0]), line_ + 2, rec_id());
        return -1;
    }
    const size_t seq_len = trim_cr_len(base + seq0, nl[1] - seq0);
    const size_t qual_len = trim_cr_len(base + qual0, end - qual0);
    if (seq_len != qual_len) {
        fail(kErrUnequalLengths, "Sequence length is " + std::to_string(seq_len) + " but quality length is " + std::to_string(qual_len), line_, rec_id());
        return -1;
    }
    rec->format = kFastq;
    rec->line = line_;
    rec->id = base + 1; rec->id_len = trim_cr_len(base + 1, nl[0] - 1);
    rec->seq = base + seq0; rec->seq_len = seq_len;
    rec->qual = base + qual0; rec->qual_len = qual_len;
    rec->num_bases = seq_len;
    prev_len_ = end < len_ - start_ ? end + 1 : end;
    prev_lines_ = 4;
    note_line_ending(base, end);
    rec->byte = byte_; rec->line_ending = line_ending_ ? line_ending_ : 1;  // record.rs:39,53 unwrap_or(Unix)
    return 1;
}

}  // namespace ntk
