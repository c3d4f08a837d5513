// needletail_amd.hpp — header-only C++ mirror of needletail's public surface for the accelerated path, on top of the
// C ABI (needletail_amd.h).  Names, argument meaning and results follow the reference so that code written against
//   needletail::{parse_fastx_file, FastxReader::next, SequenceRecord, Sequence::{normalize, strip_returns,
//   reverse_complement, canonical_kmers, kmers, bit_kmers}, QualitySequence::quality_mask, minimizer,
//   bitkmer::{reverse_complement, canonical, minimizer}}              (reference src/lib.rs:56-57, src/sequence.rs:156-303)
// ports line by line (see examples/stdin_pipe.cpp, the reference's examples/stdin_pipe.rs).  Where the reference
// panics (k = 0 ...) or returns Err(ParseError), this throws needletail::Error.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <tuple>
#include <vector>

#include "needletail_amd.h"

namespace needletail {

using Bytes = std::basic_string<uint8_t>;
using Slice = std::basic_string_view<uint8_t>;

struct Error : std::runtime_error {
    int status, kind; uint64_t line; std::string record_id;
    Error(int st, const std::string &what, int k = 0, uint64_t ln = 0, std::string id = {})
        : std::runtime_error(what), status(st), kind(k), line(ln), record_id(std::move(id)) {}
};
inline void check(int st, const char *what) { if (st != NTK_OK) throw Error(st, std::string(what) + ": " + ntk_strerror(st)); }

// One (thread, device) engine handle; the default one is created on first use (device 0).
class Context {
public:
    explicit Context(int device = 0) { check(ntk_ctx_create(device, &h_), "ntk_ctx_create"); }
    ~Context() { ntk_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ntk_ctx *get() const { return h_; }
    static Context &global() { static Context c(0); return c; }
private:
    ntk_ctx *h_ = nullptr;
};

using BitKmer = std::pair<uint64_t, uint8_t>;  // reference src/bitkmer.rs:2-3

// Sequence trait (reference src/sequence.rs:156-253) over a borrowed byte slice.
class Sequence {
public:
    Sequence(Slice s, Context &c = Context::global()) : s_(s), c_(&c) {}
    Sequence(const Bytes &b, Context &c = Context::global()) : s_(b), c_(&c) {}
    Slice sequence() const { return s_; }

    Bytes strip_returns() const {  // src/sequence.rs:165-191
        Bytes out(s_.size(), 0); uint64_t n = 0; int borrowed = 0;
        check(ntk_strip_returns(c_->get(), s_.data(), s_.size(), out.data(), &n, &borrowed), "ntk_strip_returns");
        out.resize(n); return out;
    }
    Bytes reverse_complement() const {  // src/sequence.rs:202-208
        Bytes out(s_.size(), 0);
        check(ntk_reverse_complement(c_->get(), s_.data(), s_.size(), out.data()), "ntk_reverse_complement");
        return out;
    }
    Bytes normalize(bool iupac) const {  // src/sequence.rs:226-232
        Bytes out(s_.size(), 0); uint64_t n = 0; int changed = 0;
        check(ntk_normalize(c_->get(), s_.data(), s_.size(), iupac, out.data(), &n, &changed), "ntk_normalize");
        out.resize(n); return out;
    }
    // (position, canonical k-mer slice, is_rc): slices borrow `*this` or `reverse_complement` exactly like src/kmer.rs:121-128
    std::vector<std::tuple<size_t, Slice, bool>> canonical_kmers(uint8_t k, Slice reverse_complement) const {
        std::vector<uint64_t> pos(s_.size() ? s_.size() : 1); std::vector<uint8_t> rc(pos.size()); uint64_t n = 0;
        check(ntk_canonical_kmers(c_->get(), s_.data(), s_.size(), k, pos.data(), rc.data(), pos.size(), &n), "ntk_canonical_kmers");
        std::vector<std::tuple<size_t, Slice, bool>> out; out.reserve(n);
        const Slice r = reverse_complement;
        for (uint64_t i = 0; i < n; i++)
            out.emplace_back(pos[i], rc[i] ? r.substr(r.size() - pos[i] - k, k) : s_.substr(pos[i], k), rc[i] != 0);
        return out;
    }
    std::vector<Slice> kmers(uint8_t k) const {  // src/kmer.rs:13-41
        std::vector<Slice> out;
        for (size_t i = 0; i + k <= s_.size(); i++) out.push_back(s_.substr(i, k));
        return out;
    }
    std::vector<std::tuple<size_t, BitKmer, bool>> bit_kmers(uint8_t k, bool canonical) const {  // src/bitkmer.rs:72-109
        std::vector<uint64_t> pos(s_.size() ? s_.size() : 1), val(pos.size()); std::vector<uint8_t> rc(pos.size()); uint64_t n = 0;
        check(ntk_bit_kmers(c_->get(), s_.data(), s_.size(), k, canonical, pos.data(), val.data(), rc.data(), pos.size(), &n), "ntk_bit_kmers");
        std::vector<std::tuple<size_t, BitKmer, bool>> out; out.reserve(n);
        for (uint64_t i = 0; i < n; i++) out.emplace_back(pos[i], BitKmer{val[i], k}, rc[i] != 0);
        return out;
    }
private:
    Slice s_; Context *c_;
};

// The items of Sequence::canonical_kmers for a whole batch of records from ONE call, as two bit planes
// (ntk_canonical_kmers_batch_planes): per window start "emitted" and "is_rc".  for_each(i, buffer, rc, f) walks record i's bits and
// hands f what the reference iterator yields for it, in its order: (pos, buffer[pos..pos+k] or the rc slice, is_rc)
// (reference src/kmer.rs:114-129).
class CanonicalKmersPlanes {
public:
    // seq + offsets (n + 1 entries): record i = seq[offsets[i] .. offsets[i + 1]), e.g. a reader's own buffer - uploaded as it lies
    CanonicalKmersPlanes(const uint8_t *seq, const std::vector<uint64_t> &offsets, uint8_t k, Context &c = Context::global())
        : k_(k), offsets_(offsets), rec_bit_(offsets.size())
    {
        if (offsets.empty()) return;   // no offsets at all = zero records (n + 1 entries describe n records)
        const uint64_t n = offsets.size() - 1, cap = (offsets[n] - offsets[0]) / 16 + n + 1;
        valid16_.resize(cap); rc16_.resize(cap);
        uint64_t words = 0;
        check(ntk_canonical_kmers_batch_planes(c.get(), seq, offsets_.data(), n, k, rec_bit_.data(), valid16_.data(), rc16_.data(), cap,
                                               &words, &total_), "ntk_canonical_kmers_batch_planes");
        valid16_.resize(words); rc16_.resize(words);
    }
    uint64_t total() const { return total_; }
    template <class F> void for_each(size_t i, Slice buffer, Slice rc, F &&f) const {
        const uint64_t len = offsets_[i + 1] - offsets_[i], b0 = rec_bit_[i];
        if (len < k_) return;
        for (uint64_t p = 0; p + k_ <= len; p++) {
            const uint64_t b = b0 + p;
            if (!((valid16_[b >> 4] >> (15 - (b & 15))) & 1)) continue;
            if ((rc16_[b >> 4] >> (15 - (b & 15))) & 1) f((size_t)p, rc.substr(rc.size() - p - k_, k_), true);
            else f((size_t)p, buffer.substr(p, k_), false);
        }
    }
private:
    uint64_t k_; std::vector<uint64_t> offsets_, rec_bit_; std::vector<uint16_t> valid16_, rc16_; uint64_t total_ = 0;
};

// The items of Sequence::bit_kmers(k, canonical) for a whole batch of records from ONE call (ntk_bit_kmers_batch_planes): per window start
// "emitted" and "was_rc" as bit planes, the packed values dense (one u64 per plane position).  for_each(i, f) hands f what the reference
// iterator yields for record i, in its order: (pos, value, k, was_rc) (reference src/bitkmer.rs:97-108).
class BitKmersPlanes {
public:
    BitKmersPlanes(const uint8_t *seq, const std::vector<uint64_t> &offsets, uint8_t k, bool canonical, Context &c = Context::global())
        : k_(k), offsets_(offsets), rec_bit_(offsets.size())
    {
        if (offsets.empty()) return;   // no offsets at all = zero records
        const uint64_t n = offsets.size() - 1, cap = (offsets[n] - offsets[0]) / 16 + n + 1;
        valid16_.resize(cap); rc16_.resize(cap); values_.resize(cap * 16);
        uint64_t words = 0;
        check(ntk_bit_kmers_batch_planes(c.get(), seq, offsets_.data(), n, k, canonical ? 1 : 0, rec_bit_.data(), valid16_.data(), rc16_.data(),
                                         values_.data(), cap, &words, &total_), "ntk_bit_kmers_batch_planes");
        valid16_.resize(words); rc16_.resize(words); values_.resize(words * 16);
    }
    uint64_t total() const { return total_; }
    template <class F> void for_each(size_t i, F &&f) const {
        const uint64_t len = offsets_[i + 1] - offsets_[i], b0 = rec_bit_[i];
        if (len < k_) return;
        for (uint64_t p = 0; p + k_ <= len; p++) {
            const uint64_t b = b0 + p;
            if (!((valid16_[b >> 4] >> (15 - (b & 15))) & 1)) continue;
            f((size_t)p, values_[b], (uint8_t)k_, (bool)((rc16_[b >> 4] >> (15 - (b & 15))) & 1));
        }
    }
private:
    uint64_t k_; std::vector<uint64_t> offsets_, rec_bit_, values_; std::vector<uint16_t> valid16_, rc16_; uint64_t total_ = 0;
};

// QualitySequence (reference src/sequence.rs:273-303) for a (sequence, quality) pair.
class QualitySequence : public Sequence {
public:
    QualitySequence(Slice seq, Slice qual, Context &c = Context::global()) : Sequence(seq, c), q_(qual), cq_(&c) {}
    Slice quality() const { return q_; }
    // bases whose quality byte is below `score` become N; zip semantics: the shorter of the two lengths (src/sequence.rs:285-296)
    Bytes quality_mask(uint8_t score) const {
        const uint64_t n = sequence().size() < q_.size() ? sequence().size() : q_.size();
        Bytes out(n, 0);
        check(ntk_quality_mask(cq_->get(), sequence().data(), q_.data(), n, score, out.data()), "ntk_quality_mask");
        return out;
    }
private:
    Slice q_; Context *cq_;
};

// quality::decode_phred (reference src/quality.rs:3-28): characters minus the offset; a character below it throws
enum class PhredEncoding { Phred33, Phred64 };
inline std::vector<uint8_t> decode_phred(Slice qual, PhredEncoding enc) {
    const uint8_t offset = enc == PhredEncoding::Phred33 ? '!' : '@';
    std::vector<uint8_t> scores; scores.reserve(qual.size());
    for (uint8_t q : qual) {
        if (q < offset) throw Error(NTK_ERR_BAD_ARG, "PhredOffsetError: quality " + std::to_string(q) + " is below the offset " + std::to_string(offset));
        scores.push_back((uint8_t)(q - offset));
    }
    return scores;
}

// sequence::minimizer (reference src/sequence.rs:139-152) and the bitkmer free functions (reference src/bitkmer.rs:112-162)
inline Bytes minimizer(Slice seq, size_t length, Context &c = Context::global()) {
    Bytes out(length, 0);
    check(ntk_minimizer(c.get(), seq.data(), seq.size(), (uint32_t)length, out.data()), "ntk_minimizer");
    return out;
}
// ... for every record of a reader batch in one call (record r = seq[offsets[r] .. offsets[r+1])); a record shorter than `length` throws
// (the reference panics there).  pos / is_rc (optional): the winning window's start on its strand's string, and its strand.
inline std::vector<Bytes> minimizer_batch(Slice seq, const std::vector<uint64_t> &offsets, size_t length, Context &c = Context::global(),
                                          std::vector<uint64_t> *pos = nullptr, std::vector<uint8_t> *is_rc = nullptr) {
    const size_t n = offsets.empty() ? 0 : offsets.size() - 1;
    Bytes flat(n * length, 0);
    if (pos) pos->assign(n, 0);
    if (is_rc) is_rc->assign(n, 0);
    uint64_t bad = 0;
    if (n) check(ntk_minimizer_batch(c.get(), seq.data(), offsets.data(), n, (uint32_t)length, flat.data(), pos ? pos->data() : nullptr,
                                     is_rc ? is_rc->data() : nullptr, &bad), "ntk_minimizer_batch");
    std::vector<Bytes> out(n);
    for (size_t r = 0; r < n; r++) out[r].assign(flat.begin() + r * length, flat.begin() + (r + 1) * length);
    return out;
}
// sequence::canonical (reference src/sequence.rs:110-134)
inline Bytes canonical(Slice seq, Context &c = Context::global()) {
    Bytes out(seq.size(), 0);
    check(ntk_canonical(c.get(), seq.data(), seq.size(), out.data(), nullptr), "ntk_canonical");
    return out;
}
// mask_header_tabs (reference src/parser/record.rs:188-194): tabs -> '|', nullopt when there is none
inline std::optional<Bytes> mask_header_tabs(Slice id) {
    if (id.find((uint8_t)'\t') == Slice::npos) return std::nullopt;
    Bytes out(id); for (auto &ch : out) if (ch == '\t') ch = '|';
    return out;
}
namespace bitkmer {
inline BitKmer reverse_complement(BitKmer kmer, Context &c = Context::global()) {
    uint64_t out = 0;
    check(ntk_bit_canonical(c.get(), &kmer.first, 1, kmer.second, 0, &out, nullptr), "ntk_bit_canonical");
    return BitKmer{out, kmer.second};
}
inline std::pair<BitKmer, bool> canonical(BitKmer kmer, Context &c = Context::global()) {
    uint64_t out = 0; uint8_t rc = 0;
    check(ntk_bit_canonical(c.get(), &kmer.first, 1, kmer.second, 1, &out, &rc), "ntk_bit_canonical");
    return {BitKmer{out, kmer.second}, rc != 0};
}
inline BitKmer minimizer(BitKmer kmer, uint8_t minmer_size, Context &c = Context::global()) {
    uint64_t out = 0;
    check(ntk_bit_minimizers(c.get(), &kmer.first, 1, kmer.second, minmer_size, &out), "ntk_bit_minimizers");
    return BitKmer{out, kmer.second};   // the reference returns (lowest, kmer.1): the ORIGINAL k (src/bitkmer.rs:146-162)
}
}  // namespace bitkmer

// Position / LineEnding (reference src/parser/utils.rs:50-104)
struct Position { uint64_t line_ = 0, byte_ = 0; uint64_t line() const { return line_; } uint64_t byte() const { return byte_; } };
enum class LineEnding { Windows, Unix };

// write_fasta / write_fastq (reference src/parser/record.rs:207-247)
inline void write_fasta(Slice id, Slice seq, std::ostream &w, LineEnding le = LineEnding::Unix) {
    const char *e = le == LineEnding::Windows ? "\r\n" : "\n";
    w << '>'; w.write(reinterpret_cast<const char *>(id.data()), (std::streamsize)id.size()); w << e;
    w.write(reinterpret_cast<const char *>(seq.data()), (std::streamsize)seq.size()); w << e;
}
inline void write_fastq(Slice id, Slice seq, std::optional<Slice> qual, std::ostream &w, LineEnding le = LineEnding::Unix) {
    const char *e = le == LineEnding::Windows ? "\r\n" : "\n";
    w << '@'; w.write(reinterpret_cast<const char *>(id.data()), (std::streamsize)id.size()); w << e;
    w.write(reinterpret_cast<const char *>(seq.data()), (std::streamsize)seq.size()); w << e << '+' << e;
    if (qual) w.write(reinterpret_cast<const char *>(qual->data()), (std::streamsize)qual->size());
    else w << std::string(seq.size(), 'I');   // no qualities: written as "good" (record.rs:237-244)
    w << e;
}

// SequenceRecord (reference src/parser/record.rs:21-179): a view valid until the next FastxReader::next().
struct SequenceRecord {
    Slice id_, raw_seq_, qual_; bool has_qual = false; uint64_t line = 0, bases = 0, byte = 0; int fmt = 0, ending = 1;
    Slice id() const { return id_; }
    Slice raw_seq() const { return raw_seq_; }
    Slice sequence() const { return raw_seq_; }  // impl Sequence for SequenceRecord (src/parser/record.rs:181-185)
    std::optional<Slice> qual() const { return has_qual ? std::optional<Slice>(qual_) : std::nullopt; }
    size_t num_bases() const { return bases; }
    uint64_t start_line_number() const { return line; }
    Position position() const { return Position{line, byte}; }           // src/parser/record.rs:147-149
    LineEnding line_ending() const { return ending == 2 ? LineEnding::Windows : LineEnding::Unix; }  // :152-154
    Bytes normalize(bool iupac) const { return Sequence(raw_seq_).normalize(iupac); }
    void write(std::ostream &w, std::optional<LineEnding> forced = std::nullopt) const {  // src/parser/record.rs:156-179
        const LineEnding le = forced ? *forced : line_ending();
        if (has_qual) write_fastq(id_, raw_seq_, qual_, w, le); else write_fasta(id_, raw_seq_, w, le);
    }
};

// FastxReader (reference src/parser/utils.rs:119-130)
class FastxReader {
public:
    explicit FastxReader(ntk_reader *h) : h_(h) {}
    ~FastxReader() { if (h_) ntk_reader_close(h_); }
    FastxReader(FastxReader &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    FastxReader(const FastxReader &) = delete;
    std::optional<SequenceRecord> next() {
        ntk_record r;
        const int st = ntk_reader_next(h_, &r);
        if (st == NTK_EOF) return std::nullopt;
        if (st != NTK_OK) throw_parse(st);
        SequenceRecord rec;
        rec.id_ = Slice(r.id, r.id_len); rec.raw_seq_ = Slice(r.seq, r.seq_len);
        rec.has_qual = r.qual != nullptr; if (r.qual) rec.qual_ = Slice(r.qual, r.qual_len);
        rec.line = r.line; rec.bases = r.num_bases; rec.fmt = (int)r.format; rec.byte = r.byte; rec.ending = (int)r.line_ending;
        return rec;
    }
    Position position() const {                                        // src/parser/utils.rs:125-126
        Position p; ntk_reader_position(h_, &p.line_, &p.byte_, nullptr); return p;
    }
    std::optional<LineEnding> line_ending() const {                    // src/parser/utils.rs:127-130
        int e = 0; ntk_reader_position(h_, nullptr, nullptr, &e);
        if (!e) return std::nullopt;
        return e == 2 ? LineEnding::Windows : LineEnding::Unix;
    }
    ntk_reader *get() const { return h_; }
    [[noreturn]] void throw_parse(int st) const {
        int kind = 0; uint64_t line = 0; char msg[512] = {0}, id[256] = {0};
        ntk_reader_error(h_, &kind, &line, msg, sizeof(msg), id, sizeof(id));
        throw Error(st, msg, kind, line, id);
    }
private:
    ntk_reader *h_;
};

inline FastxReader parse_fastx_file(const std::string &path) {  // reference src/parser/mod.rs:161
    ntk_reader *h = nullptr;
    const int st = ntk_reader_open_file(path.c_str(), &h);
    FastxReader rd(h);
    if (st != NTK_OK) { if (h) rd.throw_parse(st); throw Error(st, ntk_strerror(st)); }
    return rd;
}
inline FastxReader parse_fastx_stdin() { return parse_fastx_file("-"); }  // reference src/parser/mod.rs:154-159
inline FastxReader parse_fastx_reader(const uint8_t *data, uint64_t n) {  // reference src/parser/mod.rs:85 over a byte slice
    ntk_reader *h = nullptr;
    const int st = ntk_reader_open_memory(data, n, &h);
    FastxReader rd(h);
    if (st != NTK_OK) { if (h) rd.throw_parse(st); throw Error(st, ntk_strerror(st)); }
    return rd;
}

}  // namespace needletail
