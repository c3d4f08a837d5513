// mirror_check.cpp — exercises the parts of the C++ mirror (include/needletail_amd.hpp) that the ported example does not:
// QualitySequence::quality_mask, minimizer, bitkmer::{canonical, minimizer}; tests/test_abi.py compares its output with
// the reference's unit-test literals (src/sequence.rs:363-374) and the oracle.
#include "needletail_amd.hpp"
#include <cstdio>
#include <vector>
using namespace needletail;
int main() {
    const uint8_t s[] = "AGCT", q[] = "AAA0";
    QualitySequence qs(Slice(s, 4), Slice(q, 4));
    Bytes m = qs.quality_mask('5');                       // reference src/sequence.rs:369-374: b"AGCN"
    printf("%.*s\n", (int)m.size(), (const char *)m.data());
    const uint8_t t[] = "ATTTCG";
    Bytes mm = minimizer(Slice(t, 6), 3);                 // reference src/sequence.rs:363-367: b"AAA"
    printf("%.*s\n", (int)mm.size(), (const char *)mm.data());
    auto c = bitkmer::canonical(BitKmer{0xE4, 4});       // TGCA? value check below vs oracle in python
    printf("%llu %d\n", (unsigned long long)c.first.first, (int)c.second);
    auto mi = bitkmer::minimizer(BitKmer{0x1B, 4}, 2);
    printf("%llu\n", (unsigned long long)mi.first);
    const uint8_t ph[] = "#</</BBFFFBF<";                 // reference src/quality.rs:35-40: 2 27 14 27 14 33 33 37 37 37 33 37 27
    for (uint8_t v : decode_phred(Slice(ph, 13), PhredEncoding::Phred33)) printf("%d ", (int)v);
    printf("\n");
    // the batched byte-path items as bit planes (CanonicalKmersPlanes), walked per record: three records back to back in one buffer,
    // the literals of reference src/kmer.rs:182-227 among them ("ACGT" k = 2: every window is its own reverse complement or sorts after it)
    const uint8_t buf[] = "ACGTAGTCGTCAnACGTACGTN";
    const std::vector<uint64_t> offs = {0, 4, 12, 22};
    CanonicalKmersPlanes pl(buf, offs, 2);
    printf("planes %llu", (unsigned long long)pl.total());
    for (size_t i = 0; i < 3; i++) {
        const Slice rec(buf + offs[i], offs[i + 1] - offs[i]);
        const Bytes rc = Sequence(rec).reverse_complement();
        pl.for_each(i, rec, Slice(rc.data(), rc.size()), [&](size_t pos, Slice kmer, bool is_rc) {
            printf(" %zu:%zu:%.*s:%d", i, pos, (int)kmer.size(), (const char *)kmer.data(), (int)is_rc);
        });
    }
    printf("\n");
    // sequence::minimizer for a reader batch in one call: the reference's literal (src/sequence.rs:363-367: "ATTTCG", 3 -> "AAA") and two more
    const uint8_t mb[] = "ATTTCGACGTTTGGCA";
    const std::vector<uint64_t> moffs = {0, 6, 10, 16};
    std::vector<uint64_t> mpos; std::vector<uint8_t> mrc;
    const std::vector<Bytes> mins = minimizer_batch(Slice(mb, 16), moffs, 3, Context::global(), &mpos, &mrc);
    printf("minimizer_batch");
    for (size_t i = 0; i < mins.size(); i++) printf(" %.*s:%llu:%d", (int)mins[i].size(), (const char *)mins[i].data(), (unsigned long long)mpos[i], (int)mrc[i]);
    printf("\n");
    // Sequence::bit_kmers(k, canonical) for the same three records as planes + dense values (BitKmersPlanes), walked per record
    BitKmersPlanes bp(buf, offs, 3, true);
    printf("bit_planes %llu", (unsigned long long)bp.total());
    for (size_t i = 0; i < 3; i++)
        bp.for_each(i, [&](size_t pos, uint64_t value, uint8_t k, bool was_rc) { printf(" %zu:%zu:%llu:%d:%d", i, pos, (unsigned long long)value, (int)k, (int)was_rc); });
    printf("\n");
    return 0;
}
