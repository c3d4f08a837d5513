// mirror_check.cpp — exercises the parts of the C++ mirror (include/needletail_amd.hpp) that the ported example does not:
// QualitySequence::quality_mask, minimizer, bitkmer::{canonical, minimizer}; tests/test_abi.py compares its output with
// the reference's unit-test literals (src/sequence.rs:363-374) and the oracle.
#include "needletail_amd.hpp"
#include <cstdio>
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
    return 0;
}
