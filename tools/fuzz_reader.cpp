// fuzz_reader.cpp — ASan/UBSan run of the CPU reader over random valid and mutated FASTA/FASTQ texts (plain, gzip, truncated gzip).
// Built and run by tests/test_parser.py::test_reader_under_sanitizers:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -mavx2 tools/fuzz_reader.cpp -lz -ldl
#include "../needletail_amd/csrc/ntk_fastx.cpp"
#include "../needletail_amd/csrc/ntk_fastx_codecs.cpp"
#include <random>
#include <zlib.h>
int main() {
    std::mt19937_64 rng(12345);
    uint64_t total = 0, errors = 0;
    for (int it = 0; it < 6000; it++) {
        std::string t;
        const bool fq = rng() & 1, crlf = (rng() % 4) == 0;
        const char *nl = crlf ? "\r\n" : "\n";
        const int nrec = 1 + rng() % 12;
        for (int r = 0; r < nrec; r++) {
            const size_t L = (rng() % 50 == 0) ? 70000 + rng() % 80000 : rng() % 300;
            std::string s(L, 'A');
            for (auto &c : s) c = "ACGTNacgt"[rng() % 9];
            if (fq) { t += "@r" + std::to_string(r) + " d" + nl + s + nl + "+" + nl + std::string(L, 'I') + nl; }
            else { t += ">r" + std::to_string(r) + nl; const size_t w = 1 + rng() % 100; for (size_t i = 0; i < L; i += w) { t += s.substr(i, w); t += nl; } }
        }
        const int mut = rng() % 4;   // 0: valid; else corrupt
        if (mut == 1 && !t.empty()) t.resize(rng() % t.size());
        if (mut == 2) for (int k = 0; k < 3 && !t.empty(); k++) t[rng() % t.size()] = "\n>@+\r x"[rng() % 7];
        if (mut == 3 && t.size() > 10) t.erase(rng() % (t.size() - 5), 1 + rng() % 5);
        std::string blob = t;
        if (rng() % 3 == 0 && !t.empty()) {   // gzip it (sometimes truncated)
            uLongf cap = compressBound(t.size()) + 64; std::string z(cap, 0);
            z_stream zs{}; deflateInit2(&zs, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = (Bytef *)t.data(); zs.avail_in = (uInt)t.size(); zs.next_out = (Bytef *)z.data(); zs.avail_out = (uInt)cap;
            deflate(&zs, Z_FINISH); z.resize(cap - zs.avail_out); deflateEnd(&zs);
            if (rng() % 5 == 0 && z.size() > 4) z.resize(z.size() - 1 - rng() % 4);
            blob = z;
        }
        ntk::FastxReader rd;
        if (!rd.open_memory((const uint8_t *)blob.data(), blob.size())) { errors++; continue; }
        ntk::FastxRecord rec; int rc;
        while ((rc = rd.next(&rec)) == 1) { total += rec.seq_len + rec.id_len + (rec.qual ? rec.qual_len : 0); volatile uint8_t x = rec.seq_len ? rec.seq[rec.seq_len - 1] : 0; (void)x; }
        if (rc < 0) errors++;
    }
    printf("fuzz ok: %llu bytes seen, %llu inputs rejected\n", (unsigned long long)total, (unsigned long long)errors);
}
