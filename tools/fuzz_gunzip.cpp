// fuzz_gunzip.cpp - ASan / UBSan run of the speculative parallel inflater (ntk_pgzip.cpp) over valid, truncated, bit-flipped and spliced gzip
// streams at several thread counts: a valid stream must come back byte for byte, anything else must end in an error or in output whose
// CRC-32 / ISIZE still check - never in a crash, an out-of-bounds access or a hang.  Built and run by tests/test_gunzip.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined tools/fuzz_gunzip.cpp -lz -ldl -lpthread [iterations]
#include "../needletail_amd/csrc/ntk_pgzip.cpp"
#include <random>
#include <thread>
#include <string>

static std::string deflate_gz(const std::string &t, int level, int strategy, size_t flush_every, std::mt19937_64 &rng)
{
    std::string z;
    char buf[1 << 16];
    z_stream zs{};
    deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, strategy);
    size_t pos = 0;
    do {
        const size_t n = flush_every ? std::min(flush_every, t.size() - pos) : t.size() - pos;
        const int how = pos + n == t.size() ? Z_FINISH : (rng() & 1 ? Z_FULL_FLUSH : Z_SYNC_FLUSH);
        zs.next_in = (Bytef *)t.data() + pos; zs.avail_in = (uInt)n;
        int r;
        do {   // until this piece is consumed and flushed (an output buffer that fills up is not the end of the piece)
            zs.next_out = (Bytef *)buf; zs.avail_out = sizeof buf;
            r = deflate(&zs, how);
            z.append(buf, sizeof buf - zs.avail_out);
        } while (zs.avail_out == 0 || (how == Z_FINISH && r != Z_STREAM_END));
        pos += n;
    } while (pos < t.size());
    deflateEnd(&zs);
    return z;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const int only = argc > 2 ? atoi(argv[2]) : -1;          // replay: inflate iteration `only` alone (the generator is consumed as in the full run)
    const char *dump = argc > 3 ? argv[3] : nullptr;         // ... and write its stream to this file
    std::mt19937_64 rng(20260927);
    uint64_t ok = 0, rejected = 0, survived = 0, bytes = 0;
    for (int it = 0; it < iters; it++) {
        // text: FASTQ-like records, runs, random bytes, long-range repeats
        std::string t;
        const size_t target = (rng() % 8 == 0) ? 600000 + rng() % 2500000 : rng() % 200000;
        while (t.size() < target) {
            switch (rng() % 5) {
            case 0: { std::string s(100 + rng() % 100, 'A'); for (auto &c : s) c = "ACGT"[rng() % 4]; t += "@r" + std::to_string(rng() % 100000) + "\n" + s + "\n+\n" + std::string(s.size(), 'I') + "\n"; break; }
            case 1: t += std::string(1 + rng() % 70000, (char)(rng() % 256)); break;
            case 2: { const size_t n = 1 + rng() % 30000; for (size_t i = 0; i < n; i++) t += (char)(rng() % 256); break; }
            case 3: if (!t.empty()) { const size_t a = rng() % t.size(); t += t.substr(a, 1 + rng() % 100000); } break;
            default: { std::string s(1 + rng() % 400, 'A'); for (auto &c : s) c = "ACGTN"[rng() % 5]; t += s; }
            }
        }
        std::string z;
        const int members = 1 + (rng() % 4 == 0 ? (int)(rng() % 3) : 0);
        std::string whole;
        for (int m = 0; m < members; m++) {
            const std::string part = members == 1 ? t : t.substr(t.size() * m / members, t.size() * (m + 1) / members - t.size() * m / members);
            const int strat[] = {Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
            z += deflate_gz(part, (int)(rng() % 10), strat[rng() % 6], rng() % 3 == 0 ? 1 + rng() % 50000 : 0, rng);
            whole += part;
        }
        const int mut = (int)(rng() % 5);   // 0, 1: valid; 2: truncated; 3: bit flips; 4: a piece cut out or duplicated
        if (mut == 2 && !z.empty()) z.resize(rng() % z.size());
        if (mut == 3) for (int k = 0; k < 1 + (int)(rng() % 3) && !z.empty(); k++) z[rng() % z.size()] ^= (char)(1u << (rng() % 8));
        if (mut == 4 && z.size() > 40) {
            const size_t a = 10 + rng() % (z.size() - 20), n = 1 + rng() % std::min<size_t>(z.size() - a, 5000);
            if (rng() & 1) z.erase(a, n); else z.insert(a, z.substr(a, n));
        }
        const uint32_t threads = (uint32_t[]){1, 2, 3, 5, 8}[rng() % 5];
        if (only >= 0 && it != only) continue;
        if (dump) { FILE *f = fopen(dump, "wb"); if (f) { fwrite(z.data(), 1, z.size(), f); fclose(f); } fprintf(stderr, "it %d: %zu bytes of gzip, %zu of text, mutation %d, %d member(s), %u threads\n", it, z.size(), whole.size(), mut, members, threads); }
        uint8_t *out = nullptr; uint64_t out_n = 0; ntk::PgzStats st;
        int rc;
        const bool streamed = rng() % 3 == 0;   // every third stream through the progressive form, read by a consumer thread as it grows
        ntk::PgzStream S;
        std::string taken;
        if (streamed) {
            S.window = (uint64_t)(1 + rng() % 6) << 18;   // 256 KiB .. 1.5 MiB: the decoder waits for the consumer all the time
            const uint64_t step = 1 + rng() % 300000;
            const bool give_up = mut > 1 && rng() % 8 == 0;   // sometimes the consumer walks away in the middle
            std::thread consumer([&] {
                std::unique_lock<std::mutex> lk(S.mu);
                uint64_t cur = 0;
                for (;;) {
                    if (S.ready > cur) {
                        const uint64_t to = std::min(S.ready, cur + step);
                        taken.append((const char *)S.base + cur, (size_t)(to - cur));
                        if (to >= 4096) madvise((void *)S.base, (size_t)((to - 4096) & ~(uint64_t)4095), MADV_DONTNEED);   // what was read may vanish
                        cur = to; S.consumed = cur;
                        if (give_up && cur > step) { S.cancel = true; S.cv.notify_all(); return; }
                        S.cv.notify_all();
                        continue;
                    }
                    if (S.finished) return;
                    S.cv.wait(lk);
                }
            });
            rc = ntk::pgz_inflate_stream((const uint8_t *)z.data(), z.size(), threads, (uint64_t)64 << 20, &S, &st);
            consumer.join();
            if (rc == 4) rc = 1;   // cancelled by the consumer: counts as rejected
            if (rc == 0) {   // hand the text over in the other form's shape
                out_n = taken.size();
                out = ntk::pgz_alloc(out_n);
                if (out_n) memcpy(out, taken.data(), out_n);
                if (S.ready != out_n) { fprintf(stderr, "stream: ready %llu, taken %llu (it %d)\n", (unsigned long long)S.ready, (unsigned long long)out_n, it); return 1; }
            }
            ntk::pgz_stream_release(&S);
        } else
            rc = ntk::pgz_inflate((const uint8_t *)z.data(), z.size(), threads, (uint64_t)64 << 20, &out, &out_n, &st);
        bytes += z.size();
        if (rc == 2 && whole.size() <= ((uint64_t)64 << 20) && mut == 2) {   // a truncated stream is an error of the data, never "too large"
            fprintf(stderr, "truncated stream reported as too large: it %d threads %u\n", it, threads); return 1;
        }
        if (rc == 0) {
            const bool same = out_n == whole.size() && (out_n == 0 || memcmp(out, whole.data(), out_n) == 0);
            if (mut <= 1 && !same) { fprintf(stderr, "MISMATCH it %d threads %u (valid stream)\n", it, threads); return 1; }
            if (mut > 1 && !same) survived++;   // a mutation the checksums cannot see (e.g. inside a stored block's padding): allowed, counted
            if (out_n) { volatile uint8_t x = out[out_n - 1]; (void)x; }
            ntk::pgz_free(out, out_n);
            ok++;
        } else {
            if (mut <= 1) { fprintf(stderr, "valid stream rejected: it %d rc %d threads %u\n", it, rc, threads); return 1; }
            if (out) { fprintf(stderr, "error with an output buffer: it %d\n", it); return 1; }
            rejected++;
        }
    }
    printf("fuzz_gunzip ok: %d streams (%llu MB), %llu inflated, %llu rejected, %llu mutated streams whose checksums still held\n", iters,
           (unsigned long long)(bytes >> 20), (unsigned long long)ok, (unsigned long long)rejected, (unsigned long long)survived);
    return 0;
}
