// The reference's example program (reference examples/stdin_pipe.rs, README.md:17-46), ported line by line onto the
// C++ mirror: count bases and canonical AAAA 4-mers of a FASTA/FASTQ file.  Two variants are printed:
//   per-record: exactly the reference's loop (normalize -> reverse_complement -> canonical_kmers), one record at a time;
//   batched   : the same answer from ONE pipeline call (pinned batches, overlapped copies, fused scan kernel).
#include <cstdio>
#include "../include/needletail_amd.hpp"

int main(int argc, char **argv)
{
    // like the reference's example, records come from standard input when no file is named (the batched variant reads its
    // input a second time and so needs a file)
    const bool from_stdin = argc < 2;
    using namespace needletail;
    try {
        size_t n_bases = 0, n_valid_kmers = 0;
        auto reader = from_stdin ? parse_fastx_stdin() : parse_fastx_file(argv[1]);
        while (auto record = reader.next()) {
            const SequenceRecord &seqrec = *record;
            n_bases += seqrec.num_bases();
            const Bytes norm_seq = seqrec.normalize(false);
            const Bytes rc = Sequence(norm_seq).reverse_complement();
            for (auto &[pos, kmer, is_rc] : Sequence(norm_seq).canonical_kmers(4, rc)) {
                (void)pos; (void)is_rc;
                if (kmer == Slice(reinterpret_cast<const uint8_t *>("AAAA"), 4)) n_valid_kmers++;
            }
        }
        printf("There are %zu bases in your file.\n", n_bases);
        printf("There are %zu AAAAs in your file.\n", n_valid_kmers);

        if (from_stdin) return 0;
        // batched fast path: hist[0] at k = 4 is the AAAA count
        auto rd2 = parse_fastx_file(argv[1]);
        ntk_params p = {4, NTK_PATH_BYTES_CANONICAL, NTK_PRE_NORMALIZE, 0};
        uint64_t nrec = 0, nb = 0;
        check(ntk_accum_reset(Context::global().get()), "ntk_accum_reset");
        check(ntk_scan_reader(Context::global().get(), rd2.get(), &p, 1 << 22, 3, &nrec, &nb), "ntk_scan_reader");
        static ntk_result res;
        check(ntk_accum_read(Context::global().get(), &res), "ntk_accum_read");
        printf("batched: %llu records, %llu bases, %llu AAAAs\n", (unsigned long long)nrec, (unsigned long long)nb, (unsigned long long)res.hist[0]);
    } catch (const Error &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
