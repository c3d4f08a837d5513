/* abi_smoke.c - the header is plain C: compiled with `gcc -std=c11 -pedantic -Wall -Werror` against
 * include/needletail_amd.h and linked with the library (tests/test_abi.py).  Runs without a GPU: it checks the status
 * strings, the ABI version, struct layouts the bindings rely on, and that context creation fails LOUDLY (no fallback)
 * when no gfx950 device is present; with a GPU it runs one tiny reduce and one single-rank RCCL all-reduce. */
#include <stdio.h>
#include <string.h>
#include <stddef.h>
#include "needletail_amd.h"

_Static_assert(sizeof(ntk_params) == 16, "ntk_params is four u32");
_Static_assert(offsetof(ntk_result, hist) == 40, "five u64 scalars precede the histogram");
_Static_assert(sizeof(ntk_result) == 40 + 8 * NTK_HIST_BINS, "ntk_result layout");
_Static_assert(NTK_ACC_WORDS == 8 + NTK_HIST_BINS + 64, "accumulator layout");
_Static_assert(NTK_COMM_ID_BYTES == 128, "ncclUniqueId");

int main(void)
{
    if (ntk_abi_version() != NTK_ABI_VERSION) { printf("abi version mismatch\n"); return 1; }
    if (strcmp(ntk_strerror(NTK_OK), "ok") != 0 || strlen(ntk_strerror(NTK_ERR_RCCL)) == 0) { printf("strerror\n"); return 1; }
    ntk_ctx *ctx = NULL;
    int rc = ntk_ctx_create(0, &ctx);
    if (rc == NTK_ERR_NO_DEVICE) { printf("abi_smoke ok (no device: %s)\n", ntk_strerror(rc)); return ctx == NULL ? 0 : 1; }
    if (rc != NTK_OK) { printf("ntk_ctx_create: %s\n", ntk_strerror(rc)); return 1; }
    /* a GPU is present: '>id1\nAGTCGTCA' of the reference's tests/test_stdin.rs, k = 4 */
    static const uint8_t seq[] = "AGTCGTCA";
    uint64_t pos[8], val[8], n = 0; uint8_t flg[8];
    rc = ntk_bit_kmers(ctx, seq, 8, 4, 1, pos, val, flg, 8, &n);
    if (rc != NTK_OK || n != 5) { printf("ntk_bit_kmers: %s, %llu items\n", ntk_strerror(rc), (unsigned long long)n); return 1; }
    const uint64_t offsets[3] = {0, 8, 8};
    uint64_t counts[2] = {9, 9}, total = 0;
    rc = ntk_bit_kmers_batch(ctx, seq, offsets, 2, 4, 1, counts, pos, val, flg, 8, &total);
    if (rc != NTK_OK || total != 5 || counts[0] != 5 || counts[1] != 0) { printf("ntk_bit_kmers_batch: %s\n", ntk_strerror(rc)); return 1; }
    ntk_comm *comm = NULL;
    ntk_ctx *ctxs[1] = {ctx};
    rc = ntk_comm_init_all(ctxs, 1, &comm);
    if (rc != NTK_OK) { printf("ntk_comm_init_all: %s (rccl %d)\n", ntk_strerror(rc), ntk_last_rccl_error()); return 1; }
    if (ntk_comm_size(comm) != 1 || ntk_allreduce_accumulators(comm) != NTK_OK) { printf("allreduce\n"); return 1; }
    ntk_result res;
    if (ntk_accum_read(ctx, &res) != NTK_OK) return 1;
    ntk_comm_destroy(comm);
    ntk_ctx_destroy(ctx);
    printf("abi_smoke ok (gpu)\n");
    return 0;
}
