/* abi_smoke.c - the header is plain C: compiled with `gcc -std=c11 -pedantic -Wall -Werror` against
 * include/needletail_amd.h and linked with the library (tests/test_abi.py).  Runs without a GPU: it checks the status
 * strings, the ABI version, struct layouts the bindings rely on, and that context creation fails LOUDLY (no fallback)
 * when no gfx950 device is present; with a GPU it runs one tiny reduce, and then configs[3] in miniature over EVERY visible
 * device: one ctx per device (ntk_device_count), a pinned batch reduced on each, ONE ntk_comm_init_all communicator, one
 * ntk_allreduce_accumulators, and every device must hold the sum (1 device on the builder's box, 8 on a node). */
#include <stdio.h>
#include <string.h>
#include <stddef.h>
#include "needletail_amd.h"

_Static_assert(sizeof(ntk_params) == 16, "ntk_params is four u32");
_Static_assert(offsetof(ntk_result, hist) == 40, "five u64 scalars precede the histogram");
_Static_assert(sizeof(ntk_result) == 48 + 8 * NTK_HIST_BINS, "ntk_result layout");   /* ABI 4: + n_undigested */
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
    ntk_ctx_destroy(ctx);
    /* every visible device: device d reduces (d + 1) copies of the record, all devices end up with the sum */
    enum { MAX_DEV = 64 };
    int n_dev = 0;
    if (ntk_device_count(&n_dev) != NTK_OK || n_dev < 1 || n_dev > MAX_DEV) { printf("ntk_device_count: %d\n", n_dev); return 1; }
    ntk_ctx *ctxs[MAX_DEV];
    const ntk_params par = {4, NTK_PATH_BITS_CANONICAL, NTK_PRE_NONE, 0};
    uint64_t want_total = 0;
    for (int d = 0; d < n_dev; d++) {
        ctxs[d] = NULL;
        rc = ntk_ctx_create(d, &ctxs[d]);
        if (rc != NTK_OK) { printf("ntk_ctx_create(%d): %s\n", d, ntk_strerror(rc)); return 1; }
        ntk_batch *b = NULL;
        if (ntk_accum_reset(ctxs[d]) != NTK_OK || ntk_batch_acquire(ctxs[d], 4096, 128, &b) != NTK_OK) { printf("batch_acquire on device %d\n", d); return 1; }
        for (int r = 0; r <= d; r++) if (ntk_batch_append(b, seq, 8, NTK_PRE_NONE) != NTK_OK) { printf("batch_append\n"); return 1; }
        if (ntk_batch_submit(ctxs[d], b, &par) != NTK_OK || ntk_batch_wait(ctxs[d], b) != NTK_OK) { printf("batch_submit on device %d\n", d); return 1; }
        ntk_batch_release(ctxs[d], b);
        want_total += 5ull * (uint64_t)(d + 1);
    }
    ntk_comm *comm = NULL;
    rc = ntk_comm_init_all(ctxs, n_dev, &comm);
    if (rc != NTK_OK) { printf("ntk_comm_init_all over %d device(s): %s (rccl %d)\n", n_dev, ntk_strerror(rc), ntk_last_rccl_error()); return 1; }
    if (ntk_comm_size(comm) != n_dev || ntk_allreduce_accumulators(comm) != NTK_OK) { printf("allreduce (rccl %d)\n", ntk_last_rccl_error()); return 1; }
    uint64_t xr0 = 0;
    for (int d = 0; d < n_dev; d++) {
        static ntk_result res;
        if (ntk_accum_read(ctxs[d], &res) != NTK_OK) return 1;
        if (d == 0) xr0 = res.xr;
        if (res.n_total != want_total || res.n_fwd + res.n_rc != res.n_total || res.xr != xr0) {
            printf("device %d of %d: n_total %llu, want %llu\n", d, n_dev, (unsigned long long)res.n_total, (unsigned long long)want_total);
            return 1;
        }
    }
    ntk_comm_destroy(comm);
    for (int d = 0; d < n_dev; d++) ntk_ctx_destroy(ctxs[d]);
    printf("abi_smoke: %d device(s), one communicator, n_total %llu on every device\n", n_dev, (unsigned long long)want_total);
    printf("abi_smoke ok (gpu)\n");
    return 0;
}
