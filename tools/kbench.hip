// kbench.hip — standalone A/B harness for the scan kernel (no python/torch): generates the bench workload in HBM,
// times scan_kernel<2,canon,tie_rc,accept_u,reduce> with hipEvents and prints the reduced result so that variants
// (built with different -D NTK_V_* toggles / compiler flags) can be compared for speed AND equality.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../needletail_amd/csrc/ntk_kernels.hpp"

using namespace ntk;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint64_t reads = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    const uint32_t k = argc > 2 ? atoi(argv[2]) : 21;
    int blocks = argc > 3 ? atoi(argv[3]) : 512;
    const int threads = argc > 4 ? atoi(argv[4]) : 1024;
    const int iters = argc > 5 ? atoi(argv[5]) : 10;
    const char *tag = argc > 6 ? argv[6] : "kbench";
    const uint32_t L = getenv("KB_READ_LEN") ? (uint32_t)atoi(getenv("KB_READ_LEN")) : 150;   // KB_READ_LEN=10000: contigs (config 3's shape)
    const uint64_t n = reads * (L + 1);
    uint8_t *d_seq; uint32_t *d_ph; uint64_t *d_ps, *d_acc;
    CHK(hipMalloc(&d_seq, n + 4096));
    hipLaunchKernelGGL(synth_reads_kernel, dim3((unsigned)((n / 16 + 256) / 256)), dim3(256), 0, 0, 0x5EED0002ull, 0ull, reads, L, 1u, d_seq);
    CHK(hipDeviceSynchronize());
    ScanArgs a; memset(&a, 0, sizeof(a));
    scan_args_set_k(a, k);
    a.seq = d_seq; a.n_bytes = n; a.n_tiles = ((n + 15) / 16 + kTileSlots - 1) / kTileSlots; a.tile_begin = 0; a.tile_end = a.n_tiles;
    const int wpb = threads / 64;
    const uint32_t chunk = argc > 7 ? atoi(argv[7]) : 16;
    uint32_t *d_work; CHK(hipMalloc(&d_work, 65536));
    a.n_shards = argc > 8 ? atoi(argv[8]) : (blocks < 8 ? blocks : 8); a.tiles_per_shard = (uint32_t)((a.n_tiles + a.n_shards - 1) / a.n_shards);
    a.chunk_tiles = chunk; a.work_counters = d_work; a.tail_tile_rel = (uint32_t)(n / kTileStride);
#ifdef NTK_X_EXACTHALO   // (profiles/r06q) tile t emits the windows ending in [S t - 32 + (k - 1), S (t + 1) - 32 + (k - 1)) and reads up to S t + 992
    {
        const uint64_t S = (uint64_t)Sv2Geom<21>::kStride, lead = 32 - (k - 1);
        if (k != 21) { printf("NTK_X_EXACTHALO: kbench runs k = 21 only\n"); return 1; }
        a.n_tiles = (n + lead + S - 1) / S; a.tile_end = a.n_tiles;
        a.tiles_per_shard = (uint32_t)((a.n_tiles + a.n_shards - 1) / a.n_shards);
        a.tail_tile_rel = (uint32_t)(n > 992 ? (n - 992) / S : 0);
    }
#endif
    CHK(hipMalloc(&d_ph, (size_t)blocks * kHistBins * 4)); CHK(hipMalloc(&d_ps, (size_t)blocks * 32)); CHK(hipMalloc(&d_acc, (8 + kHistBins + 64) * 8));
    a.part_hist = d_ph; a.part_scalars = d_ps;
#ifdef NTK_V_CLOCKS
    uint64_t *d_dbg; CHK(hipMalloc(&d_dbg, (size_t)blocks * wpb * 32)); CHK(hipMemset(d_dbg, 0, (size_t)blocks * wpb * 32)); a.values = d_dbg;
#endif
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::vector<float> ts;
    const int warm = 300;  // ~0.2 s of load before timing: steady-state clocks
    for (int it = 0; it < iters + warm; it++) {
        CHK(hipMemsetAsync(d_acc, 0, (8 + kHistBins + 64) * 8, 0));
        CHK(hipMemsetAsync(d_work, 0, 65536, 0));
        CHK(hipEventRecord(e0, 0));
#ifdef NTK_KB_SV2
#ifndef NTK_KB_HB
#define NTK_KB_HB 12
#endif
        static const bool kb_fwd = getenv("KB_FWD") != nullptr;   // the forward-only builds (BitNuclKmer, canonical = false)
        if (kb_fwd && k == 21) hipLaunchKernelGGL((scan2_kernel<21, false, false, false, NTK_KB_HB, 0, true>), dim3(blocks), dim3(threads), 0, 0, a);
        else if (kb_fwd && k == 31) hipLaunchKernelGGL((scan2_kernel<31, false, false, false, NTK_KB_HB, 0, true>), dim3(blocks), dim3(threads), 0, 0, a);
        else if (kb_fwd && k == 16) hipLaunchKernelGGL((scan2_kernel<16, false, false, false, NTK_KB_HB, 0, true>), dim3(blocks), dim3(threads), 0, 0, a);
        else
        if (k == 21) hipLaunchKernelGGL((scan2_kernel<21, true, true, false, NTK_KB_HB>), dim3(blocks), dim3(threads), 0, 0, a);
        else if (k == 31) hipLaunchKernelGGL((scan2_kernel<31, true, true, false, NTK_KB_HB>), dim3(blocks), dim3(threads), 0, 0, a);
        else if (k == 23) hipLaunchKernelGGL((scan2_kernel<23, true, true, false, NTK_KB_HB>), dim3(blocks), dim3(threads), 0, 0, a);
        else
#endif
#ifdef NTK_KB_FIX
        if (k == 21) hipLaunchKernelGGL((scan_kernel<2, true, true, true, true, 21>), dim3(blocks), dim3(threads), 0, 0, a);
        else if (k == 31) hipLaunchKernelGGL((scan_kernel<2, true, true, true, true, 31>), dim3(blocks), dim3(threads), 0, 0, a);
        else
#endif
        if (k > 16) hipLaunchKernelGGL((scan_kernel<2, true, true, true, true>), dim3(blocks), dim3(threads), 0, 0, a);
        else hipLaunchKernelGGL((scan_kernel<1, true, true, true, true>), dim3(blocks), dim3(threads), 0, 0, a);
        CHK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(fold_kernel, dim3(kFoldBlocks), dim3(kFoldThreads), 0, 0, (const uint32_t *)d_ph, (const uint64_t *)d_ps, blocks, d_acc);
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= warm) ts.push_back(ms);
    }
    CHK(hipDeviceSynchronize());
    std::vector<uint64_t> acc(8 + kHistBins + 64);
    CHK(hipMemcpy(acc.data(), d_acc, acc.size() * 8, hipMemcpyDeviceToHost));
    uint64_t hh = 1469598103934665603ull;
    for (int i = 0; i < kHistBins; i++) { hh ^= acc[8 + i]; hh *= 1099511628211ull; }
    std::sort(ts.begin(), ts.end());
    double avg = 0; for (float t : ts) avg += t; avg /= ts.size();
    printf("%-28s k=%u grid=%dx%d  avg %.4f ms  min %.4f  med %.4f  | %.1f GB/s %.1f Gbases/s (avg) | n_total=%llu n_fwd=%llu sum=%llx xor=%llx hist=%llx\n",
           tag, k, blocks, threads, avg, ts.front(), ts[ts.size() / 2], n / (avg * 1e-3) / 1e9, reads * L / (avg * 1e-3) / 1e9,
           (unsigned long long)acc[0], (unsigned long long)acc[1], (unsigned long long)acc[3], (unsigned long long)acc[4], (unsigned long long)hh);
#ifdef NTK_V_CLOCKS
    {
        const int blocks_ = blocks; blocks = blocks * wpb;  // census entries are per wave
        std::vector<uint64_t> dbg((size_t)blocks * 4);
        CHK(hipMemcpy(dbg.data(), d_dbg, dbg.size() * 8, hipMemcpyDeviceToHost));
        uint64_t tmin = ~0ull, tmax = 0; double life = 0, cyc = 0;
        for (int b = 0; b < blocks; b++) { tmin = std::min(tmin, dbg[b * 4]); tmax = std::max(tmax, dbg[b * 4 + 1]); life += dbg[b * 4 + 1] - dbg[b * 4]; cyc += dbg[b * 4 + 2] & 0xFFFFFFFFFFull; }
        printf("   census: span %.4f ms, mean block life %.4f ms, mean concurrency %.1f blocks, mean shader clock %.3f GHz\n",
               (tmax - tmin) * 1e-5, life / blocks * 1e-5, life / (double)(tmax - tmin), cyc / (life * 10.0));
        // concurrency histogram over 20 time slices, and distinct (xcc, se, sh, cu) seen
        std::vector<int> conc(20, 0);
        std::vector<uint64_t> places;
        for (int b = 0; b < blocks; b++) {
            for (int s = 0; s < 20; s++) { uint64_t ts = tmin + (tmax - tmin) * (2 * s + 1) / 40; if (dbg[b * 4] <= ts && ts < dbg[b * 4 + 1]) conc[s]++; }
            uint64_t id = dbg[b * 4 + 3]; uint32_t hw = (uint32_t)id, xcc = (uint32_t)(id >> 32) & 0xF;
            places.push_back(((uint64_t)xcc << 16) | (hw & 0xFF00));
        }
        std::sort(places.begin(), places.end()); places.erase(std::unique(places.begin(), places.end()), places.end());
        printf("   running blocks at 20 time slices:"); for (int s = 0; s < 20; s++) printf(" %d", conc[s]); printf("\n   distinct (xcc,se,sh,cu) places: %zu; first start offsets (us):", places.size());
        std::vector<uint64_t> starts; for (int b = 0; b < blocks; b++) starts.push_back(dbg[b * 4] - tmin); std::sort(starts.begin(), starts.end());
        for (int i = 0; i < blocks; i += blocks / 16) printf(" %.1f", starts[i] * 0.01); printf("\n");
        std::vector<uint64_t> ends; for (int b = 0; b < blocks; b++) ends.push_back(dbg[b * 4 + 1] - tmin); std::sort(ends.begin(), ends.end());
        printf("   wave end-time percentiles (ms): p0 %.3f p10 %.3f p25 %.3f p50 %.3f p75 %.3f p90 %.3f p100 %.3f\n", ends[0]*1e-5, ends[blocks/10]*1e-5, ends[blocks/4]*1e-5, ends[blocks/2]*1e-5, ends[blocks*3/4]*1e-5, ends[blocks*9/10]*1e-5, ends[blocks-1]*1e-5);
        {   // tiles per wave (scan2 builds): spread, and by hardware wave slot (HW_ID bits 3:0) - is a slot starved?
            std::vector<uint32_t> tl; double by_slot[16] = {0}, end_slot[16] = {0}; int n_slot[16] = {0};
            for (int b = 0; b < blocks; b++) { uint32_t t = (uint32_t)(dbg[b * 4 + 2] >> 40); tl.push_back(t); int sl = (int)(dbg[b * 4 + 3] & 15); by_slot[sl] += t; end_slot[sl] += (dbg[b * 4 + 1] - tmin) * 1e-5; n_slot[sl]++; }
            std::sort(tl.begin(), tl.end());
            if (tl.back()) {
                printf("   tiles per wave: min %u p10 %u p50 %u p90 %u max %u\n   by wave slot (n, mean tiles, mean end ms):", tl[0], tl[blocks / 10], tl[blocks / 2], tl[blocks * 9 / 10], tl.back());
                for (int i = 0; i < 16; i++) if (n_slot[i]) printf(" [%d: %d %.0f %.3f]", i, n_slot[i], by_slot[i] / n_slot[i], end_slot[i] / n_slot[i]);
                printf("\n");
            }
        }
        blocks = blocks_;
    }
#endif
    return 0;
}
