// ntk_scan2.hip - the instantiations of ntk::scan2_kernel (canonical and forward-only reduce mode for every k <= 32, the quality-masked
// builds and the fused windowed-minimizer builds) in their own translation unit: the tile loop is one long straight-line
// block, and the ILP-driven iterative scheduler (-mllvm -amdgpu-sched-strategy=iterative-ilp, see the Makefile) orders it
// 3.6 % faster than the default one - which in turn crashes the compiler on other kernels of the library, hence the split.
#include <hip/hip_runtime.h>

#define NTK_SCAN_TEMPLATES_ONLY
#include "ntk_kernels.hpp"

using namespace ntk;

namespace {
constexpr int kScan2HistBits = 14;   // LDS histogram of the sv2 builds: 64 KiB, two 768-thread blocks per CU
}

#if !defined(NTK_SCAN2_MIN_BUILDS) && !defined(NTK_SCAN2_Q_BUILDS)
const void *ntk_pick_scan2(int k, bool tie_rc, bool accept_u)
{
#define NTK_PICK_SV(KF, T, U)                                                                       \
    if (k == KF && tie_rc == T && accept_u == U) return (const void *)&scan2_kernel<KF, T, U, false, kScan2HistBits>;
#define NTK_PICK_SV4(KF) NTK_PICK_SV(KF, false, false) NTK_PICK_SV(KF, false, true) NTK_PICK_SV(KF, true, false) NTK_PICK_SV(KF, true, true)
    NTK_PICK_SV4(1) NTK_PICK_SV4(2) NTK_PICK_SV4(3) NTK_PICK_SV4(4) NTK_PICK_SV4(5) NTK_PICK_SV4(6) NTK_PICK_SV4(7) NTK_PICK_SV4(8)
    NTK_PICK_SV4(9) NTK_PICK_SV4(10) NTK_PICK_SV4(11) NTK_PICK_SV4(12) NTK_PICK_SV4(13) NTK_PICK_SV4(14) NTK_PICK_SV4(15) NTK_PICK_SV4(16)
    NTK_PICK_SV4(17) NTK_PICK_SV4(18) NTK_PICK_SV4(19) NTK_PICK_SV4(20) NTK_PICK_SV4(21) NTK_PICK_SV4(22) NTK_PICK_SV4(23) NTK_PICK_SV4(24)
    NTK_PICK_SV4(25) NTK_PICK_SV4(26) NTK_PICK_SV4(27) NTK_PICK_SV4(28) NTK_PICK_SV4(29) NTK_PICK_SV4(30) NTK_PICK_SV4(31) NTK_PICK_SV4(32)
#undef NTK_PICK_SV4
#undef NTK_PICK_SV
    return nullptr;
}

// Forward-only builds (BitNuclKmer with canonical = false; lane_tile_sv2_fwd, lane_tile_sv2w), every k.
const void *ntk_pick_scan2_fwd(int k, bool accept_u)
{
#define NTK_PICK_FWD(KF, U) if (k == KF && accept_u == U) return (const void *)&scan2_kernel<KF, false, U, false, kScan2HistBits, 0, true>;
#define NTK_PICK_FWD2(KF) NTK_PICK_FWD(KF, false) NTK_PICK_FWD(KF, true)
    NTK_PICK_FWD2(1) NTK_PICK_FWD2(2) NTK_PICK_FWD2(3) NTK_PICK_FWD2(4) NTK_PICK_FWD2(5) NTK_PICK_FWD2(6) NTK_PICK_FWD2(7) NTK_PICK_FWD2(8)
    NTK_PICK_FWD2(9) NTK_PICK_FWD2(10) NTK_PICK_FWD2(11) NTK_PICK_FWD2(12) NTK_PICK_FWD2(13) NTK_PICK_FWD2(14) NTK_PICK_FWD2(15) NTK_PICK_FWD2(16)
    NTK_PICK_FWD2(17) NTK_PICK_FWD2(18) NTK_PICK_FWD2(19) NTK_PICK_FWD2(20) NTK_PICK_FWD2(21) NTK_PICK_FWD2(22) NTK_PICK_FWD2(23) NTK_PICK_FWD2(24)
    NTK_PICK_FWD2(25) NTK_PICK_FWD2(26) NTK_PICK_FWD2(27) NTK_PICK_FWD2(28) NTK_PICK_FWD2(29) NTK_PICK_FWD2(30) NTK_PICK_FWD2(31) NTK_PICK_FWD2(32)
#undef NTK_PICK_FWD2
#undef NTK_PICK_FWD
    return nullptr;
}

#elif defined(NTK_SCAN2_Q_BUILDS)
// (compiled a third time with -DNTK_SCAN2_Q_BUILDS into ntk_scan2_q.o, same scheduler flag: the builds that read a quality stream)
// Quality-masked builds (QualitySequence::quality_mask, reference src/sequence.rs:285-296, fused into the encode): every k,
// canonical (both tie rules) and forward-only.
const void *ntk_pick_scan2_q(int k, bool canonical, bool tie_rc, bool accept_u)
{
#define NTK_PICK_Q(KF, T, U) if (canonical && k == KF && tie_rc == T && accept_u == U) return (const void *)&scan2_kernel<KF, T, U, true, kScan2HistBits>;
#define NTK_PICK_QF(KF, U) if (!canonical && k == KF && accept_u == U) return (const void *)&scan2_kernel<KF, false, U, true, kScan2HistBits, 0, true>;
#define NTK_PICK_Q6(KF) NTK_PICK_Q(KF, false, false) NTK_PICK_Q(KF, false, true) NTK_PICK_Q(KF, true, false) NTK_PICK_Q(KF, true, true) NTK_PICK_QF(KF, false) NTK_PICK_QF(KF, true)
    NTK_PICK_Q6(1) NTK_PICK_Q6(2) NTK_PICK_Q6(3) NTK_PICK_Q6(4) NTK_PICK_Q6(5) NTK_PICK_Q6(6) NTK_PICK_Q6(7) NTK_PICK_Q6(8)
    NTK_PICK_Q6(9) NTK_PICK_Q6(10) NTK_PICK_Q6(11) NTK_PICK_Q6(12) NTK_PICK_Q6(13) NTK_PICK_Q6(14) NTK_PICK_Q6(15) NTK_PICK_Q6(16)
    NTK_PICK_Q6(17) NTK_PICK_Q6(18) NTK_PICK_Q6(19) NTK_PICK_Q6(20) NTK_PICK_Q6(21) NTK_PICK_Q6(22) NTK_PICK_Q6(23) NTK_PICK_Q6(24)
    NTK_PICK_Q6(25) NTK_PICK_Q6(26) NTK_PICK_Q6(27) NTK_PICK_Q6(28) NTK_PICK_Q6(29) NTK_PICK_Q6(30) NTK_PICK_Q6(31) NTK_PICK_Q6(32)
#undef NTK_PICK_Q6
#undef NTK_PICK_QF
#undef NTK_PICK_Q
    return nullptr;
}

#else
// (compiled with -DNTK_SCAN2_MIN_BUILDS=1 / =2 and the DEFAULT scheduler into ntk_scan2_min.o / ntk_scan2_min2.o: the iterative one
// crashes the register allocator on these builds)
// Fused windowed-minimizer builds (ntk_tile.hpp lane_tile_sv2_min): every k = 15..23 x w = 5, 9..12 - the sketch
// parameters in common use ((15, 10), (19, 10), (21, 11): configs[4]) and their neighbours - plus quality-masked builds of
// (21, 11) and (15, 10); every other (k, w) takes the generic fused kernel (k <= 31, w <= 49) or the two-pass path.
#define NTK_PICK_MIN(KF, WF, T, U, Q) if (k == KF && w == WF && tie_rc == T && accept_u == U && quality == Q) return (const void *)&scan2_kernel<KF, T, U, Q, kScan2HistBits, WF>;
#define NTK_PICK_MIN4(KF, WF, Q) NTK_PICK_MIN(KF, WF, false, false, Q) NTK_PICK_MIN(KF, WF, false, true, Q) NTK_PICK_MIN(KF, WF, true, false, Q) NTK_PICK_MIN(KF, WF, true, true, Q)
#define NTK_PICK_MINW(KF) NTK_PICK_MIN4(KF, 5, false) NTK_PICK_MIN4(KF, 9, false) NTK_PICK_MIN4(KF, 10, false) NTK_PICK_MIN4(KF, 11, false) NTK_PICK_MIN4(KF, 12, false)
#if NTK_SCAN2_MIN_BUILDS == 1
const void *ntk_pick_scan2_min_a(int k, int w, bool tie_rc, bool accept_u, bool quality)
{
    NTK_PICK_MINW(15) NTK_PICK_MINW(16) NTK_PICK_MINW(17) NTK_PICK_MINW(18)
    NTK_PICK_MIN4(15, 10, true)
    return nullptr;
}
#else
const void *ntk_pick_scan2_min_b(int k, int w, bool tie_rc, bool accept_u, bool quality)
{
    NTK_PICK_MINW(19) NTK_PICK_MINW(20) NTK_PICK_MINW(21)
    NTK_PICK_MINW(22) NTK_PICK_MINW(23)   // round 5: k = 23, windows of 33 / 34 bytes ((22, 12), (23, 11), (23, 12): three halo lanes), and w = 5
    NTK_PICK_MIN4(21, 11, true)
    return nullptr;
}
#endif
#undef NTK_PICK_MINW
#undef NTK_PICK_MIN4
#undef NTK_PICK_MIN
#endif
