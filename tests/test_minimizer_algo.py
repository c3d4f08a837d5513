"""The sliding-minimum scheme of the generic fused minimizer kernel (needletail_amd/csrc/ntk_kernels.hpp, minimizer_scan_kernel), restated
in numpy and checked against a brute-force window minimum: keys (value << 1) | strand flag, a minimum that prefers its LEFT operand on ties
and ignores the flag (take L <=> key_L <= key_R | 1), doubling with FIXED shifts and two overlapping power-of-two windows for w
    M_1 = key;  M_2q[x] = min(M_q[x - q], M_q[x]) while 2q <= w;  window[x] = min(M_q[x - (w - q)], M_q[x])
(the first version of the kernel: A'[x] = min(A[x - q], M_q[x]) for the set bits q of w, low to high - restated below as well);
for k <= 25 the kernel's keys are bit 62 | value << 11 | position << 1 | flag under a plain minimum (v_min_f64 on the bit patterns).
No GPU needed: this pins the ALGORITHM (leftmost tie rule, the flag never deciding, every w); the kernel's own per-lane source runs under the
64-lane emulator in test_tile_logic_emu.py, the device code in the gpu tests."""
import numpy as np


def min_left(l, r):
    return np.where(l <= (r | np.uint64(1)), l, r)


def shifted(a, q, fill):
    out = np.full_like(a, fill)
    out[q:] = a[:-q] if q else a
    return out


def sliding_min_binary(key, w):
    INF = np.uint64(0xFFFFFFFFFFFFFFFF)
    m = key.copy()
    acc = None
    q = 1
    while q <= w:
        if w & q:
            acc = m.copy() if acc is None else min_left(shifted(acc, q, INF), m)
        if w >= 2 * q:
            m = min_left(shifted(m, q, INF), m)
        q *= 2
    return acc


def sliding_min_overlap(key, w, minimum=min_left):
    INF = np.uint64(0xFFFFFFFFFFFFFFFF)
    m = key.copy()
    q = 1
    while 2 * q <= w:
        m = minimum(shifted(m, q, INF), m)
        q *= 2
    return m.copy() if w == q else minimum(shifted(m, w - q, INF), m)


def test_binary_decomposition_is_the_leftmost_window_minimum():
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(60, 400))
        # few distinct values: ties everywhere, with both strand flags on equal values
        values = rng.integers(0, int(rng.choice([3, 8, 1 << 20, 1 << 61])), n).astype(np.uint64)
        flags = rng.integers(0, 2, n).astype(np.uint64)
        key = (values << np.uint64(1)) | flags
        for w in list(range(1, 50)) + [int(rng.integers(50, 60))]:
            got = sliding_min_binary(key, w)
            got2 = sliding_min_overlap(key, w)
            # the f64 keys: unique per position, ordered by (value, position); a plain minimum (positive normal doubles order as integers)
            if int(values.max()) < (1 << 50):
                kf = (np.uint64(1) << np.uint64(62)) | (values << np.uint64(11)) | ((np.arange(n, dtype=np.uint64) % np.uint64(1024)) << np.uint64(1)) | flags
                assert np.array_equal(np.minimum(kf[1:], kf[:-1]), np.minimum(kf[1:].view(np.float64), kf[:-1].view(np.float64)).view(np.uint64))
                gf = sliding_min_overlap(kf, w, np.minimum) if n <= 1024 else None
            else:
                gf = None
            for x in range(w - 1, n):   # (positions whose window reaches before the start hold the fill value: the kernel's halo lanes)
                win = values[x - w + 1: x + 1]
                j = int(np.argmin(win))          # numpy: the FIRST minimum = the leftmost
                want = (int(win[j]) << 1) | int(flags[x - w + 1 + j])
                assert int(got[x]) == want and int(got2[x]) == want, (trial, w, x)
                if gf is not None:
                    g = int(gf[x])
                    assert (((g & ~(1 << 62)) >> 11) << 1) | (g & 1) == want, (trial, w, x, "f64 keys")


def test_the_flag_never_decides():
    # equal values, the right one carries flag 0 and the left one flag 1: the left one must still win
    key = np.array([(7 << 1) | 1, (7 << 1) | 0, (9 << 1) | 0], dtype=np.uint64)
    for f in (sliding_min_binary, sliding_min_overlap):
        assert int(f(key, 2)[1]) == (7 << 1) | 1
        assert int(f(key, 3)[2]) == (7 << 1) | 1
