"""The sliding-minimum scheme of the generic fused minimizer kernel (needletail_amd/csrc/ntk_kernels.hpp, minimizer_scan_kernel), restated
in numpy and checked against a brute-force window minimum: keys (value << 1) | strand flag, a minimum that prefers its LEFT operand on ties
and ignores the flag (take L <=> key_L <= key_R | 1), and the binary decomposition of w with FIXED shifts
    M_1 = key;  M_2q[x] = min(M_q[x - q], M_q[x]);  A'[x] = min(A[x - q], M_q[x]) for the set bits q of w, low to high.
No GPU needed: this pins the ALGORITHM (leftmost tie rule, the flag never deciding, every w); the device code is checked by the gpu tests."""
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
            for x in range(w - 1, n):   # (positions whose window reaches before the start hold the fill value: the kernel's halo lanes)
                win = values[x - w + 1: x + 1]
                j = int(np.argmin(win))          # numpy: the FIRST minimum = the leftmost
                want = (int(win[j]) << 1) | int(flags[x - w + 1 + j])
                assert int(got[x]) == want, (trial, w, x)


def test_the_flag_never_decides():
    # equal values, the right one carries flag 0 and the left one flag 1: the left one must still win
    key = np.array([(7 << 1) | 1, (7 << 1) | 0, (9 << 1) | 0], dtype=np.uint64)
    assert int(sliding_min_binary(key, 2)[1]) == (7 << 1) | 1
    assert int(sliding_min_binary(key, 3)[2]) == (7 << 1) | 1
