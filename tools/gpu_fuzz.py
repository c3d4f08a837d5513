#!/usr/bin/env python3
"""Time-budgeted differential fuzz of the device paths against the oracle (run on the GPU box through gpurun; the tests hold the
fixed-seed versions of these checks).  Inputs are STRUCTURED to hit what a streaming tile kernel can get wrong: record lengths
around the lane (16), tile (992 / 1024) and chunk boundaries, runs of N / break bytes of every length around k, breaks exactly
at lane and tile edges, lowercase / U / IUPAC / whitespace / high bytes, empty and one-byte inputs.

    python tools/gpu_fuzz.py --seconds 150 --seed 1 > profiles/<round>/gpu_fuzz.log
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import needletail_amd as nt
from needletail_amd import _lib as NL  # noqa: E402
import oracle as O  # noqa: E402  (checker)

MODES = [  # (path, pre, canonical, tie_rc, accept_u)
    (nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE, True, True, True),
    (nt.PATH_BITS_CANONICAL, nt.PRE_NONE, True, False, False),
    (nt.PATH_BITS, nt.PRE_STRIP_RETURNS, False, False, False),
    (nt.PATH_BITS_CANONICAL, nt.PRE_NORMALIZE_IUPAC, True, False, True),
    (nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE_IUPAC, True, True, True),
]
EDGE = [0, 1, 2, 15, 16, 17, 31, 32, 33, 150, 151, 991, 992, 993, 1007, 1008, 1009, 1023, 1024, 1025, 1984, 2016, 2048,
        992 * 24 - 1, 992 * 24, 992 * 24 + 1]
JUNK = np.frombuffer(b"NnUuRYKMSWBDHVrykm-.*\x00\x7f\x80\xff0@>+", dtype=np.uint8)


def make_input(rng, k):
    """bytes with structure; returns a uint8 array"""
    kind = rng.integers(0, 7)
    n = int(rng.choice(EDGE)) + int(rng.integers(-3, 4)) if rng.random() < 0.5 else int(rng.integers(0, 60000))
    n = max(n, 0)
    if kind == 6:
        n = int(rng.integers(200000, 3000000))
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    if n == 0:
        return a
    if kind == 0:      # clean
        pass
    elif kind == 1:    # records of edge lengths separated by newlines
        pos = 0
        while pos < n:
            pos += max(0, int(rng.choice(EDGE[:18])) + int(rng.integers(-2, 3)))
            if pos < n:
                a[pos] = 10
            pos += 1
    elif kind == 2:    # runs of N of lengths around k, at random and at lane / tile edges
        for _ in range(int(rng.integers(1, 40))):
            ln = max(1, k + int(rng.integers(-3, 4))) if rng.random() < 0.6 else int(rng.integers(1, 70))
            at = int(rng.integers(0, n))
            if rng.random() < 0.5:
                at = (at // 16) * 16 + int(rng.integers(-1, 2))
            if rng.random() < 0.3:
                at = (at // 992) * 992 + int(rng.integers(-2, 3))
            at = min(max(at, 0), n - 1)
            a[at:at + ln] = ord("N")
    elif kind == 3:    # mixed case, U, IUPAC, whitespace, junk
        m = rng.random(n)
        a[m < 0.15] |= 0x20
        sel = m > 0.93
        a[sel] = JUNK[rng.integers(0, len(JUNK), int(sel.sum()))]
        ws = (m > 0.90) & (m <= 0.93)
        a[ws] = np.frombuffer(b" \t\r\n", dtype=np.uint8)[rng.integers(0, 4, int(ws.sum()))]
        us = (m > 0.88) & (m <= 0.90)
        a[us] = np.frombuffer(b"Uu", dtype=np.uint8)[rng.integers(0, 2, int(us.sum()))]
    elif kind == 4:    # single breaks exactly every k-1, k, k+1 bases (no / one / two windows between breaks)
        step = max(1, k + int(rng.integers(-1, 2)))
        a[step - 1::step + 1] = ord("N") if rng.random() < 0.5 else 10
    elif kind == 5:    # low-complexity and palindromic stretches (strand ties for even k)
        unit = np.frombuffer([b"AT", b"ACGT", b"A", b"GC", b"AATT", b"ACGTACGTTGCA"][int(rng.integers(0, 6))], dtype=np.uint8)
        a = np.resize(unit, n).copy()
        for _ in range(int(rng.integers(0, 6))):
            at = int(rng.integers(0, n)); a[at:at + int(rng.integers(1, 40))] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4)]
    else:              # long, sparse breaks
        idx = rng.integers(0, n, max(1, n // 1500))
        a[idx] = ord("N")
    return a


def stats_equal(a, b):
    return all(int(a[x]) == int(b[x]) for x in ("n_total", "n_fwd", "n_rc", "sum", "xor")) and np.array_equal(a["hist"], b["hist"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minimizers-only", action="store_true", help="windowed minimizers only, byte path (ties -> rc, U) and bit path (ties -> fwd)")
    ap.add_argument("--replay-it", type=int, default=0, help="replay: consume the generator as the run with this seed did, do the device work of iteration N only")
    ap.add_argument("--replay-from", type=int, default=0, help="with --replay-it N: do the device work from this iteration on (default N)")
    ap.add_argument("--dump", default="", help="with --replay-it: write iteration N's input bytes to this file and stop (needs no GPU)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = None if args.dump else nt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    t_end = time.time() + args.seconds
    counts = {"reduce": 0, "materialize": 0, "minimizers": 0, "quality": 0, "compat_batch": 0, "compat_planes": 0}
    n_bytes = 0
    it = 0
    while time.time() < t_end or args.replay_it:
        it += 1
        if args.replay_it and it > args.replay_it:
            break
        skip = it < (args.replay_from or args.replay_it)   # replay: the generator is consumed exactly as in the original run, the device work is skipped
        k = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 11, 15, 16, 17, 18, 20, 21, 22, 23, 24, 27, 30, 31, 32]))
        path, pre, canon, tie, u = MODES[int(rng.integers(0, len(MODES)))]
        a = make_input(rng, k)
        n = len(a)
        n_bytes += n
        if not skip and not args.dump:
            t = torch.full(((n + 1023) // 1024 * 1024 + 1024,), 0x41, dtype=torch.uint8, device="cuda")
            if n:
                t[:n] = torch.from_numpy(a).cuda()
        buf = a.tobytes()
        what = rng.integers(0, 10)
        if args.minimizers_only:
            what = 6
            path, pre, canon, tie, u = MODES[int(rng.integers(0, 2))]
        # launch geometry: mostly the library's choice, sometimes forced (few / many blocks, small / large blocks: the shard, chunk
        # and work-counter arithmetic of the scan must not depend on it)
        launch = (int(rng.choice([1, 2, 7, 64, 300, 512, 1024, 2048])), int(rng.choice([0, 64, 128, 256, 512, 768, 1024]))) if rng.random() < 0.3 else (0, 0)
        tag = f"it {it} seed {args.seed} k {k} path {path} pre {pre} n {n} launch {launch}"
        if args.dump and not skip:
            open(args.dump, "wb").write(buf)
            print("dumped", tag, "what", int(what)); return 0
        if not skip:
            ctx.set_launch(*launch)
        if args.dump and skip and it + 6 > args.replay_it:
            print("before:", tag, "what", int(what))
        if skip:
            # the draws of the branch this iteration took, without its device work
            if what < 5:
                if path == nt.PATH_BYTES_CANONICAL and rng.random() < 0.3: rng.choice([0, 0, 0, 33, 48, 64, 255]); rng.integers(0, 2)
            elif what < 6 and n <= 400000:
                pass
            elif what < 8 and canon and (path, pre) in ((nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), (nt.PATH_BITS_CANONICAL, nt.PRE_NONE)):
                if rng.random() < 0.4:
                    if rng.random() < 0.5: rng.choice([1, 2, 5, 9, 10, 11, 12, 16, 33])
                    if not rng.random() < 0.5: rng.choice([15, 16, 17, 18, 19, 20, 21, 22, 23])
                else:
                    if rng.random() < 0.8: rng.integers(1, 52)
                    else: rng.choice([15, 16, 17, 31, 32, 33, 47, 48, 49, 50, 64])
                    rng.integers(1, 33)
            elif what < 9:
                rng.integers(33, 75, n, dtype=np.uint8)
                if n and rng.random() < 0.5:
                    rng.integers(0, n, max(1, n // 50))
                rng.integers(33, 76)
                if canon and path == nt.PATH_BYTES_CANONICAL and pre == nt.PRE_NORMALIZE and rng.random() < 0.4:
                    rng.integers(0, 4)
            elif n <= 300000:
                cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, int(rng.integers(0, 30)))])).astype(np.int64)
                if len(cuts) > 1:
                    if rng.random() < 0.25:
                        rng.choice([1, 2, 5, k, 31, 40])
                    else:
                        u3 = rng.random()
                        if u3 < 0.34:
                            rng.choice([k, k, int(rng.integers(1, 64)), int(rng.integers(64, 256))])
                            if rng.random() < 0.5: rng.choice([64, 97, 1000, 4096, 1 << 16])
                        elif u3 >= 0.84:
                            rng.integers(0, 2); rng.integers(0, 2)
                            if rng.random() < 0.5: rng.choice([64, 97, 1000, 4096, 1 << 16])
                        elif u3 >= 0.67:
                            rng.integers(0, 2)
            continue
        if what < 5:
            if path == nt.PATH_BYTES_CANONICAL and rng.random() < 0.3:
                kw, norm = int(rng.choice([0, 0, 0, 33, 48, 64, 255])), int(rng.integers(0, 2))
                if kw and n <= 60000:
                    # round 6: CanonicalKmers with k > 32 on the reduce face (counters + histogram of the leading six bases; no sum / xor), input
                    # normalised or not, against the literal iterator
                    pre_w = nt.PRE_NORMALIZE if norm else nt.PRE_NONE
                    ctx.accum_reset(); ctx.reduce_device(t, n, kw, path, pre_w)
                    got = ctx.accum_read()
                    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
                    nt_w = nrc_w = 0
                    hist_w = np.zeros(4096, dtype=np.uint64)
                    import re as _re
                    for r_ in (_re.split(rb"[\n\r\t ]", buf) if norm else buf.split(b"\n")):   # (on the device face the deleted class is a break: the packer drops it, reduce_device does not)
                        if norm: r_ = O.normalize(r_)[0]
                        rc_ = O.reverse_complement(r_)
                        pos_, flg_ = O.canonical_kmers_arrays(r_, rc_, kw)
                        for p_, f_ in zip(pos_.tolist(), flg_.tolist()):
                            sl = rc_[len(rc_) - p_ - kw: len(rc_) - p_] if f_ else r_[p_: p_ + kw]
                            b_ = 0
                            for ch in sl[:6]: b_ = b_ * 4 + code[ch]
                            hist_w[b_] += 1
                        nt_w += len(pos_); nrc_w += int(flg_.sum())
                    if not (got["n_total"] == nt_w == got["n_undigested"] and got["n_rc"] == nrc_w and got["n_fwd"] == nt_w - nrc_w
                            and np.array_equal(got["hist"], hist_w) and got["sum"] == 0 and got["xor"] == 0):
                        print("MISMATCH wide-k reduce", tag, "k", kw, "normalised", norm); return 1
                    counts["wide_k"] = counts.get("wide_k", 0) + 1
                    continue
                # the byte path on input that is NOT normalised: raw-byte strand compare (mixed case), against the literal chain per record
                ctx.accum_reset(); ctx.reduce_device(t, n, k, path, nt.PRE_NONE)
                if not stats_equal(ctx.accum_read(), O.reduce_records(buf.split(b"\n"), k, path, nt.PRE_NONE)):
                    print("MISMATCH raw-byte reduce", tag); return 1
                counts["raw_bytes"] = counts.get("raw_bytes", 0) + 1
                continue
            ctx.accum_reset(); ctx.reduce_device(t, n, k, path, pre)
            if not stats_equal(ctx.accum_read(), O.reduce_fused(buf, k, canon, tie, u)):
                print("MISMATCH reduce", tag); return 1
            counts["reduce"] += 1
        elif what < 6 and n <= 400000:
            vals = torch.zeros((n + 15) // 16 * 16 + 16, dtype=torch.int64, device="cuda")
            v16 = torch.zeros((n + 15) // 16 + 1, dtype=torch.int16, device="cuda"); r16 = torch.zeros_like(v16)
            ctx.materialize_device(t, n, k, path, pre, vals, v16, r16)
            hv = vals.cpu().numpy().view(np.uint64)
            e = np.arange(n)   # position e (window end byte) = bit 15 - e % 16 of word e / 16
            bits = ((v16.cpu().numpy().view(np.uint16)[e // 16] >> (15 - e % 16)) & 1).astype(bool)
            rb = ((r16.cpu().numpy().view(np.uint16)[e // 16] >> (15 - e % 16)) & 1).astype(bool)
            # the dense planes reduce to the oracle's result (position = window end byte)
            want = O.reduce_fused(buf, k, canon, tie, u)
            sel = hv[:n][bits]
            got_sum = int(np.add.reduce(sel, dtype=np.uint64)) if len(sel) else 0
            got_xor = int(np.bitwise_xor.reduce(sel)) if len(sel) else 0
            if not (int(bits.sum()) == want["n_total"] and int((bits & rb).sum()) == want["n_rc"] and got_sum == int(want["sum"])
                    and got_xor == int(want["xor"])):
                print("MISMATCH materialize", tag); return 1
            counts["materialize"] += 1
        elif what < 8 and canon and (path, pre) in ((nt.PATH_BYTES_CANONICAL, nt.PRE_NORMALIZE), (nt.PATH_BITS_CANONICAL, nt.PRE_NONE)):
            u4 = rng.random()
            if u4 < 0.4:     # the register-fused grid and its edges
                w = int(rng.choice([1, 2, 5, 9, 10, 11, 12, 16, 33])) if rng.random() < 0.5 else 11
                kk = k if rng.random() < 0.5 else int(rng.choice([15, 16, 17, 18, 19, 20, 21, 22, 23]))
            else:            # anything: the generic fused kernel (k <= 31, w <= 49), the two-pass path beyond it
                w = int(rng.integers(1, 52)) if rng.random() < 0.8 else int(rng.choice([15, 16, 17, 31, 32, 33, 47, 48, 49, 50, 64]))
                kk = int(rng.integers(1, 33))
            ctx.accum_reset(); ctx.reduce_device(t, n, kk, path, pre, w=w)
            if not stats_equal(ctx.accum_read(), O.minimizers_reduce(buf, kk, w, accept_u=u, tie_rc=tie)):
                print("MISMATCH minimizers", tag, "w", w, "kk", kk); return 1
            counts["minimizers"] += 1
        elif what < 9:
            q = rng.integers(33, 75, n, dtype=np.uint8)
            if n and rng.random() < 0.5:
                q[:] = 73; q[rng.integers(0, n, max(1, n // 50))] = 34
            cutoff = int(rng.integers(33, 76))
            tq = torch.full_like(t, 0x49)
            if n:
                tq[:n] = torch.from_numpy(q).cuda()
            masked = O.quality_mask(buf, q.tobytes(), cutoff)
            if canon and path == nt.PATH_BYTES_CANONICAL and pre == nt.PRE_NORMALIZE and rng.random() < 0.4:
                # quality-masked windowed minimizers: the fused quality builds ((21, 11), (15, 10)) and the two-pass path
                kq, wq = [(21, 11), (15, 10), (21, 10), (17, 5)][int(rng.integers(0, 4))]
                ctx.accum_reset(); ctx.reduce_device(t, n, kq, path, pre, w=wq, d_qual=tq, quality_cutoff=cutoff)
                if not stats_equal(ctx.accum_read(), O.minimizers_reduce(masked, kq, wq, True, True)):
                    print("MISMATCH quality minimizers", tag, "cutoff", cutoff, "k", kq, "w", wq); return 1
                counts["quality"] += 1
                continue
            ctx.accum_reset(); ctx.reduce_device(t, n, k, path, pre, d_qual=tq, quality_cutoff=cutoff)
            if not stats_equal(ctx.accum_read(), O.reduce_fused(masked, k, canon, tie, u)):
                print("MISMATCH quality", tag, "cutoff", cutoff); return 1
            counts["quality"] += 1
        elif n <= 300000:
            # batched compat face against the per-record oracle iterators
            cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, int(rng.integers(0, 30)))])).astype(np.int64)
            recs = [buf[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
            if not recs:
                continue
            if rng.random() < 0.25:
                # sequence::minimizer per record (ntk_minimizer_batch) against the oracle's per-record function; records shorter than m dropped
                mm = int(rng.choice([1, 2, 5, k, 31, 40]))
                keep = [r for r in recs if len(r) >= mm]
                if keep:
                    got = nt.minimizer_batch(keep, mm, ctx)
                    if any(g != O.minimizer(r, mm) for g, r in zip(got, keep)):
                        print("MISMATCH minimizer_batch", tag, "m", mm); return 1
                    counts["minimizer_batch"] = counts.get("minimizer_batch", 0) + 1
                continue
            u3 = rng.random()
            if u3 < 0.34:
                # the bit-plane form: any k <= 255 (the raw-byte kernel), records uploaded without break bytes
                kp = int(rng.choice([k, k, int(rng.integers(1, 64)), int(rng.integers(64, 256))]))
                ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, int(rng.choice([64, 97, 1000, 4096, 1 << 16])) if rng.random() < 0.5 else 0)
                pl = nt.canonical_kmers_planes(recs, kp, ctx=ctx)
                ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, 0)
                ok, tot = True, 0
                for i, r in enumerate(recs):
                    p_, f_ = O.canonical_kmers_arrays(r, O.reverse_complement(r), kp)
                    gp, gf = pl.arrays(i)
                    ok = ok and np.array_equal(gp, np.asarray(p_, dtype=np.uint64)) and np.array_equal(gf, np.asarray(f_, dtype=np.uint8))
                    tot += len(p_)
                ok = ok and pl.total == tot and int(np.unpackbits(pl.valid16.view(np.uint8)).sum()) == tot
                if not ok:
                    print("MISMATCH compat planes", tag, "k", kp); return 1
                counts["compat_planes"] += 1
                continue
            if u3 >= 0.84:
                # Sequence::bit_kmers per record as bit planes (+ dense values, or values rebuilt on the host)
                c2, with_values = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, int(rng.choice([64, 97, 1000, 4096, 1 << 16])) if rng.random() < 0.5 else 0)
                pl = nt.bit_kmers_planes(recs, k, c2, ctx, values=with_values)
                ctx.set_option(NL.OPT_COMPAT_CHUNK_BYTES, 0)
                ok, tot = True, 0
                for i, r in enumerate(recs):
                    p_, v_, f_ = O.bit_kmers_arrays(r, k, c2)
                    gp, gv, gf = pl.arrays(i)
                    ok = ok and np.array_equal(gp, np.asarray(p_, dtype=np.uint64)) and np.array_equal(gv, np.asarray(v_, dtype=np.uint64)) and \
                        np.array_equal(gf, np.asarray(f_, dtype=np.uint8))
                    tot += len(p_)
                if not (ok and pl.total == tot):
                    print("MISMATCH bit planes", tag, "canonical", c2, "values", with_values); return 1
                counts["bit_planes"] = counts.get("bit_planes", 0) + 1
                continue
            if u3 < 0.67:
                cnt, pos, flg = nt.canonical_kmers_batch(recs, k, ctx=ctx)
                wp, wf = [], []
                for r in recs:
                    p_, f_ = O.canonical_kmers_arrays(r, O.reverse_complement(r), k)
                    wp.append(np.asarray(p_, dtype=np.uint64)); wf.append(np.asarray(f_, dtype=np.uint8))
                ok = [len(x) for x in wp] == list(map(int, cnt)) and np.array_equal(np.concatenate(wp) if wp else [], pos) and \
                    np.array_equal(np.concatenate(wf) if wf else [], flg)
            else:
                c2 = bool(rng.integers(0, 2))
                cnt, pos, val, flg = nt.bit_kmers_batch(recs, k, c2, ctx=ctx)
                wp, wv, wf = [], [], []
                for r in recs:
                    p_, v_, f_ = O.bit_kmers_arrays(r, k, c2)
                    wp.append(np.asarray(p_, dtype=np.uint64)); wv.append(np.asarray(v_, dtype=np.uint64)); wf.append(np.asarray(f_, dtype=np.uint8))
                ok = [len(x) for x in wp] == list(map(int, cnt)) and np.array_equal(np.concatenate(wp), pos) and \
                    np.array_equal(np.concatenate(wv), val) and np.array_equal(np.concatenate(wf), flg)
            if not ok:
                print("MISMATCH compat batch", tag); return 1
            counts["compat_batch"] += 1
    print(f"gpu_fuzz: seed {args.seed}, {args.seconds:.0f} s, {it} inputs, {n_bytes / 1e6:.1f} MB, all equal to the oracle: {counts}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
