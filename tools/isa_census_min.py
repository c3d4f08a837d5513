#!/usr/bin/env python3
"""Static census of the generic fused minimizer kernel (minimizer_scan_kernel, ntk_kernels.hpp / ntk_tile.hpp) from the compiler's ISA listing
(hipcc -S of needletail_amd/csrc/ntk_api.hip): registers of every instantiation, and for the two key forms (normalising byte path, no
quality stream) the VALU instructions of the tile loop's large basic blocks by issue class (profiles/r04a/README.md: 4.1 / 2.05 cycles).
The kernel's rounds sit behind wave-uniform branches on w, so a STATIC listing holds every round and every overlap shift once; what one tile
executes at a given (k, w) is in profiles/<round>/path_pmc.txt (counters).  Usage: python tools/isa_census_min.py > profiles/<round>/isa_census_min_generic.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "a.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                           os.path.join(root, "needletail_amd", "csrc", "ntk_api.hip")] + os.environ.get("NTK_CENSUS_FLAGS", "").split(), stderr=subprocess.DEVNULL)
    text = open(out).read()

FULL_RATE = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
             "v_mov_b32", "v_bitop3_b32", "v_cndmask_b32", "v_min_u16", "v_add_u16", "v_add_f32"}


def issue_class(line):
    toks = line.split()
    base = re.sub(r"_(e32|e64)$", "", toks[0])
    if base.endswith(("_dpp", "_sdwa")) or base not in FULL_RATE:
        return "half"
    ops = re.sub(r"\bvcc(_lo|_hi)?\b", "", " ".join(toks[1:]))
    return "half" if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]", ops) else "full"


names = re.findall(r"^(_ZN3ntk21minimizer_scan_kernel\S+):", text, re.M)
print("minimizer_scan_kernel<KW, TIE_RC, ACCEPT_U, QM, F64, MODE>: registers per instantiation")
for nm in names:
    body = text[text.index(nm + ":"):]
    t = re.search(r"ILi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)E", nm).groups()
    vg = re.search(r"[.]amdhsa_next_free_vgpr (\d+)", body).group(1)
    sc = re.search(r"; ScratchSize: (\d+)", body).group(1)
    oc = re.search(r"; Occupancy: (\d+)", body).group(1)
    print(f"  <{t[0]}, {t[1]}, {t[2]}, {t[3]}, {t[4]}, {t[5]}>  VGPRs {vg:>3}  scratch {sc:>3}  waves per SIMD {oc}")
for f64, md, label in (("1", "3", "f64 keys, 19 <= k <= 23"), ("0", "2", "general keys (26 <= k <= 31)")):
    nm = f"_ZN3ntk21minimizer_scan_kernelILi2ELb1ELb1ELb0ELb{f64}ELi{md}EEEvNS_8ScanArgsE"
    body = text[text.index(nm + ":"):]
    body = body[:body.index(".end_amdhsa_kernel")]
    blocks, cur = [], ["entry", []]
    for l in body.splitlines():
        if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
            blocks.append(cur); cur = [l.split(":")[0].strip("; "), []]
        elif l.strip().startswith("v_"):
            cur[1].append(l.strip())
    blocks.append(cur)
    total = collections.Counter()
    print(f"\n<2, true, true, false, {'true' if f64 == '1' else 'false'}> - {label}: basic blocks with more than 40 VALU instructions (static)")
    for name, lines in blocks:
        c = collections.Counter(issue_class(l) for l in lines)
        total.update(c)
        if len(lines) > 40:
            top = collections.Counter(l.split()[0] for l in lines).most_common(6)
            print(f"  {name:12s} {len(lines):4d} VALU = {c['half']:3d} half-rate + {c['full']:3d} full-rate = {c['half'] * 4.1 + c['full'] * 2.05:6.0f} issue cycles   "
                  + ", ".join(f"{n} {m}" for m, n in top))
    print(f"  whole kernel (every round and every overlap shift once): {sum(total.values())} VALU = {total['half']} half-rate + {total['full']} full-rate")
