#!/usr/bin/env python3
"""Instruction census of the scan2 tile loop (k = 21 headline build) straight from the compiler's ISA listing:
hipcc -S of tools/kbench.hip, the kernel's innermost loop (the basic blocks between the tile loop's header and its back
edge), instructions counted by mnemonic and by class.  Usage: python tools/isa_census.py [K] > profiles/<round>/isa_census.txt
NTK_CENSUS_FLAGS="-DNTK_SV2_PRIV ..." adds build flags (A/B variants of the kernel)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

K = int(sys.argv[1]) if len(sys.argv) > 1 else 21
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DNTK_KB_FIX", "-DNTK_KB_SV", "-DNTK_KB_SV2",
                           "-DNTK_KB_HB=14", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-S", "--cuda-device-only", "-o", out, os.path.join(root, "tools", "kbench.hip")] + os.environ.get("NTK_CENSUS_FLAGS", "").split(),
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
name = f"_ZN3ntk12scan2_kernelILi{K}ELb1ELb1ELb0ELi14ELi0ELb0EEEvNS_8ScanArgsE"
body = text[text.index(name + ":"):]
body = body[:body.index(".end_amdhsa_kernel")]
lines = body.splitlines()
# the tile loop: the Depth=2 loop (blocks whose label comment says "Depth=2"), i.e. from the first such label to the last
# branch back into it
idx = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:.*Depth=2", l)]
if not idx:   # (builds without the prefetch branch inside the loop have a single-block tile loop)
    idx = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:.*Parent Loop", l)]
first, last = idx[0], idx[-1]
end = last
for i in range(last, len(lines)):
    if re.match(r"^\.LBB\d+_\d+:", lines[i]) and i > last and "Depth=2" not in lines[i]:
        end = i
        break
# basic blocks of the loop region; the tail-tile block (bytes beyond n_bytes: 16 v_cmp_gt_i64 against the lane's byte
# index) runs once per launch and is left out of the steady-state census
blocks, cur = [], []
for l in lines[first:end]:
    if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
loop = []
for b in blocks:
    if any("v_cmp_gt_i64" in l for l in b):
        continue
    loop += [l.split()[0] for l in b if l.startswith("\t") and not l.strip().startswith((";", "."))]
    loop += [l.split()[0] for l in b if re.match(r"^ *(v_|s_|ds_|buffer_)", l)]   # inline-asm lines are not tab-indented
cnt = collections.Counter(loop)
cls = collections.Counter()
for m, c in cnt.items():
    if m.startswith("v_"):
        cls["VALU"] += c
    elif m.startswith("ds_"):
        cls["LDS"] += c
    elif m.startswith("buffer_"):
        cls["VMEM"] += c
    elif m.startswith("s_"):
        cls["SALU (incl. branches, waitcnt, nop)"] += c
# issue classes measured in round 4 (profiles/r04a/README.md): 2 cycles per wave-instruction for the plain 32-bit ALU ops below when they
# carry no DPP / SDWA modifier and no SGPR operand (a literal is fine; vcc as the v_cndmask mask is fine), 4 cycles for everything else
FULL_RATE = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
             "v_mov_b32", "v_bitop3_b32", "v_cndmask_b32", "v_min_u16", "v_add_u16", "v_add_f32"}
def issue_class(line):
    toks = line.split()
    m = toks[0]
    base = re.sub(r"_(e32|e64)$", "", m)
    if base.endswith(("_dpp", "_sdwa")) or base not in FULL_RATE:
        return "half"
    ops = " ".join(toks[1:])
    ops = re.sub(r"\bvcc(_lo|_hi)?\b", "", ops)
    return "half" if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]", ops) else "full"
valu_lines = []
for b in blocks:
    if any("v_cmp_gt_i64" in l for l in b):
        continue
    valu_lines += [l.strip() for l in b if l.strip().startswith("v_")]
classes = collections.Counter(issue_class(l) for l in valu_lines)
if "--json" in sys.argv:
    import json
    print(json.dumps({"kernel": f"scan2_kernel<{K}, true, true, false, 14, 0, false>", "valu_per_tile": sum(classes.values()),
                      "half_rate": classes["half"], "full_rate": classes["full"], "salu_per_tile": cls["SALU (incl. branches, waitcnt, nop)"],
                      "lds_per_tile": cls["LDS"], "cycles_half": 4.1, "cycles_full": 2.05,
                      "source": "tools/isa_census.py --json (ISA listing of the tile loop; classes: profiles/r04a/README.md)"}))
    sys.exit(0)
print(f"scan2_kernel<{K}, true, true, false, 14, 0> {os.environ.get('NTK_CENSUS_FLAGS', '')}: tile loop, instructions per 992-base tile (steady state)")
print(f"  issue classes (profiles/r04a): {classes['half']} half-rate (4.1 cycles) + {classes['full']} full-rate (2.05 cycles) = "
      f"{classes['half'] * 4.1 + classes['full'] * 2.05:.0f} cycles of VALU issue per tile")
for k_, v in sorted(cls.items()):
    print(f"  {k_:40s} {v}")
for m, c in cnt.most_common():
    print(f"    {c:4d}  {m}")
rest = text[text.index(name + ":"):]
m, l_ = re.search(r"; NumVgprs: (\d+)", rest), re.search(r"; LDSByteSize: (\d+)", rest)
print("  VGPRs:", m.group(1) if m else "?", " LDS bytes:", l_.group(1) if l_ else "?")
