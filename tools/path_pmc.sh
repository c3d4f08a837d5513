#!/bin/bash
# Instruction counts and cycles per kernel family (canonical / forward-only x k = 4, 16, 21, 31; quality-masked; fused minimizers):
# separate rocprofv3 --pmc passes over tools/path_pmc_driver.py, condensed into gpurun_out/$1/path_pmc.txt
TAG=${1:-r02f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/ppmc_a -o p -- python $R/tools/path_pmc_driver.py > /dev/null 2> $O/ppmc_a.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ppmc_b -o p -- python $R/tools/path_pmc_driver.py > /dev/null 2> $O/ppmc_b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ppmc_t -o p -- python $R/tools/path_pmc_driver.py > /dev/null 2> $O/ppmc_t.err
cd $R
python3 - "$O" <<'PY'
import collections, csv, glob, os, re, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("ppmc_a", "ppmc_b"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "scan2_kernel" in r["Kernel_Name"] or "minimizer_scan_kernel" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(O, "ppmc_t", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e6
tiles62 = (10_000_000 * 151 + 15) // 16 / 62
out = ["kernel (scan2_kernel<K, TIE_RC, ACCEPT_U, QM, HB, W, FWD>; minimizer_scan_kernel<KW, TIE_RC, ACCEPT_U, QM, F64> at w = 11: 61 emitting lanes per tile)   ms(rocprof)  GB/s   VALU/tile  SALU/tile  LDS/tile  cycles/tile/SIMD  fetch/algorithmic"]
for name in sorted(agg, key=lambda s: [int(x) if x.isdigit() else x for x in re.findall(r"\d+|\D+", s)]):
    a = {k: sum(v) / len(v) for k, v in agg[name].items()}
    short = re.sub(r"void ntk::scan2_kernel<(.*)>\(.*", r"<\1>", name)
    short = re.sub(r"void ntk::minimizer_scan_kernel<(.*)>\(.*", r"minimizer_scan<\1>", short)
    tiles = tiles62 * 62 / 61 if "minimizer_scan_kernel" in name else tiles62
    ms = dur.get(name, float("nan"))
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8 / (tiles / 1024)
    out.append(f"{short:48s} {ms:8.4f} {1.51e9 / (ms * 1e-3) / 1e9:8.1f} {a.get('SQ_INSTS_VALU', 0) / tiles:9.1f} {a.get('SQ_INSTS_SALU', 0) / tiles:9.1f} "
               f"{a.get('SQ_INSTS_LDS', 0) / tiles:8.1f} {cyc:12.0f} {a.get('FETCH_SIZE', 0) * 2048 / 1.51e9:12.3f}")
open(os.path.join(O, "path_pmc.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
