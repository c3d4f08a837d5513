#!/bin/bash
# Round-6 experiments on the headline kernel: variants of tools/kbench.hip built with -DNTK_X_* / -DNTK_ABL_HALFIMPORTS
# (see ntk_kernels.hpp emit_canon), run alternately at the bench workload, then two rocprofv3 --pmc passes per variant (separate runs).
# Usage (through gpurun): bash tools/x_round6.sh <tag> <reps> <variant> ...     Output: gpurun_out/<tag>/{ab.txt,pmc.txt}
TAG=$1; REPS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R/tools
for rep in $(seq 1 $REPS); do
  for v in "$@"; do timeout 120 ./kb_$v 10000000 21 512 768 20 ${v} 24 256; done
done >> $O/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1_$v -o p -- $R/tools/kb_$v 10000000 21 512 768 4 $v 24 256 > /dev/null 2> $O/pmc1_$v.err
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2_$v -o p -- $R/tools/kb_$v 10000000 21 512 768 4 $v 24 256 > /dev/null 2> $O/pmc2_$v.err
done
cd $R
python3 - "$O" "$@" <<'PY'
import collections, csv, glob, os, sys
O, variants = sys.argv[1], sys.argv[2:]
tiles = (10_000_000 * 151 + 15) // 16 / 62
ms = collections.defaultdict(list)
for l in open(os.path.join(O, "ab.txt")):
    p = l.split()
    if len(p) > 5 and p[3] == "avg": ms[p[0]].append(float(p[4]))
out = ["variant            ms (runs)                        VALU/tile SALU/tile LDS/tile  cycles/tile/SIMD  wave-cycles/tile  wait_inst_any/tile  active_inst_any/tile  lds_bank_conflict/tile"]
for v in variants:
    a = collections.defaultdict(list)
    for d in ("pmc1_", "pmc2_"):
        for f in glob.glob(os.path.join(O, d + v, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "scan2_kernel" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(x) / len(x) for k, x in a.items()}
    g = lambda k: m.get(k, float("nan"))
    out.append(f"{v:18s} {' '.join('%.4f' % x for x in ms[v]):32s} {g('SQ_INSTS_VALU') / tiles:9.1f} {g('SQ_INSTS_SALU') / tiles:9.1f} {g('SQ_INSTS_LDS') / tiles:8.1f} "
               f"{g('GRBM_GUI_ACTIVE') / 8 / (tiles / 1024):14.0f} {g('SQ_WAVE_CYCLES') * 4 / tiles:17.0f} {g('SQ_WAIT_INST_ANY') * 4 / tiles:18.0f} {g('SQ_ACTIVE_INST_ANY') * 4 / tiles:20.0f} {g('SQ_LDS_BANK_CONFLICT') / tiles:20.1f}")
open(os.path.join(O, "pmc.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
cut -c1-150 $O/ab.txt
