#!/bin/bash
# tools/min_pmc.sh <tag> "<k,w> ...": rocprofv3 --pmc passes over tools/min_ab.py for the minimizer kernels -> gpurun_out/<tag>/min_pmc.txt
TAG=$1; PAIRS=${2:-"23,11 31,19"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python $R/tools/min_ab.py $PAIRS > /dev/null 2> $O/p$i.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt -o p -- python $R/tools/min_ab.py $PAIRS > /dev/null 2> $O/pt.err
cd $R
python3 - "$O" <<'PY'
import collections, csv, glob, os, re, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "minimizer_scan_kernel" in r["Kernel_Name"] or ("scan2_kernel" in r["Kernel_Name"]):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(O, "pt", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (float(r["AverageNs"]) / 1e6, r["Calls"])
out = []
for name, c in agg.items():
    short = re.sub(r"void ntk::(\w+)<(.*)>\(.*", r"\1<\2>", name)
    out.append(f"{short}  ms {dur.get(name)}")
    for k, v in sorted(c.items()):
        out.append(f"    {k:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
open(os.path.join(O, "min_pmc.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
