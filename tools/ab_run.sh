#!/bin/bash
# A/B harness on the GPU box: runs tools/kb_<variant> binaries (built by tools/build_kbench.sh or by hand with -D flags) alternately,
# REPS times each, at the bench workload.  Usage (through gpurun): bash tools/ab_run.sh <tag> <k> <reps> <variant> [<variant> ...]
# Output: gpurun_out/<tag>/ab.txt (one line per run: avg/min/median ms and the reduced result, so that equality can be checked).
TAG=$1; K=$2; REPS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R/tools
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    timeout 120 ./kb_$v 10000000 $K 512 768 20 ${v}_k$K 24 256
  done
done >> $O/ab.txt 2>&1
cut -c1-132 $O/ab.txt
