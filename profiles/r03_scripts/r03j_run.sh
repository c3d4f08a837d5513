#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
python tools/compat_bench.py > $O/compat_plain.txt 2>&1
cat $O/compat_plain.txt
cd /tmp && export TMPDIR=/tmp
REPS=3 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/compat_trace -o p -- python $R/tools/compat_bench.py > $O/compat_traced.txt 2> $O/compat_trace.err
cat $O/compat_traced.txt
cd $R
python tools/compat_trace_summary.py $O/compat_trace > $O/compat_trace_summary.txt 2>&1
cat $O/compat_trace_summary.txt
find $O/compat_trace -name "*.csv" | head; find $O/compat_trace -name "*memory_copy_trace.csv" | head -1 | xargs head -3
