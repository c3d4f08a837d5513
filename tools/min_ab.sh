#!/bin/bash
# tools/min_ab.sh <out file> <lib ...>: tools/min_ab.py once per library build, twice round-robin (boxes drift)
O=$1; shift
for rep in 1 2; do for lib in "$@"; do NEEDLETAIL_AMD_LIB=$PWD/needletail_amd/$lib python tools/min_ab.py $NTK_AB_PAIRS >> $O 2>/dev/null; done; done
cat $O
