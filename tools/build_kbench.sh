#!/bin/bash
# Builds the A/B binaries of tools/kbench.hip against the CURRENT kernel source (scalar-validity k = 21 build):
# kb_cur (as shipped) and the ablations used by tools/profile_round.sh, plus the instruction micro-benchmark.
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV"
rm -f kb_*
hipcc $F -o kb_cur kbench.hip &
hipcc $F -DNTK_ABL_NOHIST -o kb_nohist kbench.hip &
hipcc $F -DNTK_ABL_NODIGEST -o kb_nodigest kbench.hip &


hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench ubench.hip &
wait
ls -la kb_* ubench
