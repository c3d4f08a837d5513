#!/bin/bash
# Builds the A/B binaries of tools/kbench.hip against the CURRENT kernel source: kb_cur = the round-1 scalar-validity
# k = 21 build, kb_s2* = the sv2 kernel and its experiment / ablation builds, plus the instruction micro-benchmark.
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV"
rm -f kb_*
hipcc $F -o kb_cur kbench.hip &
hipcc $F -DNTK_KB_SV2 -o kb_s2 kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NOLDS -o kb_s2_nolds kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_LOADSONLY -o kb_s2_loads kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NOEXEC -o kb_s2_noexec kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NOMASKALG -o kb_s2_nomaskalg kbench.hip &
wait
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NOSDWA -DNTK_ABL_NOMASKALG -o kb_s2_nosdwa kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NOWINDOWS -o kb_s2_nowindows kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NODIGEST -o kb_s2_nodigest kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_ABL_NODIGEST -DNTK_ABL_NOLDS -DNTK_ABL_NOEXEC -o kb_s2_noemit kbench.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench ubench.hip 2>/dev/null &
wait
ls kb_* ubench
