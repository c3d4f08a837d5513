#!/bin/bash
# Builds the A/B binaries of tools/kbench.hip against the CURRENT kernel source, with the flags the library's scan2 object is
# built with (csrc/Makefile SCAN2_FLAGS): kb_cur = the round-1 scalar-validity k = 21 build, kb_s2_hb14 = the shipped sv2
# kernel, kb_a_* = its ablations (tools/profile_round.sh runs them), kb_a_floor = the floor kernel (window words + per-position
# work on synthetic register-resident words: no loads, no encode, no validity), kb_s2_default = the same kernel under the default
# scheduler, plus the instruction micro-benchmarks.
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KBENCH -DNTK_KB_FIX -DNTK_KB_SV"
S="$F -DNTK_KB_SV2 -DNTK_KB_HB=14 -mllvm -amdgpu-sched-strategy=iterative-ilp"
rm -f kb_*
hipcc $F -o kb_cur kbench.hip 2>/dev/null &
hipcc $S -o kb_s2_hb14 kbench.hip 2>/dev/null &
hipcc $F -DNTK_KB_SV2 -DNTK_KB_HB=14 -o kb_s2_default kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_NOLDS -o kb_a_nolds kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_LOADSONLY -o kb_a_loads kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_FLOOR -o kb_a_floor kbench.hip 2>/dev/null &
wait
hipcc $S -DNTK_ABL_NOMASKALG -o kb_a_nomaskalg kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_NOSDWA -DNTK_ABL_NOMASKALG -o kb_a_nosdwa kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_NODIGEST -o kb_a_nodigest kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_NODIGEST -DNTK_ABL_NOLDS -o kb_a_noemit kbench.hip 2>/dev/null &
hipcc $S -DNTK_ABL_NOEXEC -o kb_a_noexec kbench.hip 2>/dev/null &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench ubench.hip 2>/dev/null &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench3 ubench3.hip 2>/dev/null &
wait
ls kb_* ubench ubench3
