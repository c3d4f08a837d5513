#!/bin/bash
# Builds the A/B binaries of tools/kbench.hip against the CURRENT kernel source, with the flags the library's scan2 object is
# built with (csrc/Makefile SCAN2_FLAGS): kb_cur = the round-1 scalar-validity k = 21 build, kb_s2_hb14 = the shipped sv2
# kernel, kb_a_* = its ablations (tools/profile_round.sh runs them), kb_s2_default = the same kernel under the default
# scheduler, plus the instruction micro-benchmark.
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV"
S="$F -DNTK_KB_SV2 -DNTK_KB_HB=14 -mllvm -amdgpu-sched-strategy=iterative-ilp"
rm -f kb_*
hipcc $F -o kb_cur kbench.hip &
hipcc $S -o kb_s2_hb14 kbench.hip &
hipcc $F -DNTK_KB_SV2 -DNTK_KB_HB=14 -o kb_s2_default kbench.hip &
hipcc $S -DNTK_ABL_NOLDS -o kb_a_nolds kbench.hip &
hipcc $S -DNTK_ABL_LOADSONLY -o kb_a_loads kbench.hip &
hipcc $S -DNTK_ABL_NOEXEC -o kb_a_noexec kbench.hip &
wait
hipcc $S -DNTK_ABL_NOMASKALG -o kb_a_nomaskalg kbench.hip &
hipcc $S -DNTK_ABL_NOSDWA -DNTK_ABL_NOMASKALG -o kb_a_nosdwa kbench.hip &
hipcc $S -DNTK_ABL_NODIGEST -o kb_a_nodigest kbench.hip &
hipcc $S -DNTK_ABL_NODIGEST -DNTK_ABL_NOLDS -DNTK_ABL_NOEXEC -o kb_a_noemit kbench.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench ubench.hip 2>/dev/null &
wait
ls kb_* ubench
