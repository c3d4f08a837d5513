#!/bin/bash
# Round-3 second A/B batch (tools/r03b_run.sh): compare-inside variants of the masked region.
cd "$(dirname "$0")"
S="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV -DNTK_KB_SV2 -DNTK_KB_HB=14 -mllvm -amdgpu-sched-strategy=iterative-ilp"
rm -f kb_r3_*
b() { hipcc $S $2 -o kb_r3_$1 kbench.hip 2>/dev/null || echo "build of $1 failed"; }
C="-DNTK_SV2_CMPIN"
b base "" &
b scnt "$C -DNTK_SV2_NFWD_SCNT" &
b vaddc "$C -DNTK_SV2_NFWD_VADDC" &
b scnt_lazy "$C -DNTK_SV2_NFWD_SCNT -DNTK_SV2_LAZYV" &
b scnt_lazy_g8 "$C -DNTK_SV2_NFWD_SCNT -DNTK_SV2_LAZYV -DNTK_SV2_G8" &
b vaddc_lazy "$C -DNTK_SV2_NFWD_VADDC -DNTK_SV2_LAZYV" &
wait
b vaddc_lazy_g8 "$C -DNTK_SV2_NFWD_VADDC -DNTK_SV2_LAZYV -DNTK_SV2_G8" &
b scnt_g8 "$C -DNTK_SV2_NFWD_SCNT -DNTK_SV2_G8" &
b floor_scnt_lazy "$C -DNTK_SV2_NFWD_SCNT -DNTK_SV2_LAZYV -DNTK_ABL_FLOOR" &
b floor "-DNTK_ABL_FLOOR" &
b cell_lazy "$C -DNTK_SV2_LAZYV" &
wait
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench3 ubench3.hip 2>/dev/null
ls kb_r3_* ubench3
