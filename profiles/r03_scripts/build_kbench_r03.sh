#!/bin/bash
# Round-3 A/B binaries of tools/kbench.hip against the CURRENT kernel source (flags of csrc/Makefile's scan2 object):
#   kb_r3_base     the shipped k = 21 kernel            kb_r3_floor   floor kernel (NTK_ABL_FLOOR: window words + per-position work only)
#   kb_r3_append   forward count on ds_append           kb_r3_priv    lane-privatised histogram cells (prefix12 : lane & 3)
#   kb_r3_cmpin*   compare + select inside the masked region; _scnt: forward count = s_bcnt1 + s_add (no LDS op); _app: ds_append
# tools/r03a_run.sh runs them.
cd "$(dirname "$0")"
S="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV -DNTK_KB_SV2 -DNTK_KB_HB=14 -mllvm -amdgpu-sched-strategy=iterative-ilp"
rm -f kb_r3_*
b() { hipcc $S $2 -o kb_r3_$1 kbench.hip 2>/dev/null || echo "build of $1 failed"; }
b base "" &
b floor "-DNTK_ABL_FLOOR" &
b append "-DNTK_SV2_NFWD_APPEND" &
b priv "-DNTK_SV2_PRIV" &
b priv_append "-DNTK_SV2_PRIV -DNTK_SV2_NFWD_APPEND" &
b cmpin "-DNTK_SV2_CMPIN" &
wait
b cmpin_scnt "-DNTK_SV2_CMPIN -DNTK_SV2_NFWD_SCNT" &
b cmpin_app "-DNTK_SV2_CMPIN -DNTK_SV2_NFWD_APPEND" &
b cmpin_scnt_priv "-DNTK_SV2_CMPIN -DNTK_SV2_NFWD_SCNT -DNTK_SV2_PRIV" &
b cmpin_app_priv "-DNTK_SV2_CMPIN -DNTK_SV2_NFWD_APPEND -DNTK_SV2_PRIV" &
b nolds "-DNTK_ABL_NOLDS" &
b nodigest "-DNTK_ABL_NODIGEST" &
wait
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench3 ubench3.hip 2>/dev/null
ls kb_r3_* ubench3
