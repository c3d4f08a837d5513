#!/bin/bash
# Round-3 third A/B batch (tools/r03c_run.sh): VALU work moved to the LDS pipe (xor digest as ds_xor_b32, cross-lane moves as ds_bpermute_b32)
cd "$(dirname "$0")"
S="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV -DNTK_KB_SV2 -DNTK_KB_HB=14 -mllvm -amdgpu-sched-strategy=iterative-ilp"
rm -f kb_r3_*
b() { hipcc $S $2 -o kb_r3_$1 kbench.hip 2>/dev/null || echo "build of $1 failed"; }
C="-DNTK_SV2_CMPIN -DNTK_SV2_NFWD_SCNT -DNTK_SV2_LAZYV"
b base "" &
b sl "$C" &
b sl_cxor "$C -DNTK_SV2_CXOR_LDS" &
b sl_bperm "$C -DNTK_XL_BPERMUTE" &
b sl_cxor_bperm "$C -DNTK_SV2_CXOR_LDS -DNTK_XL_BPERMUTE" &
b sl_cxor_g8 "$C -DNTK_SV2_CXOR_LDS -DNTK_SV2_G8" &
wait
b sl_cxor_bperm_g8 "$C -DNTK_SV2_CXOR_LDS -DNTK_XL_BPERMUTE -DNTK_SV2_G8" &
b sl_default_sched "$C -DNTK_SV2_CXOR_LDS" &
wait
S="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV -DNTK_KB_SV2 -DNTK_KB_HB=14"
b sl_cxor_defsched "$C -DNTK_SV2_CXOR_LDS"
ls kb_r3_*
