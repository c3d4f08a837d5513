#!/bin/bash
cd "$(dirname "$0")"
S0="--offload-arch=gfx950 -O3 -std=c++17 -DNTK_KB_FIX -DNTK_KB_SV -DNTK_KB_SV2 -DNTK_KB_HB=14"
S="$S0 -mllvm -amdgpu-sched-strategy=iterative-ilp"
rm -f kb_r3_*
b() { hipcc $3 $2 -o kb_r3_$1 kbench.hip 2>/dev/null || echo "build of $1 failed"; }
C="-DNTK_SV2_CMPIN -DNTK_SV2_NFWD_SCNT -DNTK_SV2_LAZYV"
b sl "$C" "$S" &
b sl_acc2 "$C -DNTK_SV2_ACC2" "$S" &
b sl_ldslast "$C -DNTK_SV2_LDSLAST" "$S" &
b sl_acc2_ldslast "$C -DNTK_SV2_ACC2 -DNTK_SV2_LDSLAST" "$S" &
b sl_maxilp "$C" "$S0 -mllvm -amdgpu-sched-strategy=max-ilp" &
b sl_minreg "$C" "$S0 -mllvm -amdgpu-sched-strategy=iterative-minreg" &
wait
b sl_maxocc "$C" "$S0 -mllvm -amdgpu-sched-strategy=iterative-maxocc" &
b sl_defsched "$C" "$S0" &
b sl_k31 "$C" "$S" &
wait
ls kb_r3_*
