mkdir -p gpurun_out/r06k; cd tools
for rep in 1 2 3; do for k in 21 31 16; do for v in s2_hb14 x_prefetch2; do KB_FWD=1 ./kb_$v 10000000 $k 512 768 20 fwd_${v}_k$k 24 256; done; done; done > ../gpurun_out/r06k/ab_fwd.txt 2>&1
for rep in 1 2; do for k in 21 31; do for v in s2_hb14 x_prefetch2; do ./kb_$v 10000000 $k 512 768 20 canon_${v}_k$k 24 256; done; done; done >> ../gpurun_out/r06k/ab_fwd.txt 2>&1
cut -c1-118 ../gpurun_out/r06k/ab_fwd.txt
