mkdir -p gpurun_out/r06f; cd tools
for rep in 1 2 3; do for v in s2_hb14 f_align64 f_align256 f_bias0 f_trackers; do [ -x ./kb_$v ] && ./kb_$v 10000000 21 512 768 20 ${v} 24 256; done; done > ../gpurun_out/r06f/ab_flags.txt 2>&1
cut -c1-110 ../gpurun_out/r06f/ab_flags.txt
