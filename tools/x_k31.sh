mkdir -p gpurun_out/r06f; cd tools
for rep in 1 2 3; do for v in s2_hb14 a_halfimports; do ./kb_$v 10000000 31 512 768 20 ${v}_k31 24 256; done; done > ../gpurun_out/r06f/ab_k31.txt 2>&1
cut -c1-130 ../gpurun_out/r06f/ab_k31.txt
