mkdir -p gpurun_out/r06o; cd tools
for rep in 1 2 3; do for r in 10000000 1000000 200000; do for v in s2_hb14 x_faststart; do ./kb_$v $r 21 512 768 20 ${v}_$r 24 256; done; done; done > ../gpurun_out/r06o/start.txt 2>&1
cut -c1-150 ../gpurun_out/r06o/start.txt
