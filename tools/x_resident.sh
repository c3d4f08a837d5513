# the headline kernel on batches that fit the caches: 10 M reads (1.51 GB, HBM), 1 M (151 MB: Infinity Cache), 200 k (30 MB: L2) - is the
# tile's idle time a memory effect?  (kbench: 300 warm-up launches, 20 timed)
mkdir -p gpurun_out/r06m; cd tools
for rep in 1 2; do for r in 10000000 5000000 2000000 1000000 500000 200000; do ./kb_s2_hb14 $r 21 512 768 20 reads_$r 24 256; ./kb_a_loads $r 21 512 768 20 loads_$r 24 256; done; done > ../gpurun_out/r06m/resident.txt 2>&1
cut -c1-120 ../gpurun_out/r06m/resident.txt
