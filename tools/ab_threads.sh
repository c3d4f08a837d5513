B="python bench.py --steps 100 --warmup 5 --no-cpu-baseline"
run() { for t in 1024 512 256; do echo "$1 threads=$t $($B --threads $t | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])')"; done; }
for i in 1 2; do
run occ7
cp needletail_amd/libneedletail_amd.so /tmp/keep.so; cp tools/_occ8.so needletail_amd/libneedletail_amd.so
run occ8
cp /tmp/keep.so needletail_amd/libneedletail_amd.so
done
