# scan-kernel time across k (reduce mode, config-2 reads) and block sizes
for t in 512; do for k in 4 11 16 21 31; do
python bench.py --k $k --threads $t --steps 30 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads=$t k=$k', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"
done; done
