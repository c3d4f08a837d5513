for m in "" "--sync-allreduce"; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline $m 2>&1 | tail -1 | cut -c100-260
done
python bench.py --steps 50 --warmup 5 --no-cpu-baseline | cut -c100-260
