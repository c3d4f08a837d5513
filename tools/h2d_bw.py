import torch, time
for mb in (8, 64, 512, 1500):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d = torch.empty_like(h, device="cuda")
    for _ in range(2): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = max(1, 3000 // mb)
    for _ in range(n): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"pinned H2D {mb} MiB: {dt*1e3:.2f} ms  {(mb<<20)/dt/1e9:.1f} GB/s")
