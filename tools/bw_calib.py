import torch, time
for n in (1<<26, 1<<28):
    a = torch.empty(n, dtype=torch.int64, device="cuda").random_()
    b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): b.copy_(a)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"copy {n*8/1e6:.0f} MB: {dt*1e3:.3f} ms -> {2*n*8/dt/1e12:.2f} TB/s (read+write)")
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): s = a.sum()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"read-only sum {n*8/1e6:.0f} MB: {dt*1e3:.3f} ms -> {n*8/dt/1e12:.2f} TB/s")
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): b.fill_(7)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"write-only fill {n*8/1e6:.0f} MB: {dt*1e3:.3f} ms -> {n*8/dt/1e12:.2f} TB/s")
