"""Dev tool: what torch's own copy / reduce / fill kernels reach on this device at the sizes of the graph workload (BASELINE configs[3]:
256 MB of int32 rows) and beyond: the practical ceiling the streaming kernels are compared with."""
import torch, time
for n in (1 << 25, 1 << 26, 1 << 28):
    a = torch.empty(n, dtype=torch.int64, device="cuda").random_()
    b = torch.empty_like(a)
    def timeit(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    dt = timeit(lambda: b.copy_(a)); print(f"copy {n*8/1e6:.0f} MB: {dt*1e3:.3f} ms -> {2*n*8/dt/1e12:.2f} TB/s (read+write)")
    dt = timeit(lambda: a.sum()); print(f"read-only sum {n*8/1e6:.0f} MB: {dt*1e3:.3f} ms -> {n*8/dt/1e12:.2f} TB/s")
    dt = timeit(lambda: b.fill_(7)); print(f"write-only fill {n*8/1e6:.0f} MB: {dt*1e3:.3f} ms -> {n*8/dt/1e12:.2f} TB/s")
