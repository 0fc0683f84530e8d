"""Dev probe (GPU box): what the memory system gives simple streaming kernels (torch copy / add / sum) at the sizes of the per-Gaussian kernel."""
import torch, time
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (48, 192, 353, 768):
    n = mb * 1024 * 1024 // 4
    x = torch.rand(n, device="cuda"); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy   {mb:4d} MB read + {mb:4d} MB written: {t:7.1f} us  {2 * mb * 1.048576 / t:6.2f} TB/s")
    t = timeit(lambda: x.sum())
    print(f"sum    {mb:4d} MB read:                    {t:7.1f} us  {mb * 1.048576 / t:6.2f} TB/s")
    t = timeit(lambda: y.zero_())
    print(f"zero   {mb:4d} MB written:                 {t:7.1f} us  {mb * 1.048576 / t:6.2f} TB/s")
