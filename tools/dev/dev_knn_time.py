"""Dev script (not a test): distCUDA2 timing at 1M points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.simple_knn._C import distCUDA2
for P in (100_000, 1_000_000):
    g = torch.Generator().manual_seed(1)
    pts = torch.cat([torch.rand(P // 2, 3, generator=g) * 20, torch.randn(P // 2, 3, generator=g) * 3]).cuda()
    for _ in range(2): d = distCUDA2(pts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): d = distCUDA2(pts)
    torch.cuda.synchronize(); print(f"distCUDA2 P={P}: {1e3 * (time.perf_counter() - t0) / 5:.3f} ms  mean={d.mean().item():.6g}")
