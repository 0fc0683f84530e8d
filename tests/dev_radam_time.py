"""Dev script (not a test): time torch.optim.RAdam over the 15 parameter groups of a cfg3-sized model."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ex4dgs_amd.scene import make_scene
model, cam, bg = make_scene("cfg3", device="cuda")
params = model.parameters()
n = sum(p.numel() for p in params)
print("parameters:", n, "floats =", n * 4 / 1e6, "MB")
for p in params:
    p.requires_grad_(True)
groups = [{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(params)]
for foreach in (None, False):
    opt = torch.optim.RAdam(groups, lr=0.001, foreach=foreach)
    for p in params:
        p.grad = torch.randn_like(p)
    for _ in range(8): opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): opt.step()
    torch.cuda.synchronize(); print(f"torch RAdam foreach={foreach}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step")
    t0 = time.perf_counter()
    for _ in range(20):
        opt.zero_grad(set_to_none=True)
        for p in params: p.grad = torch.empty_like(p)
    torch.cuda.synchronize(); print(f"  zero_grad+realloc: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
from ex4dgs_amd.optim import FusedRAdam
opt = FusedRAdam(groups, lr=0.001)
for p in params:
    p.grad = torch.randn_like(p)
for _ in range(8): opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): opt.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"FusedRAdam: {1e3 * dt:.3f} ms/step = {28 * n / dt / 1e12:.2f} TB/s of 28 B/element")
