"""Dev script (not a test): time torch.optim.RAdam vs FusedRAdam over the 15 parameter groups of a cfg3-sized model."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene
from ex4dgs_amd.optim import FusedRAdam
model, cam, bg = make_scene("cfg3", device="cuda")
params = model.parameters()
n = sum(p.numel() for p in params)
print("parameters:", n, "floats =", n * 4 / 1e6, "MB")
for p in params:
    p.requires_grad_(True)
groups = [{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(params)]
for name, mk, reps in (("torch.optim.RAdam", lambda: torch.optim.RAdam(groups, lr=0.001), 20), ("FusedRAdam", lambda: FusedRAdam(groups, lr=0.001), 100)):
    if name.startswith("torch") and "--skip-torch" in sys.argv:
        continue
    opt = mk()
    for p in params:
        p.grad = torch.randn_like(p)
    for _ in range(8): opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): opt.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {1e3 * dt:.3f} ms/step = {28 * n / dt / 1e12:.2f} TB/s of 28 B/element")
