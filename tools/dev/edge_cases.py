"""Dev script: edge cases of the fused training path (static-only model, tiny images, empty tensors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene
from ex4dgs_amd.render import render
from ex4dgs_amd.loss import l1_ssim_loss, ssim, psnr
from ex4dgs_amd.optim import FusedRAdam
from ex4dgs_amd.simple_knn._C import distCUDA2

model, cam, bg = make_scene("cfg2", P=5000, device="cuda", fused=True)     # static only: Nd = 0
assert model.num_dynamic == 0
for p in model.parameters():
    p.requires_grad_(True)
opt = FusedRAdam([{"params": [p], "lr": 1e-3, "name": str(i)} for i, p in enumerate(model.parameters())], lr=0.001)
gt = torch.rand(3, cam.image_height, cam.image_width, device="cuda")
for it in range(3):
    out = render(cam, model, None, bg, timestamp=5, near=4.0, far=300.0)
    loss, l1e, sse, hook = l1_ssim_loss(out["render"], gt, 0.2, acc=out["acc"])
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
print("static-only fused loop ok, loss", float(loss))
for shape in [(3, 1, 1), (3, 2, 40), (1, 11, 11), (3, 12, 5)]:
    x = torch.rand(*shape, device="cuda", requires_grad=True); y = torch.rand(*shape, device="cuda")
    l, a, b = l1_ssim_loss(x, y, 0.3); l.backward()
    assert torch.isfinite(l) and torch.isfinite(x.grad).all(), shape
print("tiny images ok; ssim", float(ssim(x.detach(), y)), "psnr", float(psnr(x.detach()[None], y[None]).mean()))
print("knn with coincident cloud:", distCUDA2(torch.zeros(10, 3, device="cuda")).tolist()[:3])
# P == 0 through the autograd surface
from ex4dgs_amd.diff_gaussian_rasterization_df import GaussianRasterizationSettings, GaussianRasterizer
import math
s = GaussianRasterizationSettings(64, 64, 0.5, 0.5, 0.1, torch.zeros(64, 64, 2, device="cuda"), torch.zeros(3, device="cuda"), 1.0,
                                  torch.eye(4, device="cuda"), torch.eye(4, device="cuda"), 3, torch.zeros(3, device="cuda"), False, 0.2, 100.0, False)
z = lambda *sh: torch.zeros(*sh, device="cuda")
o = GaussianRasterizer(s)(means3D=z(0, 3), means2D=z(0, 3), dir3D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3), scales=z(0, 3), rotations=z(0, 4))
print("P == 0:", [tuple(t.shape) for t in o])
