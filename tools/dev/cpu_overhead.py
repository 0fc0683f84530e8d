"""Dev script: host-side (Python + ctypes + launch) cost of one fused training iteration, measured on a tiny scene."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene
from ex4dgs_amd.render import render
from ex4dgs_amd.loss import l1_ssim_loss
from ex4dgs_amd.optim import FusedRAdam
model, cam, bg = make_scene("cfg1", P=2000, device="cuda", fused=True)
cam = cam.to("cuda"); bg = bg.cuda()
for p in model.parameters():
    p.requires_grad_(True)
opt = FusedRAdam([{"params": [p], "lr": 1e-5, "name": str(i)} for i, p in enumerate(model.parameters())], lr=0.001)
gt = torch.rand(3, cam.image_height, cam.image_width, device="cuda")
def it(i):
    out = render(cam, model, None, bg, timestamp=i % 300, near=4.0, far=300.0, sync=False)
    loss, l1e, sse, hook = l1_ssim_loss(out["render"], gt, 0.2, acc=out["acc"])
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for i in range(50): it(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(500): it(i)
torch.cuda.synchronize(); print(f"tiny scene: {1e3 * (time.perf_counter() - t0) / 500:.3f} ms/iter (host-bound)")
pr = cProfile.Profile(); pr.enable()
for i in range(300): it(i)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
