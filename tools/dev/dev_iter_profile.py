"""Dev script (not a test): the fused full training iteration (getters -> render -> L1+SSIM -> backward -> RAdam), for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene, CONFIGS
from ex4dgs_amd.render import render
from ex4dgs_amd.loss import l1_ssim_loss
from ex4dgs_amd.optim import FusedRAdam
cfg = CONFIGS["cfg3"]
model, cam, bg = make_scene("cfg3", device="cuda", fused=True)
cam = cam.to("cuda"); bg = bg.cuda()
for p in model.parameters():
    p.requires_grad_(True)
opt = FusedRAdam([{"params": [p], "lr": 1e-7, "name": str(i)} for i, p in enumerate(model.parameters())], lr=0.001)
gt = torch.rand(3, cfg.height, cfg.width, device="cuda")
stamps = [0, 137, 299]
def it(i):
    o = render(cam, model, None, bg, timestamp=stamps[i % 3], near=cfg.min_depth, far=cfg.max_depth, sync=False)
    loss, l1e, sse, hook = l1_ssim_loss(o["render"], gt, 0.2, acc=o["acc"])
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
for i in range(10): it(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for i in range(n): it(i)
torch.cuda.synchronize(); print(f"fused full iteration: {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
