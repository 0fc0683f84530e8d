"""Dev script: long run of the fused training iteration; memory must stay flat and the loss finite."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene, CONFIGS
from ex4dgs_amd.render import render
from ex4dgs_amd.loss import l1_ssim_loss
from ex4dgs_amd.optim import FusedRAdam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = CONFIGS["cfg3"]
model, cam, bg = make_scene("cfg3", P=300_000, device="cuda", fused=True)
cam = cam.to("cuda"); bg = bg.cuda()
for p in model.parameters():
    p.requires_grad_(True)
opt = FusedRAdam([{"params": [p], "lr": 1e-5, "name": str(i)} for i, p in enumerate(model.parameters())], lr=0.001)
gt = torch.rand(3, cfg.height, cfg.width, device="cuda")
t0 = time.time(); mem0 = None
for i in range(n):
    out = render(cam, model, None, bg, timestamp=i % 300, near=cfg.min_depth, far=cfg.max_depth, sync=False)
    loss, l1e, sse, hook = l1_ssim_loss(out["render"], gt, 0.2, acc=out["acc"])
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    if i % 500 == 499 or i == 49:
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated() / 2**20
        res = torch.cuda.memory_reserved() / 2**20
        if mem0 is None: mem0 = mem
        print(f"iter {i+1}: loss {float(loss):.5f} allocated {mem:.0f} MiB reserved {res:.0f} MiB  {1e3*(time.time()-t0)/(i+1):.2f} ms/iter", flush=True)
        assert torch.isfinite(loss) and mem < mem0 * 1.05 + 64
print("soak ok")
