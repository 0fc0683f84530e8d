"""Dev script (GPU box): wall time per iteration of the compiled host path (include/ex4d_trainer.h) at a given size.  Run under
`rocprofv3 --kernel-trace --stats` the kernel time total / iterations is the kernel sum it is compared with (tools/prof_round.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene, CONFIGS
from ex4dgs_amd.native_trainer import NativeTrainer
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = CONFIGS["cfg3"]
model, cam, bg = make_scene("cfg3", P=P, device="cuda", fused=True)
cam = cam.to("cuda"); bg = bg.cuda()
gt = torch.rand(3, cfg.height, cfg.width, device="cuda")
nt = NativeTrainer(model, cam, optimizer=True, lrs={n: 1e-7 for n in model.PARAM_NAMES}, near=cfg.min_depth, far=cfg.max_depth)
mode = "synchronous forward"
if len(sys.argv) > 3 and sys.argv[3] == "async":
    nt.set_async(True); mode = "asynchronous forward"
for i in range(8):
    nt.step(cam, bg, (0, 137, 299)[i % 3], gt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    nt.step(cam, bg, (0, 137, 299)[i % 3], gt)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"native trainer P={P} ({mode}): {1e3 * (time.perf_counter() - t0) / steps:.4f} ms/iteration over {steps} iterations (+8 warm-up), "
      f"host time inside step() {1e3 * t_host / steps:.4f} ms/iteration, R={nt.num_rendered}, replays={nt.replays()}")
