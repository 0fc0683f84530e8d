import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ex4dgs_amd.scene import make_scene
from ex4dgs_amd.render import render
model, cam, bg = make_scene("cfg3", P=int(sys.argv[1]) if len(sys.argv) > 1 else 20000, device="cuda")
cam = cam.to("cuda")
for p in model.parameters(): p.requires_grad_(True)
def run(fused):
    for p in model.parameters(): p.grad = None
    model.fused = fused
    o = render(cam, model, None, bg, timestamp=137, near=4.0, far=300.0)
    o["render"].sum().backward()
    return {n: getattr(model, n).grad.clone() for n in ("_xyz", "_xyz_motion", "_features_rest", "_features_dc")}, o["radii"].clone()
a, ra = run(False); b, rb = run(True)
for n in a:
    d = (a[n] - b[n]).abs().reshape(a[n].shape[0], -1).max(1).values
    bad = torch.nonzero(d > 1e-4 * max(1.0, float(a[n].abs().max()))).flatten()
    print(n, "max diff", float(d.max()), "bad rows", bad.numel(), bad[:20].tolist(), [int(x) % 64 for x in bad[:20].tolist()])
Ns = model.num_static
vis = (ra > 0)
print("Ns", Ns, "visible frac", float(vis.float().mean()))
