"""Developer script (GPU box): compositing-backward stage time with all upstream gradients vs the image gradient alone."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h
from ex4dgs_amd import _C, build
build.build(); _C.load()
ins, st = h.scene_inputs("cfg3", t=137)
d = {k: v.cuda() for k, v in ins.items()}
d["dir3D"] = torch.zeros_like(d["dir3D"])
g = h.gpu_forward_raw(d, st)
H, W = st["image_height"], st["image_width"]
full = [x.cuda() for x in h.upstream_grads(g["acc"].cpu(), H, W, seed=1)]
e = torch.Tensor([])
for name, grads in (("all four upstream gradients", full), ("image gradient only", [full[0], e, e, e])):
    _C.profile_enable(True)
    tot = {}
    for i in range(10):
        h.gpu_backward_raw(d, g, grads); torch.cuda.synchronize()
        if i >= 2:
            for n, ms in _C.profile_read(1):
                tot[n] = tot.get(n, 0) + ms / 8
    _C.profile_enable(False)
    print(name, {k: round(v, 4) for k, v in tot.items()})
