"""Dev probe (GPU box): preprocess_fwd with stores left out (option "preprocess_probe"; the buffers downstream keep the previous frame's
contents -- same scene, same values)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers as h
from ex4dgs_amd import _C, build
build.build(); _C.load()
ins, st = h.scene_inputs("cfg3", t=137, dir_scale=0.0)
ins = {k: v.cuda() for k, v in ins.items()}
e = torch.Tensor([])
s = h.gpu_settings(st, "cuda")
def run(name, probe):
    _C.set_option("preprocess_probe", probe)
    _C.profile_enable(True)
    acc = 0.0; n = 12
    for i in range(n + 3):
        _C.rasterize_gaussians(s.bg, ins["means3D"], ins["dir3D"], e, ins["opacities"], ins["scales"], ins["rotations"], 1.0, e,
                               s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.kernel_size, s.subpixel_offset, s.image_height, s.image_width,
                               ins["shs"], 3, s.campos, False, s.min_depth, s.max_depth, False, prepare_backward=True)
        torch.cuda.synchronize()
        if i >= 3:
            acc += dict(_C.profile_read(0))["preprocess_fwd"]
    _C.profile_enable(False)
    _C.set_option("preprocess_probe", 0)
    print(f"{name:48s} preprocess_fwd {1e3 * acc / n:7.1f} us", flush=True)
run("all stores", 0)
for st in (1, 2, 3, 4):
    run(f"staggered start, {st} x 3.4 us per slot", st << 10)
run("all stores", 0)
