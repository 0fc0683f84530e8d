"""Dev probe (GPU box): what bounds preprocess_fwd?  Stage time of the per-Gaussian forward kernel at 1.0 M Gaussians with less and less
memory traffic (SH degree 3 / 0, precomputed colours, no direction sums) -- if the time follows the bytes it is a memory-system bound,
if it stays it is latency / issue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers as h
from ex4dgs_amd import _C, build
build.build(); _C.load()
ins, st = h.scene_inputs("cfg3", t=137, dir_scale=0.0)
ins = {k: v.cuda() for k, v in ins.items()}
P = ins["means3D"].shape[0]
e = torch.Tensor([])
s = h.gpu_settings(st, "cuda")
colors = torch.rand(P, 3, device="cuda")


def run(name, shs, colors_precomp, degree, prepare):
    _C.profile_enable(True)
    acc = 0.0
    n = 12
    for i in range(n + 3):
        _C.rasterize_gaussians(s.bg, ins["means3D"], ins["dir3D"], colors_precomp, ins["opacities"], ins["scales"], ins["rotations"], 1.0, e,
                               s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.kernel_size, s.subpixel_offset, s.image_height, s.image_width,
                               shs, degree, s.campos, False, s.min_depth, s.max_depth, False, prepare_backward=prepare)
        torch.cuda.synchronize()
        if i >= 3:
            acc += dict(_C.profile_read(0))["preprocess_fwd"]
    _C.profile_enable(False)
    print(f"{name:48s} preprocess_fwd {1e3 * acc / n:7.1f} us", flush=True)


for tune in (0, 1):
    _C.set_option("preprocess_sh_predicate", tune)
    print("preprocess_sh_predicate", tune)
    run("SH degree 3, direction sums (bench)", ins["shs"], e, 3, True)
    run("SH degree 3", ins["shs"], e, 3, False)
    run("SH degree 1 (48 of 192 B of SH per Gaussian)", ins["shs"], e, 1, False)
    run("SH degree 0 (16 B)", ins["shs"], e, 0, False)
    run("precomputed colours (no SH at all)", e, colors, 0, False)
