"""Developer script (GPU box): lane-utilisation statistics of the scan compositing backward (variant 8)."""
import ctypes as C, os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h
from ex4dgs_amd import _C, build
build.build(); lib = _C.load()
_C.set_option("composite_bwd_variant", 8)
for cfg, P in (("cfg3", None), ("cfg5", 300000)):
    ins, st = h.scene_inputs(cfg, P=P, t=137)
    g = h.gpu_forward_raw(ins, st)
    grads = h.upstream_grads(g["acc"].cpu(), st["image_height"], st["image_width"], seed=1)
    buf = (C.c_ulonglong * 8)()
    lib.ex4d_debug_bwd_stats(buf, 1)
    h.gpu_backward_raw(ins, g, grads); torch.cuda.synchronize()
    lib.ex4d_debug_bwd_stats(buf, 1)
    b, nv, run, skip, pairs, anyg = [int(x) for x in buf[:6]]
    print(f"{cfg} P={ins['means3D'].shape[0]} R={g['num_rendered']}: batches {b}, Gaussian slots {nv} ({nv / max(b,1):.1f}/batch), steps run {run} skipped {skip} "
          f"({run / max(b,1):.1f} of 16 per batch), contributing pairs {pairs} = {pairs / max(run,1):.1f} lanes of 64 per run step, "
          f"Gaussians with any contribution in their batch {anyg} = {anyg / max(nv,1):.3f} of the slots")
