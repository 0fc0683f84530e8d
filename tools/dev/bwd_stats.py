"""Developer script (GPU box): lane-utilisation statistics of the scan compositing backward (variant 8)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h
from ex4dgs_amd import _C, build
build.build(); lib = _C.load()
_C.set_option("composite_bwd_variant", 8)
for cfg, P in (("cfg3", None), ("cfg2", None), ("cfg5", 300000)):
    ins, st = h.scene_inputs(cfg, P=P, t=137, dir_scale=0.0)
    g = h.gpu_forward_raw(ins, st)
    grads = h.upstream_grads(g["acc"].cpu(), st["image_height"], st["image_width"], seed=1)
    _C.bwd_stats(reset=True, extended=True)
    h.gpu_backward_raw(ins, g, grads); torch.cuda.synchronize()
    s = _C.bwd_stats(reset=True, extended=True)
    b, nv, run, skip, pairs, anyg, alive, top, bot, both, run_top, run_bot, empty_top, empty_bot = s[:14]
    print(f"{cfg} P={ins['means3D'].shape[0]} R={g['num_rendered']}: batches {b}, Gaussian slots {nv} ({nv / max(b,1):.1f}/batch), steps run {run} skipped {skip} "
          f"({run / max(b,1):.1f} of 16 per batch), contributing pairs {pairs} = {pairs / max(run,1):.1f} lanes of 64 per run step, "
          f"Gaussians with any contribution in their batch {anyg} = {anyg / max(nv,1):.3f} of the slots")
    print(f"    pairs alive by list position {alive} = {alive / max(16 * b * 64, 1):.3f} of all lane-steps; in range among alive {pairs / max(alive, 1):.3f}")
    print(f"    Gaussians: top rows only {top} ({top / max(anyg,1):.3f}), bottom only {bot} ({bot / max(anyg,1):.3f}), both {both} ({both / max(anyg,1):.3f}); "
          f"steps run top {run_top} bottom {run_bot}; batches with an empty top half {empty_top} ({empty_top / max(b,1):.3f}), empty bottom half {empty_bot} ({empty_bot / max(b,1):.3f})")
