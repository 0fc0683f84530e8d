"""Drop-in for the reference's Python operator surface
(submodules/diff_gaussian_rasterization_df/diff_gaussian_rasterization_df/__init__.py):

  GaussianRasterizationSettings  -- NamedTuple, same 16 fields in the same order (:180-196)
  GaussianRasterizer             -- nn.Module, same forward(...) keywords and return tuple (:198-251)
  rasterize_gaussians / _RasterizeGaussians -- autograd.Function with the same 10 inputs, 6 outputs and
                                    gradient routing (:22-178)

so that gaussian_renderer.render(), train.py and render.py of the reference run unchanged with
`ex4dgs_amd/` on sys.path.  The native layer underneath is ex4dgs_amd._C (ctypes -> libex4d_hip.so).

One extension: wherever `shs` goes, a `SplitSH(features_dc, features_rest, features_dc_motion, features_rest_motion)` is accepted
too -- the four tensors CGaussianModel.get_features() concatenates every frame (scene/c_gaussian_model.py:337-353).  The kernels
then read the coefficients where the model keeps them and write dL/dsh straight into four gradient tensors: no [P,16,3] copy
forward, no split backward.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _C
from .._C import SplitSH


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    min_depth: float
    max_depth: float
    debug: bool


class AsyncFrames:
    """Opt-in policy of the autograd surface for the ASYNCHRONOUS forward (include/ex4d_rasterizer.h: Ex4dParams.instance_capacity).

    The reference blocks the host once per frame to read the instance count back (rasterizer_impl.cu:298-299) and so does this
    package by default.  `async_frames.enable()` removes that wait: the binning buffer is sized for
    `headroom x the largest instance count seen so far` (the first frame runs synchronously and seeds it), `ctx.num_rendered`
    becomes a `_C.PendingFrame` (an int on demand) and the status of a frame is looked at when it costs nothing -- at the start
    of the NEXT forward and at the start of every backward, by which time its copy has long landed (`drain()` settles whatever is
    still outstanding at the end of a loop: call it before trusting the last frames).  A frame whose instance count exceeded the capacity had its tile
    lists truncated: its outputs and gradients are invalid.  That cannot be repaired behind the caller's back (the image was
    already consumed), so it is reported: `strict` (default) raises at the next forward, otherwise `invalid_frames` counts and the
    capacity grows.  Callers that own the whole iteration -- trainer.FrameTrainer(async_forward=True), under a policy object of its
    own (`use_policy`), and the compiled NativeTrainer.set_async -- check the status before their optimizer step and RE-RUN the frame
    instead: exact results, and the wait sits where every kernel of the frame is already enqueued."""

    def __init__(self):
        self.enabled = False
        self.headroom = 1.25
        self.strict = True
        self.capacity = 0
        self.no_flow = False            # launch the flow-free kernel (learned from the frames' has_flow flag)
        self.pending = []
        self.invalid_frames = 0
        self.frames = 0

    def enable(self, headroom=1.25, capacity=0, strict=True):
        self.enabled, self.headroom, self.strict, self.capacity = True, float(headroom), bool(strict), int(capacity)
        self.pending, self.no_flow, self.invalid_frames, self.frames = [], False, 0, 0
        return self

    def disable(self):
        self.drain()
        self.enabled = False

    def observe(self, num_rendered):
        need = int(self.headroom * int(num_rendered)) + 4096
        if need > self.capacity:
            self.capacity = min(need, 0x7FFFFFFF)

    def _settle(self, fr):
        self.observe(fr.num_rendered)
        self.no_flow = not fr.has_flow
        if not fr.valid:
            self.invalid_frames += 1
            if self.strict:
                raise RuntimeError(f"asynchronous rasterizer frame invalid: {fr.num_rendered} tile instances exceed the capacity {fr.capacity}"
                                   if fr.overflowed else "asynchronous rasterizer frame invalid: dir3D was not zero although the flow-free kernel ran")

    def poll(self):
        """Settle every frame whose status has arrived (never blocks)."""
        while self.pending and self.pending[0].done():
            self._settle(self.pending.pop(0))

    def drain(self):
        """Settle every outstanding frame (waits for the stream)."""
        while self.pending:
            self._settle(self.pending.pop(0).wait())

    def next_call(self):
        """(instance_capacity, assume_no_flow) for the next forward; capacity 0 = run synchronously (nothing known yet)."""
        if not self.enabled:
            return 0, False
        self.poll()
        return self.capacity, self.no_flow

    def record(self, num_rendered):
        self.frames += 1
        if isinstance(num_rendered, _C.PendingFrame):
            self.pending.append(num_rendered)
        elif self.enabled:
            self.observe(num_rendered)


async_frames = AsyncFrames()          # the process-wide policy (disabled until somebody calls async_frames.enable())
_policy_stack = []


class use_policy:
    """`with use_policy(p): ...` -- forwards issued inside run under the AsyncFrames object `p` instead of the process-wide one
    (a trainer that re-runs overflowing frames itself keeps its own policy and leaves everybody else synchronous)."""

    def __init__(self, policy):
        self.policy = policy

    def __enter__(self):
        _policy_stack.append(self.policy)
        return self.policy

    def __exit__(self, *exc):
        _policy_stack.pop()
        return False


def current_policy():
    return _policy_stack[-1] if _policy_stack else async_frames


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_native(fn, args, debug, dump_name, what):
    """Debug mode keeps a CPU copy of the inputs and dumps it if the native call throws (reference :92-99, :152-159)."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {what}. Inputs written to {dump_name} for debugging.\n")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, dir3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, *split):
        s = raster_settings
        ctx.split = len(split) == 4
        if ctx.split:
            sh = SplitSH(*split)
        # positional order of RasterizeGaussiansCUDA (rasterize_points.cu:36-60)
        args = (s.bg, means3D, dir3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.kernel_size, s.subpixel_offset,
                s.image_height, s.image_width, sh, s.sh_degree, s.campos, s.prefiltered, s.min_depth, s.max_depth, s.debug)
        # a backward will follow iff some input wants a gradient: the forward then also leaves the SH direction sums of the SH
        # backward in its geometry buffer (it has the SH rows in registers anyway), and the backward does not read the SH tensors
        ctx.prepared = any(ctx.needs_input_grad)
        policy = current_policy()
        capacity, no_flow = (0, False) if s.debug else policy.next_call()      # (debug synchronises after every stage)
        native_fwd = lambda *a: _C.rasterize_gaussians(*a, prepare_backward=ctx.prepared, instance_capacity=capacity, assume_no_flow=no_flow)
        (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, acc, flow, idxs) = _call_native(
            native_fwd, args, s.debug, "snapshot_fw.dump", "forward")
        policy.record(num_rendered)
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered          # an int, or a _C.PendingFrame (asynchronous forward): the backward needs its capacity only
        # outputs that take no part in the loss arrive in backward as None instead of freshly filled zero tensors
        # (the native backward treats a null upstream gradient as zeros)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, torch.Tensor([]) if ctx.split else sh,
                              geomBuffer, binningBuffer, imgBuffer, depth, acc, flow, *split)
        ctx.mark_non_differentiable(radii, idxs)
        return color, radii, depth, flow, acc, idxs

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, grad_out_depth, grad_out_flow, grad_out_acc, _grad_idx):
        s = ctx.raster_settings
        # asynchronous forward: frames whose status has landed by now are settled here as well (never blocks) -- an overflow in the LAST
        # frame of a loop is then reported by that frame's own backward instead of waiting for a next forward that never comes
        # (callers that stop without a backward call async_frames.drain())
        if isinstance(ctx.num_rendered, _C.PendingFrame):
            current_policy().poll()
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
         geomBuffer, binningBuffer, imgBuffer, depth, acc, _flow) = ctx.saved_tensors[:13]
        none = torch.Tensor([])
        grad_out_color = none if grad_out_color is None else grad_out_color
        grad_out_depth = none if grad_out_depth is None else grad_out_depth
        grad_out_flow = none if grad_out_flow is None else grad_out_flow
        grad_out_acc = none if grad_out_acc is None else grad_out_acc
        if ctx.split:
            sh = SplitSH(*ctx.saved_tensors[13:])
        # positional order of RasterizeGaussiansBackwardCUDA (rasterize_points.cu:136-166)
        args = (s.bg, means3D, radii, colors_precomp, scales, rotations, depth, acc, s.min_depth, s.max_depth,
                s.scale_modifier, cov3Ds_precomp, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.kernel_size,
                s.subpixel_offset, grad_out_color, grad_out_depth, grad_out_flow, grad_out_acc, sh, s.sh_degree, s.campos,
                geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, s.debug)
        # gradients of inputs the forward did not have are not even written (the reference fills dL_dcolors / dL_dcov3D and drops them)
        need_colors, need_cov3D = colors_precomp.numel() != 0, cov3Ds_precomp.numel() != 0
        prepared = ctx.prepared                                # (read-only state: any number of backward passes may use it)
        native = lambda *a: _C.rasterize_gaussians_backward(*a, need_colors=need_colors, need_cov3D=need_cov3D, prepared=prepared)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations, grad_dir3D) = _call_native(native, args, s.debug, "snapshot_bw.dump", "backward")
        if not need_colors:
            grad_colors_precomp = None
        if not need_cov3D:
            grad_cov3Ds_precomp = None
        # one gradient per forward input, in input order (reference :165-176)
        if ctx.split:
            return (grad_means3D, grad_means2D, grad_dir3D, None, grad_colors_precomp, grad_opacities,
                    grad_scales, grad_rotations, grad_cov3Ds_precomp, None, *grad_sh)
        return (grad_means3D, grad_means2D, grad_dir3D, grad_sh, grad_colors_precomp, grad_opacities,
                grad_scales, grad_rotations, grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, dir3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    if isinstance(sh, SplitSH):
        return _RasterizeGaussians.apply(means3D, means2D, dir3D, torch.Tensor([]), colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings, *sh)
    return _RasterizeGaussians.apply(means3D, means2D, dir3D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean frustum-visibility mask (reference :203-213)."""
        s = self.raster_settings
        with torch.no_grad():
            return _C.mark_visible(positions, s.viewmatrix, s.projmatrix, s.min_depth, s.max_depth)

    def forward(self, means3D, means2D, dir3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
           ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # absent optionals travel as empty CPU tensors, exactly like the reference (:225-237)
        empty = lambda: torch.Tensor([])
        shs = empty() if shs is None else shs
        colors_precomp = empty() if colors_precomp is None else colors_precomp
        scales = empty() if scales is None else scales
        rotations = empty() if rotations is None else rotations
        cov3D_precomp = empty() if cov3D_precomp is None else cov3D_precomp
        dir3D = empty() if dir3D is None else dir3D
        return rasterize_gaussians(means3D, means2D, dir3D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
