"""render(): the glue between a (camera, timestamp, Gaussian model) and the rasterizer boundary.

Counterpart of gaussian_renderer/__init__.py:19-124 of the reference (same argument names, same nine
entries in the returned dict); the reference's own render() also runs unchanged on top of
ex4dgs_amd.diff_gaussian_rasterization_df -- this copy exists because the reference's Python never
ships to the GPU box and the bench / tests need the identical call sequence:
zero `screenspace_points` and `flow` grad-trap tensors that retain their gradients, a zero
subpixel_offset[H,W,2], prefiltered=False, min_depth/max_depth = near/far, shs (not colors_precomp),
scales/rotations (not cov3D_precomp).
"""
import math
from types import SimpleNamespace

import torch

from .diff_gaussian_rasterization_df import GaussianRasterizationSettings, GaussianRasterizer

DEFAULT_PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)


_ZERO_OFFSETS = {}


def _zero_offsets(H, W, device):
    """The all-zero [H,W,2] subpixel offset the reference allocates and fills on every call (gaussian_renderer/__init__.py:39-40);
    it is read-only, so one tensor per (H, W, device) is kept."""
    key = (H, W, str(device))
    if key not in _ZERO_OFFSETS:
        _ZERO_OFFSETS[key] = torch.zeros((H, W, 2), dtype=torch.float32, device=device)
    return _ZERO_OFFSETS[key]


def render(viewpoint_camera, pc, pipe, bg_color, timestamp=None, scaling_modifier=1.0, override_color=None,
           subpixel_offset=None, mode=0, training=False, near=0.2, far=100.0, sync=True):
    pipe = DEFAULT_PIPE if pipe is None else pipe
    if getattr(pipe, "compute_cov3D_python", False) or getattr(pipe, "convert_SHs_python", False):
        # gaussian_renderer/__init__.py:76-78, :87-93: the reference's Python-side covariance / SH evaluation bypasses the kernels
        # this package exists for; refusing is better than silently rendering through the native path instead
        raise NotImplementedError("pipe.compute_cov3D_python / pipe.convert_SHs_python are not supported: covariance and SH colour "
                                  "are evaluated by the HIP preprocess kernel")
    if mode not in (0, 1, 2):
        raise ValueError(f"mode must be 0 (all Gaussians), 1 (static only) or 2 (dynamic only), got {mode}")
    timestamp = timestamp if timestamp is not None else viewpoint_camera.timestamp
    means3D = pc.get_xyz_at_t(timestamp, mode=mode, training=training)
    device = means3D.device

    # gradient traps: the 2D-mean gradient lands in screenspace_points.grad, the per-Gaussian "flow"/error channel gradient
    # in flow.grad (gaussian_renderer/__init__.py:28-32, :66-70).  The reference builds both as `zeros(requires_grad=True) + 0`
    # plus retain_grad(); leaf tensors receive the same gradients without the `+ 0` copies, and the VALUES of means2D are never
    # read by the rasterizer (only dir3D's are: it is composited into the flow image), so that one need not even be filled.
    track = torch.is_grad_enabled()
    screenspace_points = torch.empty_like(means3D, requires_grad=track)
    flow = torch.zeros_like(means3D, requires_grad=track)

    H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    if subpixel_offset is None:
        subpixel_offset = _zero_offsets(H, W, device)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W,
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        kernel_size=pc.kernel_size, subpixel_offset=subpixel_offset, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False,
        min_depth=near, max_depth=far, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=settings)

    opacity = pc.get_opacity_at_t(timestamp, mode=mode, training=training)
    scales = pc.get_scaling(mode=mode)
    rotations = pc.get_rotation_at_t(timestamp, mode=mode)
    shs, colors_precomp = (pc.get_features(mode=mode), None) if override_color is None else (None, override_color)

    rendered_image, radii, rendered_depth, out_flow, acc, idxs = rasterizer(
        means3D=means3D, means2D=screenspace_points, dir3D=flow, shs=shs, colors_precomp=colors_precomp,
        opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None)
    if sync and means3D.is_cuda:
        torch.cuda.synchronize()        # gaussian_renderer/__init__.py:111
    return {"render": rendered_image, "depth": rendered_depth, "opticalflow": out_flow, "acc": acc,
            "viewspace_points": screenspace_points, "viewspace_l1points": flow, "dominent_idxs": idxs,
            "visibility_filter": radii > 0, "radii": radii}
