"""Fused per-frame attribute evaluation: one HIP forward + one HIP backward instead of the
reference's five Python getters and their autograd graph
(scene/c_gaussian_model.py:170-215, :330-375; utils/interpolations.py:33-93; paths under /root/reference).

`evaluate_attributes(params, t, ...)` is a torch.autograd.Function over the 15 parameter tensors of the model and
returns `(means3D[N,3], rotations[N,4], opacities[N,1], scales[N,3], shs[N,16,3])`, static rows first -- exactly the
five tensors gaussian_renderer.render() feeds to the rasterizer.  It calls the C ABI of include/ex4d_attributes.h
(libex4d_hip.so); no CPU fallback.
"""
import ctypes as C

import torch

from . import _C

PARAM_ORDER = ("_xyz", "_xyz_disp", "_rotation", "_opacity", "_scaling", "_features_dc", "_features_rest",
               "_xyz_motion", "_rotation_motion", "_opacity_motion", "_opacity_duration_center",
               "_opacity_duration_var", "_scaling_motion", "_features_dc_motion", "_features_rest_motion")

EXPORTS = ("ex4d_attributes_forward", "ex4d_attributes_backward", "ex4d_attributes_backward_sliced", "ex4d_attributes_last_error")


class Ex4dAttrParams(C.Structure):
    _fields_ = [("Ns", C.c_int32), ("Nd", C.c_int32), ("K", C.c_int32), ("k", C.c_int32), ("t", C.c_float), ("duration", C.c_float),
                ("delta", C.c_float), ("h00", C.c_float), ("h10", C.c_float), ("h01", C.c_float), ("h11", C.c_float),
                ("tau", C.c_float), ("var_min", C.c_float)]


def time_scalars(t, Ns, Nd, K, duration, interval, time_shift, var_pad):
    """The Python-number arithmetic of c_gaussian_model.py:184-186, :364 and interpolations.py:83-86 (double precision)."""
    tp = t + time_shift
    k = int(tp // interval)
    d = (tp % interval) / interval
    h00 = 2 * d ** 3 - 3 * d ** 2 + 1
    h10 = d ** 3 - 2 * d ** 2 + d
    h01 = -2 * d ** 3 + 3 * d ** 2
    h11 = d ** 3 - d ** 2
    return Ex4dAttrParams(Ns, Nd, K, k, float(t), float(max(duration, 1)), d, h00, h10, h01, h11, tp / interval, var_pad / interval)


def _lib():
    lib = _C.load()
    if not getattr(lib, "_attr_ready", False):
        lib.ex4d_attributes_last_error.restype = C.c_char_p
        lib.ex4d_attributes_forward.restype = C.c_int
        lib.ex4d_attributes_backward.restype = C.c_int
        lib.ex4d_attributes_forward.argtypes = [C.POINTER(Ex4dAttrParams)] + [C.c_void_p] * 21
        lib.ex4d_attributes_backward.argtypes = [C.POINTER(Ex4dAttrParams)] + [C.c_void_p] * 28
        lib.ex4d_attributes_backward_sliced.restype = C.c_int
        lib.ex4d_attributes_backward_sliced.argtypes = [C.POINTER(Ex4dAttrParams)] + [C.c_void_p] * 27 + [C.POINTER(C.c_int32), C.c_void_p]
        lib._attr_ready = True
    return lib


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


FEATURE_NAMES = ("_features_dc", "_features_rest", "_features_dc_motion", "_features_rest_motion")
_BWD_INPUTS = ("_opacity", "_scaling", "_rotation_motion", "_opacity_motion", "_opacity_duration_center", "_opacity_duration_var", "_scaling_motion")


def forward_raw(scal, params, with_shs=True):
    """One launch of ex4d_attributes_forward on the current stream.  params: the 15 tensors in PARAM_ORDER (contiguous float32, one
    ROCm device).  Returns [means3D, rotations, opacities, scales, shs-or-empty]; no autograd."""
    lib = _lib()
    dev = params[0].device
    if not params[0].is_cuda:
        raise RuntimeError(f"parameters are on {dev}: the fused attribute evaluation only runs on a ROCm GPU (no CPU fallback)")
    for x in params:
        if x.dtype != torch.float32 or x.device != dev or not x.is_contiguous():
            raise RuntimeError("all model parameters must be contiguous float32 tensors on the same ROCm device")
    N = scal.Ns + scal.Nd
    f32 = dict(dtype=torch.float32, device=dev)
    outs = [torch.empty(N, 3, **f32), torch.empty(N, 4, **f32), torch.empty(N, 1, **f32), torch.empty(N, 3, **f32),
            torch.empty(N, 16, 3, **f32) if with_shs else torch.empty(0, **f32)]
    with torch.cuda.device(dev):
        rc = lib.ex4d_attributes_forward(C.byref(scal), *[_ptr(x) for x in params], *[_ptr(o) for o in outs],
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(lib.ex4d_attributes_last_error().decode())
    return outs


SLICED_SHAPES = {"_xyz_motion": (4, 3), "_rotation_motion": (2, 4)}     # [Nd, slices, C] of ex4d_attributes_backward_sliced


def backward_raw(scal, params, grads_in, with_shs=True, out=None, sliced=False):
    """One launch of ex4d_attributes_backward on the current stream (sliced=True: ex4d_attributes_backward_sliced -- the gradients of
    `_xyz_motion` / `_rotation_motion` come back as [Nd,4,3] / [Nd,2,4] slices and the call returns (gradients, slice hint) with
    slice hint = (xyz first keyframe, 4, rotation first keyframe, 2)).  grads_in: dL/d(means3D, rotations, opacities, scales, shs)
    (None = zeros; the shs entry is ignored when with_shs is False).  out: optional list of 15 preallocated gradient tensors
    (PARAM_ORDER; None entries are allocated) -- every one is written exactly once, dense.  Returns the 15 gradients (None for the
    feature tensors when with_shs is False: their gradient comes out of the rasterizer's SplitSH path)."""
    lib = _lib()
    dev = params[0].device
    N = scal.Ns + scal.Nd
    f32 = dict(dtype=torch.float32, device=dev)
    shapes = ((N, 3), (N, 4), (N, 1), (N, 3), (N, 16, 3))
    gin = [(torch.zeros(*s, **f32) if g is None else g.contiguous()) for g, s in zip(grads_in, shapes)]
    if not with_shs:
        gin[4] = None                              # dL/dsh goes to the feature tensors through the rasterizer (SplitSH), not through here
    gout = []
    for i, (n, x) in enumerate(zip(PARAM_ORDER, params)):
        if not with_shs and n in FEATURE_NAMES:
            gout.append(None)
        else:
            shape = (x.shape[0],) + SLICED_SHAPES[n] if (sliced and n in SLICED_SHAPES) else tuple(x.shape)
            if out is not None and out[i] is not None:
                if tuple(out[i].shape) != shape or out[i].dtype != torch.float32 or out[i].device != dev or not out[i].is_contiguous():
                    raise RuntimeError(f"gradient buffer for {n} must be a contiguous float32 tensor of shape {shape} on {dev}")
                gout.append(out[i])
            else:
                gout.append(torch.empty(shape, **f32))
    byname = dict(zip(PARAM_ORDER, params))
    hint = (C.c_int32 * 4)()
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if sliced:
            rc = lib.ex4d_attributes_backward_sliced(C.byref(scal), *[_ptr(byname[n]) for n in _BWD_INPUTS],
                                                     *[_ptr(g) for g in gin], *[_ptr(g) for g in gout], hint, stream)
        else:
            rc = lib.ex4d_attributes_backward(C.byref(scal), *[_ptr(byname[n]) for n in _BWD_INPUTS],
                                              *[_ptr(g) for g in gin], *[_ptr(g) for g in gout], stream)
    if rc:
        raise RuntimeError(lib.ex4d_attributes_last_error().decode())
    return (gout, tuple(int(v) for v in hint)) if sliced else gout


class _EvaluateAttributes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scal, with_shs, on_backward, *params):
        p = [x.contiguous() for x in params]
        outs = forward_raw(scal, p, with_shs)
        ctx.scal = scal
        ctx.with_shs = with_shs
        ctx.on_backward = on_backward
        ctx.save_for_backward(*p)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_means3D, g_rotations, g_opacities, g_scales, g_shs):
        if ctx.on_backward is not None:
            ctx.on_backward()                      # the owner's cache of these outputs is stale from here on (graph consumed)
        gout = backward_raw(ctx.scal, list(ctx.saved_tensors), (g_means3D, g_rotations, g_opacities, g_scales, g_shs), ctx.with_shs)
        return (None, None, None) + tuple(gout)


def evaluate_attributes(params, t, duration=300, interval=10, time_shift=12, var_pad=3, with_shs=True, on_backward=None):
    """params: mapping with the 15 CGaussianModel parameter tensors (PARAM_ORDER).  Returns the five boundary tensors; with
    with_shs=False the [N,16,3] SH block is not gathered (fifth value empty): hand the rasterizer a SplitSH of the four feature
    tensors instead.  `on_backward` (optional callable) runs when a backward pass consumes the graph of this evaluation."""
    p = [params[n] for n in PARAM_ORDER]
    Ns, Nd = p[0].shape[0], p[7].shape[0]
    K = p[7].shape[1] if Nd > 0 else 0
    scal = time_scalars(t, Ns, Nd, K, duration, interval, time_shift, var_pad)
    return _EvaluateAttributes.apply(scal, bool(with_shs), on_backward, *p)
