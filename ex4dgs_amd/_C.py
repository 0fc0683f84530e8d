"""Host-side mirror of the reference's pybind module `diff_gaussian_rasterization_df._C`
(submodules/diff_gaussian_rasterization_df/ext.cpp:15-19): the same three functions with the same
positional arguments and return tuples as RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA /
markVisible (submodules/diff_gaussian_rasterization_df/rasterize_points.cu:35-133, :135-234, :236-259),
implemented by calling the C ABI of include/ex4d_rasterizer.h (libex4d_hip.so, hand-written HIP for
gfx950) through ctypes.  PyTorch is only used for device memory and the current HIP stream.

There is NO fallback: if the library is missing or a tensor is not on a ROCm device, the call raises.
"""
import ctypes as C
import os

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# EX4D_HIP_LIB: developer override (a variant build of the same sources, e.g. tools/dev/spill_probe.py); the product loads the in-tree library
_LIB_PATH = os.environ.get("EX4D_HIP_LIB") or os.path.join(_CSRC, "libex4d_hip.so")
_lib = None

NUM_CHANNELS = 3   # cuda_rasterizer/config.h:15


class Ex4dParams(C.Structure):
    _fields_ = [("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("kernel_size", C.c_float), ("scale_modifier", C.c_float),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("prepare_backward", C.c_int32), ("instance_capacity", C.c_int32), ("assume_no_flow", C.c_int32), ("reserved", C.c_int32)]


class PendingFrame:
    """`num_rendered` of an ASYNCHRONOUS forward (Ex4dParams.instance_capacity > 0): the capacity the binning buffer was sized for is
    known now, the frame's status (instance count, overflow, flow flag) arrives in pinned host memory behind an event.  Stands in for
    the reference's Python int: int(frame) waits for the event and returns the count (DGR/py:95-105 keeps it in ctx.num_rendered and
    hands it back to the backward, which here only needs the capacity -- nobody waits unless somebody looks)."""
    __slots__ = ("capacity", "assumed_no_flow", "_status", "_event", "_pool")

    def __init__(self, capacity, assumed_no_flow, status, event, pool):
        self.capacity, self.assumed_no_flow, self._status, self._event, self._pool = int(capacity), bool(assumed_no_flow), status, event, pool

    def done(self):
        return self._event is not None and self._event.query()

    def wait(self):
        if self._event is not None:
            self._event.synchronize()
        else:                                  # forward recorded into a graph: its status is valid after a replay has finished
            torch.cuda.synchronize()
        return self

    def _word(self, i):
        return int(self.wait()._status[i].item()) & 0xFFFFFFFF

    @property
    def num_rendered(self):
        return self._word(0)

    @property
    def overflowed(self):
        """The frame has more (Gaussian, tile) instances than the capacity: its tile lists were truncated, its outputs are invalid."""
        return self.num_rendered > self.capacity

    @property
    def has_flow(self):
        return self._word(2) != 0

    @property
    def prefilter_violation(self):
        return self._word(1) != 0

    @property
    def valid(self):
        return not self.overflowed and not (self.assumed_no_flow and self.has_flow)

    def __int__(self):
        return self.num_rendered

    __index__ = __int__

    def __repr__(self):
        return f"PendingFrame(capacity={self.capacity}, " + (f"num_rendered={self.num_rendered})" if self.done() else "pending)")

    def __del__(self):
        try:
            if self._pool is not None and self._event is not None and not self._status.is_cuda:
                if self._event.query():
                    self._pool.append(self._status)      # the pinned status words go back to the pool once the copy has landed
                else:
                    # the asynchronous copy into these words may still be in flight: keep the view (and with it the pinned block) alive
                    # and hand the slot back once its event has passed (dropping it here lost the slot for good, and a block
                    # whose 16 views were all gone could be handed out again by torch's host allocator while a copy was pending)
                    _status_parked.append((self._event, self._status))
        except Exception:
            pass


_status_pool = []
_status_parked = []          # (event, pinned status words) of frames dropped before their status copy had landed


def _recycle_parked():
    if _status_parked:
        still = []
        for ev, st in _status_parked:
            if ev.query():
                _status_pool.append(st)
            else:
                still.append((ev, st))
        _status_parked[:] = still


def _pinned_status():
    """8 pinned int32 words for one frame's Ex4dFrameStatus (pooled: pinning host memory costs far more than a frame, and is not
    allowed while a stream is capturing -- the pool is filled 16 buffers at a time outside capture)."""
    _recycle_parked()
    if not _status_pool:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("no pinned status buffer left during graph capture: run one asynchronous forward before capturing")
        block = torch.zeros(16, 8, dtype=torch.int32).pin_memory()
        _status_pool.extend(block[i] for i in range(16))
    return _status_pool.pop()


class GeomLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("records", "cov3D", "clamped", "tiles_touched", "depth_order", "sorted_offsets", "rects", "total")]


class BinningLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("point_list", "tile_ids", "qlist", "qcount", "total")]


class ImgLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("final_T", "n_contrib", "ranges", "total")]


class Ex4dSplitSH(C.Structure):          # include/ex4d_rasterizer.h: Ex4dSplitSH / Ex4dSplitSHGrad (same layout)
    _fields_ = [("dc", C.c_void_p * 2), ("rest", C.c_void_p * 2), ("n_static", C.c_int32)]


class SplitSH(tuple):
    """The SH coefficients as CGaussianModel stores them -- (features_dc [Ns,1,3], features_rest [Ns,15,3],
    features_dc_motion [Nd,1,3], features_rest_motion [Nd,15,3]) -- accepted wherever the `sh` / `shs` tensor goes:
    rows [0,Ns) come from the first pair, rows [Ns,Ns+Nd) from the second, exactly get_features()'s concatenation
    (scene/c_gaussian_model.py:337-353) without materialising it; the backward returns a SplitSH of the four gradients."""
    def __new__(cls, dc_static, rest_static, dc_dynamic, rest_dynamic):
        return super().__new__(cls, (dc_static, rest_static, dc_dynamic, rest_dynamic))

    @property
    def n_static(self):
        return self[0].shape[0]

    @property
    def n_dynamic(self):
        return self[2].shape[0]

    def numel(self):
        return sum(t.numel() for t in self)

    def size(self, dim):
        return (self.n_static + self.n_dynamic, 16, 3)[dim]


def _split_struct(split, device, what):
    keep, ptrs = [], []
    for t, shape1 in zip(split, (1, 15, 1, 15)):
        if t.dim() != 3 or t.shape[1] != shape1 or t.shape[2] != 3:
            raise RuntimeError(f"{what}: expected [n,{shape1},3] tensors (dc/rest, static then dynamic)")
        kt, pt = _dev_f32(t, what, device)
        keep.append(kt); ptrs.append(pt)
    if split[0].shape[0] != split[1].shape[0] or split[2].shape[0] != split[3].shape[0]:
        raise RuntimeError(f"{what}: dc and rest must have the same number of rows")
    st = Ex4dSplitSH((C.c_void_p * 2)(ptrs[0], ptrs[2]), (C.c_void_p * 2)(ptrs[1], ptrs[3]), int(split[0].shape[0]))
    return keep, st


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

EXPORTS = ("ex4d_last_error", "ex4d_abi_version", "ex4d_target_arch", "ex4d_forward", "ex4d_backward",
           "ex4d_forward_split_sh", "ex4d_backward_split_sh",
           "ex4d_backward_scratch_bytes", "ex4d_mark_visible", "ex4d_geom_bytes", "ex4d_binning_bytes", "ex4d_img_bytes",
           "ex4d_geom_layout", "ex4d_binning_layout", "ex4d_img_layout",
           "ex4d_profile_enable", "ex4d_profile_read", "ex4d_set_option", "ex4d_get_option", "ex4d_debug_bwd_stats", "ex4d_debug_bwd_stats16", "ex4d_debug_rows_prof")


def library_path():
    return _LIB_PATH


def load():
    """dlopen libex4d_hip.so (built in-tree by ex4dgs_amd.build); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} not found: build it with `python -m ex4dgs_amd.build` "
                           "(there is no CPU / PyTorch fallback for the rasterizer)")
    lib = C.CDLL(_LIB_PATH)
    lib.ex4d_last_error.restype = C.c_char_p
    lib.ex4d_target_arch.restype = C.c_char_p
    lib.ex4d_abi_version.restype = C.c_int
    for n in ("ex4d_backward_scratch_bytes", "ex4d_geom_bytes", "ex4d_binning_bytes", "ex4d_img_bytes"):
        getattr(lib, n).restype = C.c_size_t
    lib.ex4d_forward.restype = C.c_int
    lib.ex4d_backward.restype = C.c_int
    lib.ex4d_mark_visible.restype = C.c_int
    lib.ex4d_forward.argtypes = ([C.POINTER(Ex4dParams)] + [C.c_void_p] * 13 + [ALLOC_FN, C.c_void_p] * 3
                                 + [C.c_void_p] * 6 + [C.c_void_p, C.POINTER(C.c_int32)])
    lib.ex4d_backward.argtypes = [C.POINTER(Ex4dParams), C.c_int32] + [C.c_void_p] * 32
    lib.ex4d_forward_split_sh.restype = C.c_int
    lib.ex4d_backward_split_sh.restype = C.c_int
    lib.ex4d_forward_split_sh.argtypes = ([C.POINTER(Ex4dParams)] + [C.c_void_p] * 3 + [C.POINTER(Ex4dSplitSH)] + [C.c_void_p] * 8
                                          + [ALLOC_FN, C.c_void_p] * 3 + [C.c_void_p] * 6 + [C.c_void_p, C.POINTER(C.c_int32)])
    lib.ex4d_backward_split_sh.argtypes = ([C.POINTER(Ex4dParams), C.c_int32] + [C.c_void_p] * 3 + [C.POINTER(Ex4dSplitSH)] + [C.c_void_p] * 21
                                           + [C.POINTER(Ex4dSplitSH)] + [C.c_void_p] * 5)
    lib.ex4d_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.ex4d_set_option.argtypes = [C.c_char_p, C.c_int]
    lib.ex4d_get_option.argtypes = [C.c_char_p]
    _lib = lib
    return lib


def _check(code):
    if code != 0:
        raise RuntimeError(load().ex4d_last_error().decode() or f"ex4d error {code}")


def _dev_f32(t, name, device):
    """contiguous float32 tensor on `device` -> (tensor kept alive, pointer); empty tensor -> (None, None)
    (the reference distinguishes absent optionals by a null data pointer, forward.cu:218,254)."""
    if t is None or t.numel() == 0:
        return None, None
    if t.device != device:
        raise RuntimeError(f"{name} must be on {device}, got {t.device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    t = t.contiguous()
    return t, t.data_ptr()


def _rows(t, name, rows, cols=None):
    """Raw pointers carry no shape: a per-Gaussian tensor with fewer rows than P (or an image-sized one with the wrong element
    count) would be read out of bounds by the kernels, so shapes are checked here."""
    if t is None or t.numel() == 0:
        return
    if t.size(0) != rows or (cols is not None and t.numel() != rows * cols):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {rows} rows" + (f" of {cols} values" if cols else ""))


def _numel(t, name, n):
    if t is not None and t.numel() not in (0, n):
        raise RuntimeError(f"{name} has {t.numel()} elements, expected {n}")


def _require_rocm(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the ex4dgs_amd rasterizer only runs on a ROCm GPU (no CPU fallback)")


def _params(P, D, M, W, H, tanfovx, tanfovy, kernel_size, scale_modifier, min_depth, max_depth, prefiltered, debug, prepare_backward=False,
            instance_capacity=0, assume_no_flow=False):
    return Ex4dParams(P, D, M, W, H, tanfovx, tanfovy, kernel_size, scale_modifier, min_depth, max_depth, int(bool(prefiltered)), int(bool(debug)),
                      int(bool(prepare_backward)), int(instance_capacity), int(bool(assume_no_flow)), 0)


def _resizer(t):
    """rasterize_points.cu:27-33 resizeFunctional: grow a torch byte tensor, hand back its data pointer."""
    def fn(_user, nbytes):
        t.resize_(int(nbytes))
        return t.data_ptr()
    return ALLOC_FN(fn)


def rasterize_gaussians(background, means3D, dir3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, image_height, image_width,
                        sh, degree, campos, prefiltered, min_depth, max_depth, debug, prepare_backward=False,
                        instance_capacity=0, assume_no_flow=False):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-133): 24 positional arguments ->
    (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, acc, flow, idx).
    prepare_backward (keyword, not in the reference): a backward will follow -- the forward also leaves the SH direction sums the
    backward needs (include/ex4d_rasterizer.h: Ex4dParams.prepare_backward); hand `prepared=True` to the backward on these buffers.
    instance_capacity > 0 (keyword, not in the reference): ASYNCHRONOUS forward -- no instance-count read-back, the host does not wait,
    every launch has a host-constant grid; num_rendered comes back as a PendingFrame (capacity now, count / overflow on demand).
    assume_no_flow: with it, launch the flow-free compositing kernel (the caller's dir3D is all zeros; PendingFrame.has_flow checks)."""
    lib = load()
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_rocm(means3D, "means3D")
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    geomBuffer = torch.empty(0, dtype=torch.uint8, device=dev)
    binningBuffer = torch.empty(0, dtype=torch.uint8, device=dev)
    imgBuffer = torch.empty(0, dtype=torch.uint8, device=dev)
    if P == 0:   # rasterize_points.cu:90
        return (0, torch.zeros(NUM_CHANNELS, H, W, **f32), torch.zeros(P, **i32), geomBuffer, binningBuffer, imgBuffer,
                torch.zeros(1, H, W, **f32), torch.zeros(1, H, W, **f32), torch.zeros(3, H, W, **f32), torch.full((1, H, W), -1, **i32))
    out_color = torch.empty(NUM_CHANNELS, H, W, **f32)
    radii = torch.empty(P, **i32)
    out_depth = torch.empty(1, H, W, **f32)
    out_acc = torch.empty(1, H, W, **f32)
    out_flow = torch.empty(3, H, W, **f32)
    out_idx = torch.empty(1, H, W, **i32)

    split = sh if isinstance(sh, SplitSH) else None
    if split is not None:
        if split.n_static + split.n_dynamic != P:
            raise RuntimeError("SplitSH rows do not add up to the number of Gaussians")
        sh = torch.empty(0, **f32)
    M = 16 if split is not None else (sh.size(1) if sh.numel() != 0 else 0)     # rasterize_points.cu:92-96
    for name, t, cols in (("dir3D", dir3D, 3), ("colors", colors, NUM_CHANNELS), ("opacity", opacity, 1), ("scales", scales, 3),
                          ("rotations", rotations, 4), ("cov3D_precomp", cov3D_precomp, 6), ("sh", sh, 3 * M)):
        _rows(t, name, P, cols)
    _numel(subpixel_offset, "subpixel_offset", 2 * H * W)
    _numel(background, "background", NUM_CHANNELS)
    for name, t in (("viewmatrix", viewmatrix), ("projmatrix", projmatrix)):
        _numel(t, name, 16)
    _numel(campos, "campos", 3)
    keep = []
    ptr = {}
    for name, t in (("background", background), ("means3D", means3D), ("dir3D", dir3D), ("sh", sh), ("colors", colors),
                    ("opacity", opacity), ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("campos", campos), ("subpixel_offset", subpixel_offset)):
        kt, ptr[name] = _dev_f32(t, name, dev)
        keep.append(kt)
    prm = _params(P, int(degree), M, W, H, tan_fovx, tan_fovy, kernel_size, scale_modifier, min_depth, max_depth, prefiltered, debug, prepare_backward,
                  instance_capacity, assume_no_flow)
    cbs = [_resizer(geomBuffer), _resizer(binningBuffer), _resizer(imgBuffer)]
    num_rendered = C.c_int32(0)
    count_ref = C.byref(num_rendered)
    status = None
    if instance_capacity > 0:
        # Ex4dFrameStatus: pinned host memory behind an event; for a call recorded into a graph, device memory (it is read with an
        # ordinary copy after a replay -- a pinned destination of a captured copy was observed to be clobbered between replays)
        status = torch.zeros(8, dtype=torch.int32, device=dev) if torch.cuda.is_current_stream_capturing() else _pinned_status()
        count_ref = C.cast(status.data_ptr(), C.POINTER(C.c_int32))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if split is not None:
            if colors.numel() != 0:
                raise RuntimeError("Please provide excatly one of either SHs or precomputed colors!")
            kp, st = _split_struct(split, dev, "sh")
            keep.append(kp)
            code = lib.ex4d_forward_split_sh(
                C.byref(prm), ptr["background"], ptr["means3D"], ptr["dir3D"], C.byref(st), ptr["opacity"],
                ptr["scales"], ptr["rotations"], ptr["cov3D_precomp"], ptr["viewmatrix"], ptr["projmatrix"], ptr["campos"],
                ptr["subpixel_offset"], cbs[0], None, cbs[1], None, cbs[2], None,
                out_color.data_ptr(), radii.data_ptr(), out_depth.data_ptr(), out_acc.data_ptr(), out_flow.data_ptr(), out_idx.data_ptr(),
                C.c_void_p(stream), count_ref)
        else:
            code = lib.ex4d_forward(
                C.byref(prm), ptr["background"], ptr["means3D"], ptr["dir3D"], ptr["sh"], ptr["colors"], ptr["opacity"],
                ptr["scales"], ptr["rotations"], ptr["cov3D_precomp"], ptr["viewmatrix"], ptr["projmatrix"], ptr["campos"],
                ptr["subpixel_offset"], cbs[0], None, cbs[1], None, cbs[2], None,
                out_color.data_ptr(), radii.data_ptr(), out_depth.data_ptr(), out_acc.data_ptr(), out_flow.data_ptr(), out_idx.data_ptr(),
                C.c_void_p(stream), count_ref)
        ev = None
        if status is not None and code == 0 and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
    _check(code)
    if status is not None:
        return (PendingFrame(instance_capacity, assume_no_flow, status, ev, _status_pool), out_color, radii, geomBuffer, binningBuffer, imgBuffer,
                out_depth, out_acc, out_flow, out_idx)
    return (int(num_rendered.value), out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_depth, out_acc, out_flow, out_idx)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, acc_depth, acc, min_depth, max_depth,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size,
                                 subpixel_offset, dL_dout_color, dL_dout_depth, dL_grad_out_flow, dL_grad_out_acc, sh, degree,
                                 campos, geomBuffer, R, binningBuffer, imageBuffer, debug, need_colors=True, need_cov3D=True, prepared=False):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:135-234): 30 positional arguments ->
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dflow).
    need_colors / need_cov3D = False: that gradient is not written at all (the C ABI takes NULL) and comes back as an empty tensor --
    the autograd op passes False when the forward had no colors_precomp / cov3D_precomp input to receive it."""
    lib = load()
    _require_rocm(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    if isinstance(R, PendingFrame):
        R = R.capacity            # asynchronous forward: the buffers are laid out for the capacity, the kernels read the actual ranges
    H, W = acc_depth.size(1), acc_depth.size(2)      # forward output [1,H,W]; any upstream gradient may be absent (empty = zeros)
    f32 = dict(dtype=torch.float32, device=dev)
    split = sh if isinstance(sh, SplitSH) else None
    if split is not None:
        sh = torch.empty(0, **f32)
    M = 16 if split is not None else (sh.size(1) if sh.numel() != 0 else 0)
    shapes = [(P, 3), (P, NUM_CHANNELS), (P, 1), (P, 3), (P, 6), (P, M, 3) if split is None else (0,), (P, 3), (P, 4), (P, 3)]
    for name, t, cols in (("radii", radii, 1), ("colors", colors, NUM_CHANNELS), ("scales", scales, 3), ("rotations", rotations, 4),
                          ("cov3D_precomp", cov3D_precomp, 6), ("sh", sh, 3 * M)):
        _rows(t, name, P, cols)
    if split is not None and split.n_static + split.n_dynamic != P:
        raise RuntimeError("SplitSH rows do not add up to the number of Gaussians")
    _numel(subpixel_offset, "subpixel_offset", 2 * H * W)
    for name, t, n in (("acc", acc, H * W), ("dL_dout_color", dL_dout_color, NUM_CHANNELS * H * W), ("dL_dout_depth", dL_dout_depth, H * W),
                       ("dL_grad_out_flow", dL_grad_out_flow, 3 * H * W), ("dL_grad_out_acc", dL_grad_out_acc, H * W),
                       ("viewmatrix", viewmatrix, 16), ("projmatrix", projmatrix, 16), ("campos", campos, 3)):
        _numel(t, name, n)
    if P == 0:   # rasterize_points.cu:189
        outs0 = [torch.zeros(*s, **f32) for s in shapes]
        if split is not None:
            outs0[5] = SplitSH(*[torch.zeros_like(t) for t in split])
        return tuple(outs0)
    outs = [torch.empty(*s, **f32) for s in shapes]
    if not need_colors:
        outs[1] = torch.empty(0, **f32)
    if not need_cov3D:
        outs[4] = torch.empty(0, **f32)
    optr = lambda t: t.data_ptr() if t.numel() else None
    if split is not None:
        outs[5] = SplitSH(*[torch.empty_like(t, memory_format=torch.contiguous_format) for t in split])
    keep = []
    ptr = {}
    for name, t in (("background", background), ("means3D", means3D), ("sh", sh), ("colors", colors), ("scales", scales),
                    ("rotations", rotations), ("cov3D_precomp", cov3D_precomp), ("viewmatrix", viewmatrix), ("projmatrix", projmatrix),
                    ("campos", campos), ("subpixel_offset", subpixel_offset), ("acc_depth", acc_depth), ("acc", acc),
                    ("dL_dout_color", dL_dout_color), ("dL_dout_depth", dL_dout_depth), ("dL_grad_out_flow", dL_grad_out_flow),
                    ("dL_grad_out_acc", dL_grad_out_acc)):
        kt, ptr[name] = _dev_f32(t, name, dev)
        keep.append(kt)
    radii_c = radii.contiguous()
    # prepared: the forward that produced geomBuffer left the SH direction sums in it (the backward then does not read the SH tensors)
    scratch = torch.empty(lib.ex4d_backward_scratch_bytes(P), dtype=torch.uint8, device=dev)
    scratch_ptr = scratch.data_ptr()
    prm = _params(P, int(degree), M, W, H, tan_fovx, tan_fovy, kernel_size, scale_modifier, min_depth, max_depth, False, debug, prepared)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if split is not None:
            kp, st = _split_struct(split, dev, "sh")
            kg, gst = _split_struct(outs[5], dev, "dL_dsh")
            keep += [kp, kg]
            code = lib.ex4d_backward_split_sh(
                C.byref(prm), C.c_int32(int(R)), ptr["background"], ptr["means3D"], radii_c.data_ptr(), C.byref(st),
                ptr["scales"], ptr["rotations"], ptr["cov3D_precomp"], ptr["viewmatrix"], ptr["projmatrix"], ptr["campos"],
                ptr["subpixel_offset"], ptr["acc_depth"], ptr["acc"],
                geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(),
                ptr["dL_dout_color"], ptr["dL_dout_depth"], ptr["dL_grad_out_flow"], ptr["dL_grad_out_acc"],
                outs[0].data_ptr(), optr(outs[1]), outs[2].data_ptr(), outs[3].data_ptr(), optr(outs[4]),
                C.byref(gst), outs[6].data_ptr(), outs[7].data_ptr(), outs[8].data_ptr(),
                scratch_ptr, C.c_void_p(stream))
            _check(code)
            rasterize_gaussians_backward.last_scratch = scratch
            return tuple(outs)
        code = lib.ex4d_backward(
            C.byref(prm), C.c_int32(int(R)), ptr["background"], ptr["means3D"], radii_c.data_ptr(), ptr["sh"], ptr["colors"],
            ptr["scales"], ptr["rotations"], ptr["cov3D_precomp"], ptr["viewmatrix"], ptr["projmatrix"], ptr["campos"],
            ptr["subpixel_offset"], ptr["acc_depth"], ptr["acc"],
            geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(),
            ptr["dL_dout_color"], ptr["dL_dout_depth"], ptr["dL_grad_out_flow"], ptr["dL_grad_out_acc"],
            outs[0].data_ptr(), optr(outs[1]), outs[2].data_ptr(), outs[3].data_ptr(), optr(outs[4]),
            outs[5].data_ptr() if M > 0 else None, outs[6].data_ptr(), outs[7].data_ptr(), outs[8].data_ptr(),
            scratch_ptr, C.c_void_p(stream))
    _check(code)
    rasterize_gaussians_backward.last_scratch = scratch      # kept for parity tests (internal accumulators)
    return tuple(outs)


def mark_visible(means3D, viewmatrix, projmatrix, min_depth, max_depth=3.4028234663852886e38):
    """markVisible (rasterize_points.cu:236-259).  The reference's Python wrapper passes 4 arguments to a
    5-argument C++ function (diff_gaussian_rasterization_df/__init__.py:207-211) and therefore cannot be
    called; here max_depth defaults to FLT_MAX so the 4-argument call works."""
    lib = load()
    _require_rocm(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros(P, dtype=torch.bool, device=dev)
    if P == 0:
        return present
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    (m, _), (v, _), (p, _) = _dev_f32(means3D, "means3D", dev), _dev_f32(viewmatrix, "viewmatrix", dev), _dev_f32(projmatrix, "projmatrix", dev)
    if v is None or p is None or v.numel() != 16 or p.numel() != 16:
        raise RuntimeError("viewmatrix and projmatrix must be 4x4 float32 tensors")
    with torch.cuda.device(dev):
        _check(lib.ex4d_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr(), C.c_float(min_depth), C.c_float(max_depth),
                                     present.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return present


# ---- parity-test helpers: typed views into the opaque buffers -------------------------------------
def geom_views(geomBuffer, P):
    lay = GeomLayout()
    load().ex4d_geom_layout(P, C.byref(lay))
    g = geomBuffer
    v = lambda off, n, dt: g[off: off + n * torch.empty(0, dtype=dt).element_size()].view(dt)
    rec = v(lay.records, 16 * P, torch.float32).view(P, 16)
    return dict(records=rec, depths=rec[:, 8], means2D=rec[:, 0:2], conic_opacity=rec[:, [2, 3, 4, 15]], rgb=rec[:, 9:12], dir3D=rec[:, 12:15],
                cov3D=v(lay.cov3D, 6 * P, torch.float32).view(P, 6), clamped=v(lay.clamped, P, torch.uint8),
                tiles_touched=v(lay.tiles_touched, P, torch.int32), depth_order=v(lay.depth_order, P, torch.int32),
                rects=v(lay.rects, 2 * P, torch.int32).view(P, 2))


def binning_views(binningBuffer, R, W, H):
    """R: the count the buffer was laid out for (the instance count, or the capacity of an asynchronous forward)."""
    if isinstance(R, PendingFrame):
        R = R.capacity
    lay = BinningLayout()
    load().ex4d_binning_layout(R, W, H, C.byref(lay))
    b = binningBuffer
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return dict(point_list=b[lay.point_list: lay.point_list + 4 * R].view(torch.int32),
                tile_ids=b[lay.tile_ids: lay.tile_ids + 4 * R].view(torch.int32),
                qlist=b[lay.qlist: lay.qlist + 16 * max(R, 1)].view(torch.int32),
                qcount=b[lay.qcount: lay.qcount + 16 * T].view(torch.int32).view(T, 4))


def img_views(imgBuffer, W, H):
    lay = ImgLayout()
    load().ex4d_img_layout(W, H, C.byref(lay))
    T = ((W + 15) // 16) * ((H + 15) // 16)
    i = imgBuffer
    return dict(final_T=i[lay.final_T: lay.final_T + 4 * W * H].view(torch.float32).view(H, W),
                n_contrib=i[lay.n_contrib: lay.n_contrib + 4 * W * H].view(torch.int32).view(H, W),
                ranges=i[lay.ranges: lay.ranges + 8 * T].view(torch.int32).view(T, 2))


def set_option(name, value):
    """Tuning knobs of include/ex4d_rasterizer.h (e.g. "composite_bwd_variant")."""
    _check(load().ex4d_set_option(name.encode(), int(value)))


def get_option(name):
    return int(load().ex4d_get_option(name.encode()))


def bwd_stats(reset=True, extended=False):
    """Developer counters of the compositing backward's variant 8 (ex4d_debug_bwd_stats): batches, valid Gaussians, steps run,
    steps skipped, contributing (pixel, Gaussian) pairs, Gaussians with a contributing pair, ... ; extended=True: the 16 counters of
    ex4d_debug_bwd_stats16 (include/ex4d_rasterizer.h lists them)."""
    n = 16 if extended else 8
    buf = (C.c_ulonglong * n)()
    fn = load().ex4d_debug_bwd_stats16 if extended else load().ex4d_debug_bwd_stats
    rc = fn(buf, int(bool(reset)))
    if rc != 0:
        raise RuntimeError("ex4d_debug_bwd_stats failed")
    return [int(x) for x in buf]


def profile_enable(on=True):
    load().ex4d_profile_enable(int(bool(on)))


def profile_read(which):
    """[(stage name, ms)] of the most recent forward (which=0) / backward (which=1) call."""
    ms = (C.c_float * 16)()
    names = (C.c_char_p * 16)()
    n = load().ex4d_profile_read(int(which), ms, names, 16)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]
