"""ctypes mirror of include/ex4d_trainer.h: the compiled host path of one training iteration.

    attributes -> render (SplitSH) -> L1+SSIM -> rasterizer backward -> attribute backward (sliced) -> RAdam
    (train.py:124-153, :244-255 of the reference for one view), sequenced in C++ on the current stream with a persistent workspace.

Same kernels as trainer.FrameTrainer + loss.l1_ssim_loss, without the Python / autograd / allocator work per iteration (one ctypes
call instead of ~40 tensor operations).  Single process; for N ranks use FrameTrainer (it owns the gradient exchange).
No CPU fallback: the library and a ROCm device are required.
"""
import ctypes as C
import math

import torch

from . import _C
from . import attributes as attr
from .loss import _WINDOW
from .trainer import reference_lrs

EXPORTS = ("ex4d_trainer_last_error", "ex4d_trainer_create", "ex4d_trainer_destroy", "ex4d_trainer_step", "ex4d_trainer_output",
           "ex4d_trainer_grad", "ex4d_trainer_read", "ex4d_trainer_bytes", "ex4d_trainer_time_scalars", "ex4d_trainer_set_lr",
           "ex4d_trainer_set_sh_degree", "ex4d_trainer_set_async", "ex4d_trainer_replays")


class Ex4dTrainerConfig(C.Structure):
    _fields_ = [("Ns", C.c_int32), ("Nd", C.c_int32), ("K", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("kernel_size", C.c_float), ("min_depth", C.c_float), ("max_depth", C.c_float),
                ("duration", C.c_double), ("interval", C.c_double), ("time_shift", C.c_double), ("var_pad", C.c_double),
                ("lambda_dssim", C.c_float), ("window", C.c_float * 11), ("lr", C.c_double * 15),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("optimizer", C.c_int32)]


def _lib():
    lib = _C.load()
    if not getattr(lib, "_trainer_ready", False):
        lib.ex4d_trainer_last_error.restype = C.c_char_p
        lib.ex4d_trainer_create.restype = C.c_void_p
        lib.ex4d_trainer_create.argtypes = [C.POINTER(Ex4dTrainerConfig), C.POINTER(C.c_void_p)]
        lib.ex4d_trainer_destroy.restype = None
        lib.ex4d_trainer_destroy.argtypes = [C.c_void_p]
        lib.ex4d_trainer_step.restype = C.c_int
        lib.ex4d_trainer_step.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 6 + [C.POINTER(C.c_int32)]
        lib.ex4d_trainer_output.restype = C.c_void_p
        lib.ex4d_trainer_output.argtypes = [C.c_void_p, C.c_int32]
        lib.ex4d_trainer_grad.restype = C.c_void_p
        lib.ex4d_trainer_grad.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        lib.ex4d_trainer_read.restype = C.c_int
        lib.ex4d_trainer_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.ex4d_trainer_time_scalars.restype = None
        lib.ex4d_trainer_time_scalars.argtypes = [C.POINTER(Ex4dTrainerConfig), C.c_double, C.POINTER(attr.Ex4dAttrParams)]
        lib.ex4d_trainer_set_lr.restype = C.c_int
        lib.ex4d_trainer_set_lr.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.ex4d_trainer_set_sh_degree.restype = C.c_int
        lib.ex4d_trainer_set_sh_degree.argtypes = [C.c_void_p, C.c_int32]
        lib.ex4d_trainer_set_async.restype = C.c_int
        lib.ex4d_trainer_set_async.argtypes = [C.c_void_p, C.c_int32]
        lib.ex4d_trainer_replays.restype = C.c_int64
        lib.ex4d_trainer_replays.argtypes = [C.c_void_p]
        lib.ex4d_trainer_bytes.restype = C.c_size_t
        lib.ex4d_trainer_bytes.argtypes = [C.c_void_p]
        lib._trainer_ready = True
    return lib


class NativeTrainer:
    """model: scene.DynamicGaussians on a ROCm device (its 15 parameter tensors are updated in place).  cam: the image size and field
    of view are fixed at construction; step() takes any camera of that size."""

    def __init__(self, model, cam, optimizer=True, lrs=None, lambda_dssim=0.2, near=0.2, far=300.0, betas=(0.9, 0.999), eps=1e-8,
                 spatial_lr_scale=1.0):
        """lrs: per-parameter learning rates overriding the reference table (trainer.reference_lrs(spatial_lr_scale));
        near / far default to the reference's dataset.near / dataset.far (arguments/__init__.py:74-75).  This is the render +
        L1/SSIM + RAdam core of the iteration (include/ex4d_trainer.h: SCOPE): no regularisers, no l1_accum hook, no densification."""
        self.model = model
        self.names = list(attr.PARAM_ORDER)
        self.params = [getattr(model, n) for n in self.names]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("NativeTrainer needs the model on a ROCm device (no CPU fallback)")
        for n, p in zip(self.names, self.params):
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError(f"{n} must be a contiguous float32 tensor on {dev}")
        self.device = dev
        self.H, self.W = int(cam.image_height), int(cam.image_width)
        cfg = Ex4dTrainerConfig()
        cfg.Ns, cfg.Nd = model.num_static, model.num_dynamic
        cfg.K = model._xyz_motion.shape[1] if model.num_dynamic else 0
        cfg.W, cfg.H, cfg.sh_degree = self.W, self.H, model.active_sh_degree
        cfg.tanfovx, cfg.tanfovy, cfg.kernel_size = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), model.kernel_size
        cfg.min_depth, cfg.max_depth = near, far
        cfg.duration, cfg.interval, cfg.time_shift, cfg.var_pad = model.duration, model.interval, model.time_shift, model.var_pad
        cfg.lambda_dssim = lambda_dssim
        cfg.window = (C.c_float * 11)(*[float(x) for x in _WINDOW])
        lrs = dict(reference_lrs(spatial_lr_scale), **(lrs or {}))
        cfg.lr = (C.c_double * 15)(*[float(lrs[n]) for n in self.names])
        cfg.beta1, cfg.beta2, cfg.eps, cfg.optimizer = betas[0], betas[1], eps, int(bool(optimizer))
        self.cfg = cfg
        lib = _lib()
        ptrs = (C.c_void_p * 15)(*[p.data_ptr() if p.numel() else None for p in self.params])
        with torch.cuda.device(dev):
            self.handle = lib.ex4d_trainer_create(C.byref(cfg), ptrs)
        if not self.handle:
            raise RuntimeError(lib.ex4d_trainer_last_error().decode())
        self.num_rendered = 0

    def step(self, cam, bg, t, gt_image):
        """One iteration on the current stream (asynchronous apart from the rasterizer's instance-count read-back)."""
        if int(cam.image_height) != self.H or int(cam.image_width) != self.W:
            raise RuntimeError("camera size differs from the one the trainer was built for")
        if tuple(gt_image.shape) != (3, self.H, self.W) or gt_image.dtype != torch.float32 or not gt_image.is_contiguous() or gt_image.device != self.device:
            raise RuntimeError(f"gt_image must be a contiguous float32 [3,{self.H},{self.W}] tensor on {self.device}")
        lib = _lib()
        R = C.c_int32(0)
        with torch.cuda.device(self.device):
            rc = lib.ex4d_trainer_step(self.handle, float(t), cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr(),
                                       cam.camera_center.data_ptr(), bg.data_ptr(), gt_image.data_ptr(),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(R))
        if rc:
            raise RuntimeError(lib.ex4d_trainer_last_error().decode())
        self.num_rendered = R.value
        if self.cfg.optimizer:
            torch.autograd.graph.increment_version(self.params)

    def set_lrs(self, lrs):
        """Learning rates from the next step on (dict name -> value; unnamed groups keep theirs): the reference's update_learning_rate."""
        cur = {n: self.cfg.lr[i] for i, n in enumerate(self.names)}
        cur.update(lrs)
        arr = (C.c_double * 15)(*[float(cur[n]) for n in self.names])
        if _lib().ex4d_trainer_set_lr(self.handle, arr):
            raise RuntimeError(_lib().ex4d_trainer_last_error().decode())
        self.cfg.lr = arr

    def set_sh_degree(self, degree):
        """Active SH degree from the next step on (oneupSHdegree, train.py:113-114)."""
        if _lib().ex4d_trainer_set_sh_degree(self.handle, int(degree)):
            raise RuntimeError(_lib().ex4d_trainer_last_error().decode())
        self.cfg.sh_degree = int(degree)
        self.model.active_sh_degree = int(degree)

    def set_async(self, on=True):
        """Asynchronous rasterizer forward (no instance-count read-back in the middle of the frame; include/ex4d_trainer.h):
        same parameters as the synchronous path -- a frame that overflows its capacity is re-run before the optimizer step."""
        if _lib().ex4d_trainer_set_async(self.handle, int(bool(on))):
            raise RuntimeError(_lib().ex4d_trainer_last_error().decode())

    def replays(self):
        return int(_lib().ex4d_trainer_replays(self.handle))

    def output(self, what):
        """Copies of the trainer's outputs of the last step: 'loss', 'render', 'radii', 'viewspace_grad', 'depth', 'acc'."""
        idx = {"loss": 0, "render": 1, "radii": 2, "viewspace_grad": 3, "depth": 4, "acc": 5}[what]
        P = self.cfg.Ns + self.cfg.Nd
        shape, dtype = {0: ((1,), torch.float32), 1: ((3, self.H, self.W), torch.float32), 2: ((P,), torch.int32), 3: ((P, 3), torch.float32),
                        4: ((1, self.H, self.W), torch.float32), 5: ((1, self.H, self.W), torch.float32)}[idx]
        return self._read(idx, shape, dtype)

    def grad(self, name):
        """Copy of the gradient of parameter `name` of the last step (slices [Nd,4,3] / [Nd,2,4] for the two keyframe tensors) and the
        slice hint (xyz first keyframe, 4, rotation first keyframe, 2)."""
        i = self.names.index(name)
        hint = (C.c_int32 * 4)()
        _lib().ex4d_trainer_grad(self.handle, i, hint)
        p = self.params[i]
        shape = (p.shape[0],) + attr.SLICED_SHAPES[name] if name in attr.SLICED_SHAPES else tuple(p.shape)
        return self._read(100 + i, shape, torch.float32), tuple(int(x) for x in hint)

    def _read(self, what, shape, dtype):
        out = torch.empty(shape, dtype=dtype, device=self.device)
        if out.numel():
            lib = _lib()
            with torch.cuda.device(self.device):
                if lib.ex4d_trainer_read(self.handle, what, out.data_ptr(), out.numel() * out.element_size(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)):
                    raise RuntimeError(lib.ex4d_trainer_last_error().decode())
        return out

    def bytes(self):
        return int(_lib().ex4d_trainer_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            _lib().ex4d_trainer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
