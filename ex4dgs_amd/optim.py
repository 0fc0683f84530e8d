"""Fused RAdam: a torch.optim.Optimizer with the constructor, param-group and state layout of
`torch.optim.RAdam` -- what the reference builds at scene/c_gaussian_model.py:449 and steps at train.py:250 -- whose
step() is ONE HIP launch over every parameter tensor (include/ex4d_optim.h) instead of ~10 element-wise passes per tensor.

State per parameter: {"step": float32 CPU scalar tensor, "exp_avg", "exp_avg_sq"} -- the same keys and shapes as
torch.optim.RAdam, so the reference's densification code that edits optimizer state in place
(c_gaussian_model.py: replace_tensor_to_optimizer / _prune_optimizer / cat_tensors_to_optimizer) works unchanged.
No CPU fallback: parameters must live on a ROCm device.
"""
import ctypes as C

import torch

from . import _C

EXPORTS = ("ex4d_optim_last_error", "ex4d_radam_step", "ex4d_radam_step_sliced")
MAX_WINDOWS = 8
MAX_SLICED = 4
MAX_TENSORS = 32


class Ex4dRadamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64), ("lr", C.c_double), ("step", C.c_int64), ("nan_to_num", C.c_int32), ("reserved", C.c_int32)]


class Ex4dRadamSlicedTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("rows", C.c_int64), ("K", C.c_int32), ("C", C.c_int32),
                ("lr", C.c_double), ("step", C.c_int64), ("n_windows", C.c_int32), ("first", C.c_int32 * 8), ("count", C.c_int32 * 8),
                ("grad", C.c_void_p * 8), ("first_dev", C.c_void_p)]


def _lib():
    lib = _C.load()
    if not getattr(lib, "_optim_ready", False):
        lib.ex4d_radam_step_sliced.restype = C.c_int
        lib.ex4d_radam_step_sliced.argtypes = [C.POINTER(Ex4dRadamSlicedTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p]
        lib.ex4d_optim_last_error.restype = C.c_char_p
        lib.ex4d_radam_step.restype = C.c_int
        lib.ex4d_radam_step.argtypes = [C.POINTER(Ex4dRadamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p]
        lib._optim_ready = True
    return lib


def radam_step_raw(items, betas, eps, device):
    """One fused launch (per <= 32 tensors) over raw element ranges.  items: iterable of
    (param_ptr, grad_ptr, exp_avg_ptr, exp_avg_sq_ptr, numel, lr, step[, nan_to_num]) -- plain device pointers, so a range may be a
    whole tensor or a rank's shard of one (RAdam is element-wise: updating ranges separately gives bit-identical results).
    nan_to_num = 1 reads the gradient through torch.nan_to_num (train.py:244-247 does that to _opacity_duration_var.grad)."""
    lib = _lib()
    descs = [Ex4dRadamTensor(int(it[0]), int(it[1]), int(it[2]), int(it[3]), int(it[4]), float(it[5]), int(it[6]), int(it[7]) if len(it) > 7 else 0, 0)
             for it in items if it[4] > 0]
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(0, len(descs), MAX_TENSORS):
            chunk = descs[i:i + MAX_TENSORS]
            arr = (Ex4dRadamTensor * len(chunk))(*chunk)
            if lib.ex4d_radam_step(arr, len(chunk), betas[0], betas[1], eps, stream):
                raise RuntimeError(lib.ex4d_optim_last_error().decode())


def radam_step_sliced_raw(items, betas, eps, device):
    """ex4d_radam_step_sliced over keyframe tensors [rows, K, C] with windowed gradients.  items: iterable of
    (param_ptr, exp_avg_ptr, exp_avg_sq_ptr, rows, K, C, lr, step, windows[, first_dev_ptr]) with windows = [(first, count, grad_ptr), ...]
    (<= 8), grad_ptr -> [rows, count, C] floats; first_dev_ptr (optional): device int32 array of the windows' first keyframes, read by
    the kernel instead of the host values (no device -> host round trip when the positions were gathered from other ranks).
    Bit-identical to radam_step_raw on the dense gradient the windows add up to."""
    lib = _lib()
    descs = []
    for it in items:
        (p, m, v, rows, K, Cc, lr, step, windows), first_dev = it[:9], (it[9] if len(it) > 9 else None)
        if rows <= 0:
            continue
        if len(windows) > MAX_WINDOWS:
            raise RuntimeError(f"at most {MAX_WINDOWS} gradient windows per tensor and step")
        if first_dev is None and any(w[0] is None for w in windows):
            raise RuntimeError("radam_step_sliced_raw: a window without a host position needs first_dev (SliceGather.first_device_ptr())")
        first = (C.c_int32 * 8)(*([(0 if w[0] is None else int(w[0])) for w in windows] + [0] * (8 - len(windows))))
        count = (C.c_int32 * 8)(*([w[1] for w in windows] + [0] * (8 - len(windows))))
        grad = (C.c_void_p * 8)(*([int(w[2]) for w in windows] + [None] * (8 - len(windows))))
        descs.append(Ex4dRadamSlicedTensor(int(p), int(m), int(v), int(rows), int(K), int(Cc), float(lr), int(step), len(windows), first, count, grad,
                                           int(first_dev) if first_dev else None))
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(0, len(descs), MAX_SLICED):
            chunk = descs[i:i + MAX_SLICED]
            arr = (Ex4dRadamSlicedTensor * len(chunk))(*chunk)
            if lib.ex4d_radam_step_sliced(arr, len(chunk), betas[0], betas[1], eps, stream):
                raise RuntimeError(lib.ex4d_optim_last_error().decode())


class FusedRAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise ValueError("FusedRAdam: weight_decay is not used by the reference (c_gaussian_model.py:449) and is not implemented")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("FusedRAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib()
        batches = {}                               # (device, betas, eps) -> descriptors
        touched = []                               # tensors the library writes: their autograd version counters are bumped below
        f32 = torch.float32
        for group in self.param_groups:
            key_tail = (tuple(group["betas"]), float(group["eps"]))
            lr = float(group["lr"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue                       # like torch: tensors without a gradient keep their step count
                if not p.is_cuda:
                    raise RuntimeError(f"parameter on {p.device}: FusedRAdam only runs on a ROCm GPU (no CPU fallback)")
                if p.dtype != f32 or g.dtype != f32 or g.is_sparse or not p.is_contiguous():
                    raise RuntimeError("FusedRAdam: parameters and gradients must be dense contiguous float32")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=f32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step_t = state["step"]
                step_t += 1
                if not g.is_contiguous():
                    g = g.contiguous()
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    raise RuntimeError("FusedRAdam: optimizer state must be contiguous")
                batches.setdefault((p.device,) + key_tail, []).append(
                    (Ex4dRadamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, int(step_t.item()), 0, 0), g))
                touched += (p, m, v)
        for (dev, betas, eps), items in batches.items():
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for i in range(0, len(items), MAX_TENSORS):
                    chunk = items[i:i + MAX_TENSORS]
                    arr = (Ex4dRadamTensor * len(chunk))(*[c[0] for c in chunk])
                    if lib.ex4d_radam_step(arr, len(chunk), betas[0], betas[1], eps, stream):
                        raise RuntimeError(lib.ex4d_optim_last_error().decode())
        # the library wrote through raw pointers: tell autograd the tensors changed (version counters), exactly what the
        # in-place torch ops of torch.optim.RAdam do -- caches keyed on parameter versions depend on it
        if touched:
            torch.autograd.graph.increment_version(touched)
        return loss
